#!/bin/bash
# parity of the NTT / he_mul suites for the working build, then interleaved he_mul/s A/B of the given libraries:
#   tools/gpu_parity_ab.sh rounds lib...
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ntt_gpu.py tests/test_he_mul_gpu.py tests/test_he_mul_full_gpu.py tests/test_dense_full_size_gpu.py tests/test_bridge_gpu.py tests/test_parity_sweep_gpu.py -m gpu -x -q 2>&1 | tee gpurun_out/pytest_parity.txt || exit 1
bash tools/gpu_abc.sh "$@"
