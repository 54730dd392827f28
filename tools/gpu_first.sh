#!/bin/bash
# first GPU session: ALU microbenchmarks + parity tests
set -o pipefail
mkdir -p gpurun_out
(cd tools && hipcc --offload-arch=gfx950 -O3 -I../gpqhe_amd/csrc microbench.hip -o microbench && ./microbench) > gpurun_out/microbench.txt 2>&1
cat gpurun_out/microbench.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.txt | tail -30
