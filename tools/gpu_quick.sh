#!/bin/bash
# quick loop: microbench + gpu tests + bench (no rocprof)
set -o pipefail
mkdir -p gpurun_out
(cd tools && hipcc --offload-arch=gfx950 -O3 -I../gpqhe_amd/csrc microbench.hip -o microbench && ./microbench) > gpurun_out/microbench.txt 2>&1
cat gpurun_out/microbench.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.txt | tail -5 || exit 1
timeout -k 10 600 python bench.py --steps 5 --warmup 2 "$@" 2>gpurun_out/bench.err | tee gpurun_out/bench.json || { tail -20 gpurun_out/bench.err; exit 1; }
