#!/bin/bash
# tests + smoke + bench + rocprofv3 kernel stats
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.txt | tail -5 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.txt || exit 1
timeout -k 10 600 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json || { tail -20 gpurun_out/bench.err; exit 1; }
rm -rf gpurun_out/prof && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-ntt > gpurun_out/bench_prof.json 2> gpurun_out/prof.err || { tail -20 gpurun_out/prof.err; exit 1; }
find gpurun_out/prof -name "*stats*" | head; cat $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) | head -20
