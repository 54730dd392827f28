#!/bin/bash
# round 2, first call: GPU suite (incl. the two-rank HIP test), bench line, Infinity-Cache probe, limb-block A/B
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_dist_hip_gpu.py -m gpu -x -q 2>&1 | tee gpurun_out/pytest_dist.txt || exit 1
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_dist_hip_gpu.py 2>&1 | tee gpurun_out/pytest_gpu.txt || exit 1
timeout -k 10 500 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-600 || { tail -20 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python tools/mall_probe.py 48 2>gpurun_out/mall.err | tee gpurun_out/mall_probe.txt || { tail -20 gpurun_out/mall.err; exit 1; }
for r in 1 2; do for lb in 0 1 2 3 5 8 15; do
  echo "limb_block $lb: $(timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-ntt --limb-block $lb 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], {k:v["avg_ms"] for k,v in d["kernels"].items()})')"
done; done | tee gpurun_out/limb_block_ab.txt
