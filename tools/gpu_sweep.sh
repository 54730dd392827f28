#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for c in 1 2 4 8 16 32; do
  echo "chunk $c: $(python bench.py --steps 5 --warmup 2 --chunk $c --cpu-sample 0 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], {k:v["avg_ms"] for k,v in d["kernels"].items()})')"
done | tee gpurun_out/sweep_chunk.txt
