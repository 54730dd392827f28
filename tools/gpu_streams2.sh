#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for r in 1 2; do for st in 1 2; do
  echo "streams=$st: $(python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-ntt --streams $st 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], {k:v["avg_ms"] for k,v in d["kernels"].items()})')"
done; done | tee gpurun_out/streams2.txt
