#!/bin/bash
# interleaved timing of several library builds on ONE device: tools/gpu_abc.sh rounds lib...
set -o pipefail
mkdir -p gpurun_out
R=$1; shift
for r in $(seq 1 $R); do
  for L in "$@"; do
    echo "$L: $(python bench.py --variant $PWD/$L --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], {k:v["avg_ms"] for k,v in d["kernels"].items()})')"
  done
done | tee gpurun_out/ab.txt
