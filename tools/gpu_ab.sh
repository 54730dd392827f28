#!/bin/bash
# interleaved A/B of two library builds on ONE device: tools/gpu_ab.sh <libA> <libB> [rounds]
set -o pipefail
mkdir -p gpurun_out
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for L in $A $B; do
    echo "$L: $(GPQHE_HIP_LIB=$PWD/$L python bench.py --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], {k:v["avg_ms"] for k,v in d["kernels"].items()})')"
  done
done | tee gpurun_out/ab.txt
