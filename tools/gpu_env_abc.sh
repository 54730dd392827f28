#!/bin/bash
# interleaved timing of library builds with one environment setting: tools/gpu_env_abc.sh "ENV=V ..." rounds lib...
set -o pipefail
mkdir -p gpurun_out
E=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for L in "$@"; do
    echo "$L: $(env $E GPQHE_HIP_LIB=$PWD/$L python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-ntt 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], {k:v["avg_ms"] for k,v in d["kernels"].items()})')"
  done
done | tee gpurun_out/ab.txt
