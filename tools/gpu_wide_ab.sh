#!/bin/bash
# correctness of the wide-split forward butterflies, then interleaved A/B against one subtraction per stage (same box, same build)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.txt | tail -5 || exit 1
for i in 1 2; do
  for v in 0 1; do
    echo "== GPQHE_NO_WIDE=$v run $i"
    GPQHE_NO_WIDE=$v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 2>gpurun_out/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], [ (x['shape'][:18], x['GBps']) for x in d['ntt']], d.get('he_mul_mpi_level',{}).get('he_mul_per_s'), d.get('keyswitch_n17',{}).get('keyswitch_per_s'))
print({k:(round(v['avg_ms'],4) if isinstance(v,dict) and 'avg_ms' in v else v) for k,v in d.get('kernels',{}).items()})
" || { tail -20 gpurun_out/bench.err; exit 1; }
  done
done
