#!/bin/bash
# correctness of the 8-per-lane middle kernels, then interleaved A/B (same box)
set -o pipefail
mkdir -p gpurun_out
GPQHE_MID8=3 timeout -k 10 900 python -m pytest tests/test_he_mul_gpu.py tests/test_he_mul_full_gpu.py tests/test_parity_sweep_gpu.py -m gpu -x -q 2>&1 | tail -5 || exit 1
for i in 1 2; do
  for v in 0 1 2 3; do
    for L in gpqhe_amd/libgpqhe_hip.so gpqhe_amd/libgpqhe_hip_K3.so; do
    if [ $L != gpqhe_amd/libgpqhe_hip.so ] && [ $v -lt 2 ]; then continue; fi
    echo "== GPQHE_MID8=$v $L run $i"
    GPQHE_HIP_LIB=$PWD/$L GPQHE_MID8=$v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-ntt 2>gpurun_out/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k:(round(v['avg_ms'],4) if isinstance(v,dict) and 'avg_ms' in v else v) for k,v in d.get('kernels',{}).items()})
" || { tail -20 gpurun_out/bench.err; exit 1; }
    done
  done
done
