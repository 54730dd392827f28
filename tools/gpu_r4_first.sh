#!/bin/bash
# round 4: parity of the streaming bridge kernels, then an interleaved A/B against round 3's kernels
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_stream_bridge_gpu.py tests/test_bridge_gpu.py -x -q > gpurun_out/r4_first_tests.txt 2>&1
rc=$?
tail -5 gpurun_out/r4_first_tests.txt
[ $rc -ne 0 ] && exit $rc
for round in 1 2 3; do
  for st in 0 1; do
    MPI_STREAM=$st timeout -k 10 300 python3 tools/mpi_profile.py 2>/dev/null | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"'he_mul_per_s': ([0-9.]+)\",l); b=re.search(r\"'bridge_ms_per_batch': ([0-9.]+)\",l)
    ks=re.findall(r\"'(bridge_\w+)': \{'ms_per_batch': ([0-9.]+)\",l)
    print('MPI_STREAM=$st', m.group(1), b.group(1), ks)
" | tee -a gpurun_out/r4_first_ab.txt || exit 1
  done
done
