#!/bin/bash
# round 4, first GPU contact of bridge_stream.hpp: parity of the streaming kernels, then an interleaved A/B against round 3's kernels
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_stream_bridge_gpu.py -x -q > gpurun_out/r4_first_tests.txt 2>&1
rc=$?
tail -15 gpurun_out/r4_first_tests.txt
[ $rc -ne 0 ] && exit $rc
for round in 1 2 3; do
  for st in 0 1; do
    MPI_STREAM=$st timeout -k 10 300 python3 tools/mpi_profile.py 2>/dev/null | sed "s|^|MPI_STREAM=$st: |" | tee -a gpurun_out/r4_first_ab.txt || exit 1
  done
done
