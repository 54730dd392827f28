#!/bin/bash
# interleaved MPI-level he_mul timing of several library builds on ONE device: tools/gpu_abc_mpi.sh rounds lib...
set -o pipefail
mkdir -p gpurun_out
R=$1; shift
for r in $(seq 1 $R); do
  for L in "$@"; do
    echo "$L: $(GPQHE_HIP_LIB=$PWD/$L python tools/mpi_profile.py 2>/dev/null)"
  done
done | tee gpurun_out/ab_mpi.txt
