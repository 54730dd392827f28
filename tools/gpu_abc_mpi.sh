#!/bin/bash
# interleaved A/B of the MPI-level he_mul between library builds / settings on ONE device:
#   tools/gpu_abc_mpi.sh "libA.so" "libB.so MPI_FUSED=0" ...   (library path relative to the repo, then optional VAR=value settings)
set -o pipefail
mkdir -p gpurun_out
specs=("$@")
for round in 1 2 3; do
  for spec in "${specs[@]}"; do
    parts=($spec)
    lib=${parts[0]}
    env GPQHE_HIP_LIB=$PWD/$lib "${parts[@]:1}" python3 tools/mpi_profile.py 2>/dev/null | sed "s|^|$spec: |"
  done
done
