#!/bin/bash
# wall time of the reference-signature he_mul / he_rescale (tests/c/mpi_host.c hemultime), with the per-stage breakdown
set -o pipefail
mkdir -p gpurun_out
gcc -O1 -std=gnu11 -I include tests/c/mpi_host.c -L gpqhe_amd -lgpqhe_hip -lgpqhe_hip_ctx -l:libgcrypt.so.20 -Wl,-rpath,$PWD/gpqhe_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/mpi_host || exit 1
nproc
for i in 1 2 3; do timeout -k 10 120 /tmp/mpi_host hemultime 16 850; done 2>&1 | tee gpurun_out/hemultime.txt
