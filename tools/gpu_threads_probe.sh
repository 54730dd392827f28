#!/bin/bash
gcc -O1 -std=gnu11 -I include tests/c/mpi_host.c -L gpqhe_amd -lgpqhe_hip -lgpqhe_hip_ctx -l:libgcrypt.so.20 -Wl,-rpath,$PWD/gpqhe_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/mpi_host || exit 1
for t in 16 32 8 16 32; do echo "== threads $t"; timeout -k 10 120 /tmp/mpi_host hemultime 16 850 $t 2>&1 | grep -E "conversion threads|calls each|chained \(|he_inv" ; done
