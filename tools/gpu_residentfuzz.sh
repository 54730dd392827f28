#!/bin/bash
# long random walk over the MPI-typed calls, resident polynomials vs fresh uploads (tests/c/mpi_host.c residentfuzz), several seeds and shapes
set -o pipefail
mkdir -p gpurun_out
gcc -O1 -std=gnu11 -I include tests/c/mpi_host.c -L gpqhe_amd -lgpqhe_hip -lgpqhe_hip_ctx -l:libgcrypt.so.20 -Wl,-rpath,$PWD/gpqhe_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/mpi_host || exit 1
: > gpurun_out/residentfuzz.txt
for spec in "12 109 20 3000" "12 109 30 3000" "13 200 25 2000" "13 218 40 2000" "14 300 30 1000" "16 850 50 150"; do
  for seed in 11 12 13 14; do
    echo "== $spec seed $seed" >> gpurun_out/residentfuzz.txt
    timeout -k 10 600 /tmp/mpi_host residentfuzz $spec $seed >> gpurun_out/residentfuzz.txt 2>&1 || { echo "FAILED: $spec $seed" | tee -a gpurun_out/residentfuzz.txt; tail -5 gpurun_out/residentfuzz.txt; exit 1; }
  done
done
grep -c "residentfuzz ok" gpurun_out/residentfuzz.txt
tail -4 gpurun_out/residentfuzz.txt
