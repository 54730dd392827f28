#!/bin/bash
gcc -O1 -std=gnu11 -I include tests/c/mpi_host.c -L gpqhe_amd -lgpqhe_hip -lgpqhe_hip_ctx -l:libgcrypt.so.20 -Wl,-rpath,$PWD/gpqhe_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/mpi_host || exit 1
mkdir -p gpurun_out; : > gpurun_out/residentfuzz_long.txt
for seed in 101 102 103 104 105 106; do
  for spec in "16 850 50 500" "15 590 40 600" "12 109 20 6000"; do
    echo "== $spec seed $seed" >> gpurun_out/residentfuzz_long.txt
    timeout -k 10 900 /tmp/mpi_host residentfuzz $spec $seed >> gpurun_out/residentfuzz_long.txt 2>&1 || { echo "FAILED $spec $seed"; tail -3 gpurun_out/residentfuzz_long.txt; exit 1; }
  done
done
grep -c "residentfuzz ok" gpurun_out/residentfuzz_long.txt; tail -2 gpurun_out/residentfuzz_long.txt
