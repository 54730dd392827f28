#!/bin/bash
# rocprofv3 kernel statistics of the reference-signature walk (tests/c/mpi_host.c hemultime: cold and chained calls, additive calls, he_inv's sequence, the ladder)
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
gcc -O1 -std=gnu11 -I include tests/c/mpi_host.c -L gpqhe_amd -lgpqhe_hip -lgpqhe_hip_ctx -l:libgcrypt.so.20 -Wl,-rpath,$PWD/gpqhe_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/mpi_host || exit 1
rm -rf gpurun_out/prof_hemultime
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hemultime -- /tmp/mpi_host hemultime 16 850 > gpurun_out/prof_hemultime.txt 2>&1 || { tail gpurun_out/prof_hemultime.txt; exit 1; }
f=$(ls gpurun_out/prof_hemultime/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/r3_hemultime_kernel_stats.csv
head -40 gpurun_out/r3_hemultime_kernel_stats.csv | cut -c1-200
