#!/bin/bash
# round 4: the evidence the bench line and DESIGN.md point at, all from one commit (tools/.head):
#   bench line; rocprofv3 --kernel-trace --stats of the same command; PMC passes of the RNS core (pmc_summary.json: traffic, VALU
#   instructions per he_mul); kernel stats and PMC of the whole-he_mul leg (the streaming bridge kernels).
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python3 bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err || { tail -5 gpurun_out/r4_bench.err; exit 1; }
rm -rf gpurun_out/prof_r4 && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r4 -- python3 bench.py --steps 100 --warmup 2 --cpu-sample 0 --no-ntt > gpurun_out/r4_bench_under_rocprof.json 2> gpurun_out/r4_prof.err || { tail gpurun_out/r4_prof.err; exit 1; }
cp $(find gpurun_out/prof_r4 -name "*kernel_stats.csv" | head -1) gpurun_out/r4_kernel_stats.csv
bash tools/gpu_pmc.sh > gpurun_out/r4_pmc.log 2>&1 || { tail gpurun_out/r4_pmc.log; exit 1; }
cp gpurun_out/pmc_summary.json gpurun_out/r4_pmc_summary.json
export MPI_OVERLAP=0   # kernel stats and counters of the whole-he_mul leg on ONE lane (gpq_set_overlap(ctx, 0)): with two lanes the durations are those of kernels sharing the chip
rm -rf gpurun_out/prof_r4m && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r4m -- python3 tools/mpi_profile.py > gpurun_out/r4_mpi.txt 2> gpurun_out/r4_prof_mpi.err || { tail gpurun_out/r4_prof_mpi.err; exit 1; }
cp $(find gpurun_out/prof_r4m -name "*kernel_stats.csv" | head -1) gpurun_out/r4_mpi_kernel_stats.csv
bash tools/gpu_pmc_mpi.sh > gpurun_out/r4_mpi_pmc.txt 2>&1 || { tail gpurun_out/r4_mpi_pmc.txt; exit 1; }
tail -32 gpurun_out/r4_mpi_pmc.txt | cut -c1-220
head -12 gpurun_out/r4_mpi_kernel_stats.csv | cut -c1-200
