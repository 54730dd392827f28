#!/bin/bash
# round 3: the evidence bench.py's `valu_issue` and `roofline.traffic` objects point at, all from one commit (tools/.head)
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
tools/instr_rate > gpurun_out/r3_instr_rate.txt 2>&1 || exit 1
tools/issue_probe > gpurun_out/r3_issue_probe.txt 2>&1 || exit 1
cat gpurun_out/r3_issue_probe.txt
bash tools/gpu_pmc.sh > gpurun_out/r3_pmc.log 2>&1 || { tail gpurun_out/r3_pmc.log; exit 1; }
cp gpurun_out/pmc_summary.json gpurun_out/r3_pmc_summary.json
rm -rf gpurun_out/prof_r3 && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3 -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-ntt > gpurun_out/r3_bench_prof.json 2> gpurun_out/r3_prof.err || { tail gpurun_out/r3_prof.err; exit 1; }
cp $(find gpurun_out/prof_r3 -name "*kernel_stats.csv" | head -1) gpurun_out/r3_kernel_stats.csv
bash tools/gpu_pmc_mpi.sh > gpurun_out/r3_pmc_mpi.txt 2>&1 || { tail gpurun_out/r3_pmc_mpi.txt; exit 1; }
tail -30 gpurun_out/r3_pmc_mpi.txt
bash tools/gpu_r3_probe.sh > /dev/null 2>&1
bash tools/gpu_r3_mpi.sh final > gpurun_out/r3_mpi_final.log 2>&1; tail -14 gpurun_out/r3_mpi_final.log
