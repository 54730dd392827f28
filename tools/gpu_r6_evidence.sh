#!/bin/bash
# round 6: the evidence the bench line and DESIGN.md point at, all from one commit (tools/.head) -- the same session as tools/gpu_r5_evidence.sh without
# the n = 2^17 passes (those kernels did not change): bench line; one-lane rocprofv3 --kernel-trace --stats of the RNS-core command; PMC passes of the
# core (pmc_summary.json: traffic, VALU instructions per he_mul); kernel stats + PMC of the whole-he_mul leg; the issue probe for tools/valu_bound.py
# (rebuilt on this tree's kernel headers: bench.py omits the probe-derived denominators otherwise).
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python3 bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err || { tail -5 gpurun_out/r6_bench.err; exit 1; }
rm -rf gpurun_out/prof_r6 && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r6 -- python3 bench.py --lanes 1 --steps 100 --warmup 2 --cpu-sample 0 --no-ntt > gpurun_out/r6_bench_under_rocprof.json 2> gpurun_out/r6_prof.err || { tail gpurun_out/r6_prof.err; exit 1; }
cp $(find gpurun_out/prof_r6 -name "*kernel_stats.csv" | head -1) gpurun_out/r6_kernel_stats.csv
bash tools/gpu_pmc.sh > gpurun_out/r6_pmc.log 2>&1 || { tail gpurun_out/r6_pmc.log; exit 1; }
cp gpurun_out/pmc_summary.json gpurun_out/r6_pmc_summary.json
export MPI_OVERLAP=0
rm -rf gpurun_out/prof_r6m && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r6m -- python3 tools/mpi_profile.py > gpurun_out/r6_mpi.txt 2> gpurun_out/r6_prof_mpi.err || { tail gpurun_out/r6_prof_mpi.err; exit 1; }
cp $(find gpurun_out/prof_r6m -name "*kernel_stats.csv" | head -1) gpurun_out/r6_mpi_kernel_stats.csv
bash tools/gpu_pmc_mpi.sh > gpurun_out/r6_mpi_pmc.txt 2>&1 || { tail gpurun_out/r6_mpi_pmc.txt; exit 1; }
tools/instr_rate > gpurun_out/r6_instr_rate.txt 2>&1 || exit 1
tools/issue_probe > gpurun_out/r6_issue_probe.txt 2>&1 || exit 1
for w in 4 3; do
  rm -f gpurun_out/.long gpurun_out/.smi
  (tools/issue_probe long $w > gpurun_out/.long) &
  BP=$!
  sleep 0.8
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | sed 's/.*: //' | tr '\n' ' ' >> gpurun_out/.smi; echo >> gpurun_out/.smi; sleep 0.3; done
  wait $BP
  cat gpurun_out/.long >> gpurun_out/r6_issue_probe.txt
  echo "# sclk / package power while it ran (rocm-smi, 0.3 s apart):" >> gpurun_out/r6_issue_probe.txt
  cat gpurun_out/.smi >> gpurun_out/r6_issue_probe.txt
done
rm -f gpurun_out/.long gpurun_out/.smi
rm -rf gpurun_out/prof_r6 gpurun_out/prof_r6m gpurun_out/pmc_* gpurun_out/pmcm_*
tail -12 gpurun_out/r6_issue_probe.txt
tail -24 gpurun_out/r6_mpi_pmc.txt | cut -c1-220
head -8 gpurun_out/r6_kernel_stats.csv | cut -c1-200
