#!/bin/bash
# round 5: the evidence the bench line and DESIGN.md point at, all from one commit (tools/.head):
#   bench line; rocprofv3 --kernel-trace --stats of the RNS-core command on ONE lane (--lanes 1: durations of kernels with nothing running beside
#   them); PMC passes of the RNS core (pmc_summary.json: traffic, VALU instructions per he_mul); kernel stats and PMC of the
#   whole-he_mul leg and of the n = 2^17 key switch, one lane each; the issue probe (tools/instr_rate, tools/issue_probe + rocm-smi) for
#   tools/valu_bound.py.
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python3 bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err || { tail -5 gpurun_out/r5_bench.err; exit 1; }
rm -rf gpurun_out/prof_r5 && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r5 -- python3 bench.py --lanes 1 --steps 100 --warmup 2 --cpu-sample 0 --no-ntt > gpurun_out/r5_bench_under_rocprof.json 2> gpurun_out/r5_prof.err || { tail gpurun_out/r5_prof.err; exit 1; }
cp $(find gpurun_out/prof_r5 -name "*kernel_stats.csv" | head -1) gpurun_out/r5_kernel_stats.csv
bash tools/gpu_pmc.sh > gpurun_out/r5_pmc.log 2>&1 || { tail gpurun_out/r5_pmc.log; exit 1; }
cp gpurun_out/pmc_summary.json gpurun_out/r5_pmc_summary.json
export MPI_OVERLAP=0
rm -rf gpurun_out/prof_r5m && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r5m -- python3 tools/mpi_profile.py > gpurun_out/r5_mpi.txt 2> gpurun_out/r5_prof_mpi.err || { tail gpurun_out/r5_prof_mpi.err; exit 1; }
cp $(find gpurun_out/prof_r5m -name "*kernel_stats.csv" | head -1) gpurun_out/r5_mpi_kernel_stats.csv
bash tools/gpu_pmc_mpi.sh > gpurun_out/r5_mpi_pmc.txt 2>&1 || { tail gpurun_out/r5_mpi_pmc.txt; exit 1; }
bash tools/gpu_n17.sh r5_n17 > gpurun_out/r5_n17_log.txt 2>&1 || { tail gpurun_out/r5_n17_log.txt; exit 1; }
N17_WHAT=he_swk bash tools/gpu_n17.sh r5_swk17 > gpurun_out/r5_swk17_log.txt 2>&1 || { tail gpurun_out/r5_swk17_log.txt; exit 1; }
# the issue probe: single-instruction rates, the library's butterfly mix with no memory traffic, then the same mix for a few seconds with rocm-smi sampling
tools/instr_rate > gpurun_out/r5_instr_rate.txt 2>&1 || exit 1
tools/issue_probe > gpurun_out/r5_issue_probe.txt 2>&1 || exit 1
for w in 4 3; do
  rm -f gpurun_out/.long gpurun_out/.smi
  (tools/issue_probe long $w > gpurun_out/.long) &
  BP=$!
  sleep 0.8
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | sed 's/.*: //' | tr '\n' ' ' >> gpurun_out/.smi; echo >> gpurun_out/.smi; sleep 0.3; done
  wait $BP
  cat gpurun_out/.long >> gpurun_out/r5_issue_probe.txt
  echo "# sclk / package power while it ran (rocm-smi, 0.3 s apart):" >> gpurun_out/r5_issue_probe.txt
  cat gpurun_out/.smi >> gpurun_out/r5_issue_probe.txt
done
rm -f gpurun_out/.long gpurun_out/.smi
tail -12 gpurun_out/r5_issue_probe.txt
tail -24 gpurun_out/r5_mpi_pmc.txt | cut -c1-220
head -8 gpurun_out/r5_kernel_stats.csv | cut -c1-200
