#!/bin/bash
# clocks and package power while a workload runs (dev tool): tools/gpu_clocks.sh [bench|ntt|lab]
mkdir -p gpurun_out
case "${1:-bench}" in
  bench) (python bench.py --steps 1500 --warmup 2 --cpu-sample 0 --no-ntt > gpurun_out/clk_bench.json 2>/dev/null) & ;;
  ntt)   (ITERS=3000 python tools/ntt_profile.py > gpurun_out/clk_ntt.txt 2>/dev/null) & ;;
  lab)   (for i in 1 2 3 4 5 6 7 8; do ./tools/bfly_lab; done > gpurun_out/clk_lab.txt 2>/dev/null) & ;;
esac
BP=$!
sleep ${2:-6}
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.4; done
wait $BP
