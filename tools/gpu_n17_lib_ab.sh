#!/bin/bash
# alternating PROCESSES on one device: key switch at n = 2^17, 44 limbs, batch 64, one lane, over library builds (args: lib names under gpqhe_amd/)
set -o pipefail
mkdir -p gpurun_out; OUT=gpurun_out/r5_n17_lib_ab.txt; : > $OUT
for r in 1 2 3 4; do
  for L in "$@"; do
    echo "$L round $r: $(N17_LIB=$PWD/gpqhe_amd/$L N17_ITERS=8 timeout -k 10 120 python3 tools/n17_profile.py 2>/dev/null | tail -1)" | tee -a $OUT
  done
done
