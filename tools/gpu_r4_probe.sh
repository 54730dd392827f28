#!/bin/bash
# interleaved runs of the MPI-level he_mul over library builds / settings on ONE device: "lib VAR=value ..." per spec
set -o pipefail
mkdir -p gpurun_out
specs=("$@")
for round in 1 2 3; do
  for spec in "${specs[@]}"; do
    parts=($spec)
    env MPI_LIB=$PWD/gpqhe_amd/${parts[0]} "${parts[@]:1}" timeout -k 10 300 python3 tools/mpi_profile.py 2>/dev/null | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"'he_mul_per_s': ([0-9.]+)\",l); b=re.search(r\"'bridge_ms_per_batch': ([0-9.]+)\",l)
    ks=re.findall(r\"'(bridge_\w+)': \{'ms_per_batch': ([0-9.]+)\",l)
    print('$spec:', m.group(1), b.group(1), ' '.join('%s=%s' % (k[7:], v) for k, v in ks))
" | tee -a gpurun_out/r4_probe.txt
  done
done
