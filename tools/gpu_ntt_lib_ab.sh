#!/bin/bash
# alternating PROCESSES on one device: the standalone NTT pairs of bench.py (ntt_rate) over library builds (args: lib names under gpqhe_amd/)
set -o pipefail
mkdir -p gpurun_out; OUT=gpurun_out/r5_ntt_lib_ab.txt; : > $OUT
for r in 1 2 3; do
  for L in "$@"; do
    python3 - $PWD/gpqhe_amd/$L <<'PY' 2>/dev/null | tee -a $OUT
import sys, os
sys.path.insert(0, os.getcwd())
import torch, gpqhe_amd
from gpqhe_amd import _native
_native.use_variant(sys.argv[1])
import bench
torch.cuda.set_device(0)
res = [bench.ntt_rate(torch, gpqhe_amd, 15, 10, 64), bench.ntt_rate(torch, gpqhe_amd, 16, 30, 64), bench.ntt_rate(torch, gpqhe_amd, 14, 8, 64)]
print(os.path.basename(sys.argv[1]), " ".join("%s %.4f ms %.4f" % (r["shape"][:12], r["ms_per_pair"], r["hbm_frac"]) for r in res))
PY
  done
done
