#!/bin/bash
R=$1; shift
for r in $(seq 1 $R); do for L in "$@"; do
  echo "$L: $(GPQHE_HIP_LIB=$PWD/$L python tools/ntt_profile.py 2>/dev/null | head -1)"
done; done
