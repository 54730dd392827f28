#!/bin/bash
# round 5: is the tail kernel's slow mode (1.31-1.36 ms per 64 on some devices, 1.19-1.25 on others) the 512 KiB limb stride after all?  Classify the device with
# one run of the whole-he_mul leg, then walk the tail's read pattern with padded row strides on the SAME device.
set -o pipefail
mkdir -p gpurun_out; export MPI_OVERLAP=0
T=$(timeout -k 10 200 python3 tools/mpi_profile.py 2>/dev/null | python3 -c "
import sys, ast
d = ast.literal_eval(sys.stdin.read().strip().splitlines()[-1])
print(d['kernels']['bridge_tail_stream']['ms_per_batch'])")
echo "tail_stream ms per 64 on this device: $T" | tee gpurun_out/r5_slowmode_probe.txt
timeout -k 10 200 tools/stride_probe 75 64 2>&1 | grep "round 2" | tee -a gpurun_out/r5_slowmode_probe.txt
