#!/bin/bash
# round 2: suite + smoke + bench + rocprofv3 kernel stats, then the PMC passes, then a marker trace of one MPI-level he_mul
set -o pipefail
bash tools/gpu_second.sh || exit 1
bash tools/gpu_pmc.sh > gpurun_out/pmc.txt 2>&1 || { tail -20 gpurun_out/pmc.txt; exit 1; }
tail -30 gpurun_out/pmc.txt | cut -c1-300
rm -rf gpurun_out/prof_marker && timeout -k 10 300 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d gpurun_out/prof_marker -- python3 tools/mpi_profile.py > gpurun_out/marker.txt 2> gpurun_out/marker.err || { tail -20 gpurun_out/marker.err; exit 1; }
find gpurun_out/prof_marker -name "*marker*" | head; for f in $(find gpurun_out/prof_marker -name "*marker_api_stats.csv" -o -name "*marker_api_trace.csv" | head -2); do echo "== $f"; head -12 $f | cut -c1-250; done
