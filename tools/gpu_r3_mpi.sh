#!/bin/bash
# round 3: per-kernel breakdown (rocprofv3 kernel stats) + PMC of the MPI-level he_mul at batch 64 -> gpurun_out/r3_mpi_*
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-a}
rm -rf gpurun_out/prof_mpi_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mpi_$tag -- python3 tools/mpi_profile.py > gpurun_out/r3_mpi_$tag.txt 2> gpurun_out/r3_mpi_$tag.err || { tail gpurun_out/r3_mpi_$tag.err; exit 1; }
cat gpurun_out/r3_mpi_$tag.txt
cp gpurun_out/prof_mpi_$tag/*/*kernel_stats.csv gpurun_out/r3_mpi_${tag}_kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/r3_mpi_${tag}_kernel_stats.csv')):
    if 'gpq' in r['Name']:
        print("%-86s calls %4s total %9.3f ms avg %8.1f us  %5s%%" % (r['Name'][:86], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
