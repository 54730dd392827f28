#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_mpi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mpi -- python3 tools/mpi_profile.py > gpurun_out/prof_mpi.txt 2> gpurun_out/prof_mpi.err || { tail gpurun_out/prof_mpi.err; exit 1; }
cat gpurun_out/prof_mpi.txt
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_mpi/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'gpq' in r['Name']:
        print("%-70s calls %4s total %9.3f ms avg %8.1f us  %5s%%" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
