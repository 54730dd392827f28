#!/bin/bash
# per-kernel breakdown of the MPI-level he_mul (rocprofv3 kernel stats), matrix-core bridge on and off
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1; do
  export GPQ_BRIDGE_VALU=$v
  rm -rf gpurun_out/prof_mpi$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mpi$v -- python3 tools/mpi_profile.py > gpurun_out/prof_mpi$v.txt 2> gpurun_out/prof_mpi$v.err || { tail gpurun_out/prof_mpi$v.err; exit 1; }
  echo "== GPQ_BRIDGE_VALU=$v"; cat gpurun_out/prof_mpi$v.txt
  python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_mpi$v/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'gpq' in r['Name']:
        print("%-70s calls %4s total %9.3f ms avg %8.1f us  %5s%%" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
done
