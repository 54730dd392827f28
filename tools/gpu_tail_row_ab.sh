#!/bin/bash
# round 5: rows of the tail kernel's tables of multiples padded to 17 words in LDS (product) against 16 (make variant NAME=row16 DEFS=-DGPQ_TAIL_ROW=16):
# alternating PROCESSES on one device (the tail is bimodal per process: several of each), one lane; then the LDS bank-conflict counter of both builds.
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp MPI_OVERLAP=0 MPI_ITERS=8
OUT=gpurun_out/r5_tail_row_ab.txt; : > $OUT
for r in 1 2 3 4 5; do
  for L in gpqhe_amd/libgpqhe_hip.so gpqhe_amd/libgpqhe_hip_row16.so; do
    MPI_LIB=$PWD/$L timeout -k 10 120 python3 tools/mpi_profile.py 2>/dev/null | python3 -c "
import sys, ast
d = ast.literal_eval(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('$L round $r: he_mul/s one lane %.1f  tail %.4f ms  crt_decompose %.4f ms  decompose %.4f ms  bridge %.3f ms' % (d['he_mul_per_s'], k['bridge_tail_stream']['ms_per_batch'], k['bridge_crt_decompose']['ms_per_batch'], k['bridge_decompose']['ms_per_batch'], d['bridge_ms_per_batch']))" | tee -a $OUT
  done
done
for L in gpqhe_amd/libgpqhe_hip.so gpqhe_amd/libgpqhe_hip_row16.so; do
  rm -rf gpurun_out/pmc_row
  MPI_LIB=$PWD/$L MPI_ITERS=2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_row -- python3 tools/mpi_profile.py > /dev/null 2> gpurun_out/pmc_row.err || { tail -3 gpurun_out/pmc_row.err; continue; }
  python3 - $L <<'PY' | tee -a $OUT
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/pmc_row/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'bridge_tail_stream' in r['Kernel_Name'] or 'bridge_crt_decompose' in r['Kernel_Name']:
            k = r['Kernel_Name'].split('<')[0].replace('void ', '')
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); acc[k]['_n_' + r['Counter_Name']] += 1
for k, v in acc.items():
    print(sys.argv[1], k, {c: round(v[c] / v['_n_' + c]) for c in v if not c.startswith('_n_')})
PY
done
