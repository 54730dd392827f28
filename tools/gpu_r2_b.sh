#!/bin/bash
# round 2, second call: GPU suite, reference-signature latency, standalone-NTT kernel profile (stats + PMC)
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.txt || exit 1
gcc -O1 -std=gnu11 -I include tests/c/mpi_host.c -L gpqhe_amd -lgpqhe_hip -l:libgcrypt.so.20 -Wl,-rpath,$PWD/gpqhe_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/mpi_host || exit 1
for i in 1 2 3; do timeout -k 10 120 /tmp/mpi_host hemultime 16 850; done 2>&1 | tee gpurun_out/hemultime.txt
rm -rf gpurun_out/prof_ntt && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ntt -- python3 tools/ntt_profile.py > gpurun_out/ntt_profile.txt 2> gpurun_out/prof_ntt.err || { tail -20 gpurun_out/prof_ntt.err; exit 1; }
cat gpurun_out/ntt_profile.txt
cp $(find gpurun_out/prof_ntt -name "*kernel_stats.csv" | head -1) gpurun_out/ntt_kernel_stats.csv && grep gpq gpurun_out/ntt_kernel_stats.csv | cut -c1-200
pass() { # name, counters...
  local name=$1; shift
  rm -rf gpurun_out/pmcn_$name
  ITERS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcn_$name -- python3 tools/ntt_profile.py > gpurun_out/pmcn_$name.txt 2> gpurun_out/pmcn_$name.err || { tail -5 gpurun_out/pmcn_$name.err; return 1; }
}
pass fetch FETCH_SIZE && pass write WRITE_SIZE && pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS && pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv,glob,collections,json,subprocess
out={}
for f in sorted(glob.glob('gpurun_out/pmcn_*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name']
        if 'gpq::' not in name: continue
        name=name.replace('void ','').split('(')[0]+' grid='+r.get('Grid_Size','?')
        d=out.setdefault(name, collections.defaultdict(float))
        d[r['Counter_Name']]+=float(r['Counter_Value'])
        d['_n_'+r['Counter_Name']]+=1
res={}
for k,v in out.items():
    res[k]={c:(v[c]/v['_n_'+c]) for c in v if not c.startswith('_n_')}
    res[k]['launches']=max(v[c] for c in v if c.startswith('_n_'))
res["_what"]="tools/ntt_profile.py, ITERS=1: per-launch averages; FETCH_SIZE/WRITE_SIZE in KiB"
json.dump(res, open('gpurun_out/ntt_pmc_summary.json','w'), indent=1)
for k,v in res.items():
    if k.startswith('_'): continue
    print(k); print('   ', {a:(round(b,1) if b<1e6 else int(b)) for a,b in sorted(v.items())})
PY
