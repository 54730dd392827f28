#!/bin/bash
# PMC passes over a short bench run (kernel-trace + counters only; one counter group per pass).
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
BENCH="python3 bench.py --lanes 1 --steps 1 --warmup 1 --batch 16 --cpu-sample 0 --no-ntt"
pass() { # name, counters...
  local name=$1; shift
  rm -rf gpurun_out/pmc_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- $BENCH > gpurun_out/pmc_$name.json 2> gpurun_out/pmc_$name.err || { tail -5 gpurun_out/pmc_$name.err; return 1; }
}
pass fetch FETCH_SIZE && pass write WRITE_SIZE && pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS && pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH GRBM_GUI_ACTIVE && pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - <<'PY'
import csv,glob,collections,json
out={}
for f in sorted(glob.glob('gpurun_out/pmc_*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name']
        if 'gpq::' not in name: continue
        name=name.replace('void ','').split('(')[0]
        d=out.setdefault(name, collections.defaultdict(float))
        d[r['Counter_Name']]+=float(r['Counter_Value'])
        d['_n_'+r['Counter_Name']]+=1
res={}
for k,v in out.items():
    res[k]={c:(v[c]/v['_n_'+c]) for c in v if not c.startswith('_n_')}
    res[k]['launches']=max(v[c] for c in v if c.startswith('_n_'))
res["_chunk"]=16
import os
res["_head"]=open("tools/.head").read().strip() if os.path.exists("tools/.head") else None
res["_command"]="python3 bench.py --lanes 1 --steps 1 --warmup 1 --batch 16 --cpu-sample 0 --no-ntt (one rocprofv3 --pmc pass per counter group)"
json.dump(res, open('gpurun_out/pmc_summary.json','w'), indent=1)
for k,v in res.items():
    if k.startswith('_'): continue
    print(k); print('   ', {a:(round(b,1) if b<1e6 else int(b)) for a,b in sorted(v.items())})
PY
