#!/bin/bash
# round 5: kernel stats + PMC of the n = 2^17 kernels (BASELINE configs[4] shape), one lane.  N17_WHAT / N17_LIB pass through (tools/n17_profile.py).
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-n17}
rm -rf gpurun_out/prof_$TAG && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 tools/n17_profile.py > gpurun_out/${TAG}.txt 2> gpurun_out/${TAG}.err || { tail gpurun_out/${TAG}.err; exit 1; }
cp $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
pass() { local name=$1; shift
  rm -rf gpurun_out/pmc17_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc17_$name -- python3 tools/n17_profile.py > gpurun_out/pmc17_$name.txt 2> gpurun_out/pmc17_$name.err || { tail -5 gpurun_out/pmc17_$name.err; return 1; }
}
export N17_ITERS=2
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS && pass sq2 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE && pass mem FETCH_SIZE && pass memw WRITE_SIZE
python3 - $TAG <<'PY'
import csv,glob,collections,sys
out={}
for f in sorted(glob.glob('gpurun_out/pmc17_*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name']
        if 'gpq::' not in name: continue
        name=name.replace('void ','').split('(')[0]
        d=out.setdefault(name, collections.defaultdict(float))
        d[r['Counter_Name']]+=float(r['Counter_Value']); d['_n_'+r['Counter_Name']]+=1
lines=[]
for k,v in out.items():
    res={c:(v[c]/v['_n_'+c]) for c in v if not c.startswith('_n_')}
    if res.get('SQ_WAVE_CYCLES',0) < 1e6: continue
    wc=res['SQ_WAVE_CYCLES']
    lines.append(k[:70])
    lines.append('   valu/SIMD %.0f  lds_insts/SIMD %.0f  gui/8 %.0f  wait_any %.2f  wait_inst %.2f  active_valu %.2f  active_any %.2f  active_lds %.2f  lds_wait %.2f  bankconf %.0f  MB %.0f' % (
        res['SQ_INSTS_VALU']/1024, res.get('SQ_INSTS_LDS',0)/1024, res.get('GRBM_GUI_ACTIVE',0)/8, res['SQ_WAIT_ANY']/wc, res['SQ_WAIT_INST_ANY']/wc, res['SQ_ACTIVE_INST_VALU']/wc, res.get('SQ_ACTIVE_INST_ANY',0)/wc, res.get('SQ_ACTIVE_INST_LDS',0)/wc, res.get('SQ_WAIT_INST_LDS',0)/wc, res.get('SQ_LDS_BANK_CONFLICT',0), (2*res.get('FETCH_SIZE',0)+res.get('WRITE_SIZE',0))*1024/1e6))
open('gpurun_out/%s_pmc.txt' % sys.argv[1],'w').write("\n".join(lines)+"\n")
print("\n".join(lines))
PY
cut -c1-170 gpurun_out/${TAG}_kernel_stats.csv | head -14
