#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
cd tools && hipcc --offload-arch=gfx950 -O3 -I../gpqhe_amd/csrc bfly_lab.hip -o bfly_lab && ./bfly_lab | tee ../gpurun_out/bfly_lab.txt
cd .. && rm -rf gpurun_out/pmc_lab && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_lab -- ./tools/bfly_lab > /dev/null 2> gpurun_out/pmc_lab.err; ls gpurun_out/pmc_lab/*/ | head
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_lab/*/*counter_collection.csv')
if f:
    agg=collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        key=(r['Kernel_Name'][:40], r['Grid_Size'])
        agg.setdefault(key, collections.defaultdict(float))[r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items(): print(k, {a:int(b) for a,b in v.items()})
PY
