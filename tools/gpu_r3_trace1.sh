#!/bin/bash
# round 3: kernel trace of single-ciphertext he_mul calls through the reference signature (durations and gaps of one call)
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
gcc -O1 -std=gnu11 -I include tests/c/mpi_host.c -L gpqhe_amd -lgpqhe_hip -lgpqhe_hip_ctx -l:libgcrypt.so.20 -Wl,-rpath,$PWD/gpqhe_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/mpi_host || exit 1
rm -rf gpurun_out/trace1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace1 -- /tmp/mpi_host hemultime 16 850 > gpurun_out/trace1.txt 2>&1 || { tail gpurun_out/trace1.txt; exit 1; }
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/trace1/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the last general he_mul: sequences start with bridge_decompose_mfma; take the sequence number -60 (before squaring/rescale parts)
starts=[i for i,r in enumerate(rows) if 'bridge_decompose_mfma' in r['Kernel_Name'] and (i==0 or 'bridge_decompose_mfma' not in rows[i-1]['Kernel_Name'])]
# he_mul has two decompose launches (inputs, d2): pick a start whose next decompose is within 12 kernels
cands=[s for s in starts if any('tensor_mid8' in rows[j]['Kernel_Name'] for j in range(s,min(s+4,len(rows))))]
s=cands[len(cands)//2]
e=s
t0=int(rows[s]['Start_Timestamp'])
prev_end=t0
tot=0
out=[]
for j in range(s,min(s+40,len(rows))):
    r=rows[j]
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if j>s and 'bridge_decompose_mfma' in r['Kernel_Name'] and any('tensor_mid8' in rows[k]['Kernel_Name'] for k in range(j,min(j+4,len(rows)))): break
    out.append("%-70s start %8.1f us  dur %7.1f us  gap before %6.1f us" % (r['Kernel_Name'][:70].replace('void gpq::',''), (st-t0)/1e3, (en-st)/1e3, (st-prev_end)/1e3))
    prev_end=en; tot+=en-st
print("\n".join(out))
print("kernels %d  busy %.1f us  span %.1f us" % (len(out), tot/1e3, (prev_end-t0)/1e3))
PY
