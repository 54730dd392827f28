#!/bin/bash
# round 3: BASELINE configs[3]'s own batch (512 ciphertexts) on the one device of the box
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_config3_batch512_gpu.py tests/test_shard_c_gpu.py tests/test_dist_hip_gpu.py -x -q -m gpu > gpurun_out/r3_c3_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3_c3_tests.log
python bench.py --gpus 1 --total-batch 512 --steps 5 --warmup 2 --cpu-sample 0 --no-ntt > gpurun_out/r3_c3_bench512.json 2> gpurun_out/r3_c3_bench512.err; echo "bench rc=$?"; cat gpurun_out/r3_c3_bench512.json
gcc -O2 -std=gnu11 -I include tests/c/shard_host.c -L gpqhe_amd -lgpqhe_hip -Wl,-rpath,$PWD/gpqhe_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/shard_host || exit 1
( time /tmp/shard_host 16 30 45 512 0,0,0,0,0,0,0,0 8 ) > gpurun_out/r3_c3_shard512.txt 2>&1; echo "shard rc=$?"
head -3 gpurun_out/r3_c3_shard512.txt; tail -5 gpurun_out/r3_c3_shard512.txt
