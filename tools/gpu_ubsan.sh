#!/bin/bash
# Host side of the library under UndefinedBehaviorSanitizer (device code is not instrumented: no GPU sanitizer on this pool),
# then the GPU suite through it, then the two C hosts (real libgcrypt MPIs through the MPI-typed surface, the single-limb
# drop-in symbols) against the same instrumented library.
set -o pipefail
mkdir -p gpurun_out
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
# (built here unless the caller already built it in the container: the .so travels with the snapshot and saves GPU minutes)
[ -f gpqhe_amd/libgpqhe_hip_UBSAN.so ] || (cd gpqhe_amd/csrc && hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Xarch_host -fsanitize=undefined \
   -shared engine.hip bridge.hip dropin.hip mpi_shim.hip affinity.hip -ldl -pthread -o ../libgpqhe_hip_UBSAN.so) || exit 1
echo "instrumented library built" | tee gpurun_out/ubsan.txt
LD_PRELOAD=$RT UBSAN_OPTIONS=print_stacktrace=1 \
  timeout -k 10 1000 python -m pytest tests -m gpu -q --variant $PWD/gpqhe_amd/libgpqhe_hip_UBSAN.so -k "not c_host and not mpi_surface and not dropin and not reference_signature" >> gpurun_out/ubsan.txt 2>&1
tail -3 gpurun_out/ubsan.txt
D=$(mktemp -d) && cp gpqhe_amd/libgpqhe_hip_UBSAN.so $D/libgpqhe_hip.so && cp gpqhe_amd/libgpqhe_hip_ctx.so $D/
gcc -O1 -std=gnu11 -I include tests/c/mpi_host.c -L $D -lgpqhe_hip -lgpqhe_hip_ctx -l:libgcrypt.so.20 -Wl,-rpath,$D -Wl,-rpath,/opt/rocm/lib -Wl,--unresolved-symbols=ignore-in-shared-libs -o $D/mpi_host || exit 1
gcc -O1 -std=gnu11 -I include tests/c/dropin_host.c -L $D -lgpqhe_hip -Wl,-rpath,$D -Wl,-rpath,/opt/rocm/lib -Wl,--unresolved-symbols=ignore-in-shared-libs -o $D/dropin_host || exit 1
gcc -O1 -std=gnu11 -I include tests/c/shard_host.c -L $D -lgpqhe_hip -pthread -Wl,-rpath,$D -Wl,-rpath,/opt/rocm/lib -Wl,--unresolved-symbols=ignore-in-shared-libs -o $D/shard_host || exit 1
for cmd in "mpi_host polymul" "mpi_host polymulodd" "mpi_host crt" "mpi_host polymulmono 13" "mpi_host keygen 7 120" "mpi_host ctxcheck 7 61 1073741824" \
           "mpi_host ctxcheck 16 850 1125899906842624" "mpi_host hemultime 16 850" "dropin_host 13 3 1" "dropin_host 16 2 5" "shard_host 13 3 4 7 0,0,0" "shard_host 13 3 4 7 0,0,0 0 2"; do
  echo "== $cmd" >> gpurun_out/ubsan.txt
  LD_PRELOAD=$RT UBSAN_OPTIONS=print_stacktrace=1 timeout -k 10 300 $D/$cmd 2>&1 | tail -3 | cut -c1-200 >> gpurun_out/ubsan.txt || echo "FAILED: $cmd" | tee -a gpurun_out/ubsan.txt
done
tail -12 gpurun_out/ubsan.txt
echo "runtime errors reported: $(grep -c 'runtime error' gpurun_out/ubsan.txt)" | tee -a gpurun_out/ubsan.txt
rm -f gpqhe_amd/libgpqhe_hip_UBSAN.so
