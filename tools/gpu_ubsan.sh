#!/bin/bash
# Host side of the library under UndefinedBehaviorSanitizer (device code is not instrumented: no GPU sanitizer on this pool),
# then the GPU suite through it.  The C hosts are skipped (they link the regular build).
set -o pipefail
mkdir -p gpurun_out
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
(cd gpqhe_amd/csrc && hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Xarch_host -fsanitize=undefined \
   -shared engine.hip bridge.hip dropin.hip mpi_shim.hip -ldl -pthread -o ../libgpqhe_hip_UBSAN.so) || exit 1
LD_PRELOAD=$RT UBSAN_OPTIONS=print_stacktrace=1 GPQHE_HIP_LIB=$PWD/gpqhe_amd/libgpqhe_hip_UBSAN.so \
  timeout -k 10 1000 python -m pytest tests -m gpu -q -k "not c_host and not mpi_surface and not dropin" > gpurun_out/ubsan.txt 2>&1
tail -3 gpurun_out/ubsan.txt
echo "runtime errors reported: $(grep -c 'runtime error' gpurun_out/ubsan.txt)"
