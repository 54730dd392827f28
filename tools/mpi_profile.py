"""rocprofv3 target: a few MPI-level he_mul's (dev tool).  MPI_BATCH (default 64) ciphertexts per call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from gpqhe_amd import _native
if os.environ.get("MPI_LIB"):                      # A/B of library builds (tools/gpu_r4_probe.sh): explicit, before the first load
    _native.use_variant(os.environ["MPI_LIB"])
from bench import he_mul_mpi_rate
ctx = gpqhe_amd.PolyContext(16, 45)
if os.environ.get("GPQ_BRIDGE_VALU") == "1":      # tool-side switch (tools/gpu_prof_mpi.sh): the library itself reads no environment
    ctx.set_bridge_mfma(False)
if os.environ.get("MPI_FUSED"):
    ctx.set_fused_tail(os.environ["MPI_FUSED"] == "1")
if os.environ.get("MPI_PRESCALE"):
    ctx.set_prescale(int(os.environ["MPI_PRESCALE"]))
if os.environ.get("MPI_STREAM"):
    ctx.set_stream_bridge(os.environ["MPI_STREAM"] == "1")
    ctx.set_lazy_decompose(os.environ["MPI_STREAM"] == "1")
if os.environ.get("MPI_LAZY"):
    ctx.set_lazy_decompose(os.environ["MPI_LAZY"] == "1")
# MPI_OVERLAP=0: one lane (what a kernel trace / PMC pass should see: nothing running beside the kernel it times)
print(he_mul_mpi_rate(torch, gpqhe_amd, ctx, int(os.environ.get("MPI_BATCH", "64")), iters=int(os.environ.get("MPI_ITERS", "6")),
                      two_lanes=os.environ.get("MPI_OVERLAP", "1") != "0"))
