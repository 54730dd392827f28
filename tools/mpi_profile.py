"""rocprofv3 target: a few MPI-level he_mul's (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import he_mul_mpi_rate
ctx = gpqhe_amd.PolyContext(16, 45)
print(he_mul_mpi_rate(torch, gpqhe_amd, ctx, 16, iters=6))
