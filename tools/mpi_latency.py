"""single-ciphertext latency of the MPI-level he_mul on device slabs (dev tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import he_mul_mpi_rate
ctx = gpqhe_amd.PolyContext(16, 45)
for b in (1, 2, 4, 16):
    print(b, he_mul_mpi_rate(torch, gpqhe_amd, ctx, b, iters=10))
