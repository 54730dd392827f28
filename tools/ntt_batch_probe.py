import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import ntt_rate
for logn, dim, batch in ((16, 30, 256), (16, 30, 512), (16, 30, 1024), (15, 10, 512), (15, 10, 2048), (15, 10, 8192)):
    print(ntt_rate(torch, gpqhe_amd, logn, dim, batch), flush=True)
