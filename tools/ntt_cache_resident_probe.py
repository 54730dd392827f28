import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd, bench
torch.cuda.set_device(0)
for logn, dim in ((15, 10), (14, 8)):
    for batch in (2, 4, 8, 16, 32, 64, 256):
        r = bench.ntt_rate(torch, gpqhe_amd, logn, dim, batch, iters=40)
        mb = 8 * (1 << logn) * dim * batch / 1e6
        print("n=2^%d %d limbs batch %3d (%6.1f MB slab): %.4f ms per pair, %.4f of 8 TB/s algorithmic, %.1f us per launch group of 5 kernels" % (logn, dim, batch, mb, r["ms_per_pair"], r["hbm_frac"], r["ms_per_pair"] * 1e3))
