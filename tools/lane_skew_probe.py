"""Two lanes of whole he_mul (two contexts, two streams, 32 ciphertexts each, n = 2^16, q = 2^850) with DIFFERENT launch-group sizes per lane, so
that the lanes do not run the same kernel at the same time: does a bridge kernel of one lane beside a transform of the other pay more than
lockstep lanes do?  And the same for long batches (256 per call), where lanes drift apart by themselves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from tools.two_lanes_ab import setup, timed


def lanes(parts, iters=6):
    streams = [torch.cuda.Stream() for _ in parts]
    def once():
        for s, (ctx, cts, rlk, outs, W, (dA, dB, dP)) in zip(streams, parts):
            with torch.cuda.stream(s):
                ctx.he_mul(outs[0], outs[1], *cts, rlk[0], rlk[1], W, 850, dA, dB, dP)
    return timed(once, iters)


one = setup(16, 850, 64, 21)
one[0].set_overlap(0)
a, b = setup(16, 850, 32, 22), setup(16, 850, 32, 23)
for p in (a, b):
    p[0].set_overlap(0)
for rnd in range(3):
    res = ["one lane x64: %.3f ms" % lanes([one])]
    for ca, cb in ((32, 32), (32, 16), (32, 11), (32, 8), (16, 11), (16, 16)):
        a[0].set_chunk(ca); b[0].set_chunk(cb)
        res.append("groups %d | %d: %.3f ms" % (ca, cb, lanes([a, b])))
    print("round %d: %s" % (rnd, " ; ".join(res)), flush=True)
