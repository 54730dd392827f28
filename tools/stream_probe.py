"""Sweep of gpq_probe_stream (dev tool): which plain-stream configuration this device's memory system serves fastest.
usage: python tools/stream_probe.py [GiB]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from gpqhe_amd import _native
lib = _native.load()
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
nbytes = int(gib * (1 << 30)) // 16 * 16
src = torch.randint(0, 1 << 62, (nbytes // 8,), dtype=torch.int64, device="cuda")
dst = torch.empty_like(src)
st = torch.cuda.current_stream().cuda_stream
def run(kind, blocks, threads, unroll, reps=5):
    for _ in range(2):
        _native.check(lib.gpq_probe_stream(dst.data_ptr(), src.data_ptr(), nbytes, kind, blocks, threads, unroll, st), "probe")
    t = gpqhe_amd.StreamTimer(); t.start()
    for _ in range(reps):
        _native.check(lib.gpq_probe_stream(dst.data_ptr(), src.data_ptr(), nbytes, kind, blocks, threads, unroll, st), "probe")
    t.stop()
    ms = t.elapsed_ms() / reps
    return (2 if kind == 0 else 1) * nbytes / ms / 1e6
for kind, name in ((0, "copy"), (1, "read"), (2, "write")):
    best = (0, None)
    for threads in (256, 512, 1024):
        for per_cu in (1, 2, 4, 8, 16):
            blocks = 256 * per_cu
            if blocks * threads > 256 * 2048 * 4: continue
            for unroll in (1, 2, 4, 8):
                r = run(kind, blocks, threads, unroll)
                if r > best[0]: best = (r, (blocks, threads, unroll))
                print("%s blocks %5d threads %4d unroll %d: %7.1f GB/s" % (name, blocks, threads, unroll, r), flush=True)
    print("BEST %s: %.1f GB/s at blocks/threads/unroll %s" % (name, best[0], best[1]), flush=True)
