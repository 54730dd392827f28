"""Rate of a plain device-to-device copy (bytes read + bytes written per second) at a few sizes: the yardstick the HBM-bound kernels are held against."""
import torch
for mb in (256, 1024, 4096):
    a = torch.empty(mb * 1024 * 1024 // 8, dtype=torch.int64, device="cuda").random_()
    b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("copy of %5d MiB: %.3f ms, %.0f GB/s read+write" % (mb, ms, 2 * a.numel() * 8 / ms / 1e6), flush=True)

# read-only and write-only streams for comparison (a reduction and a fill)
a = torch.empty(2048 * 1024 * 1024 // 8, dtype=torch.int64, device="cuda").random_()
for name, fn, factor in (("read-only (int64 sum)", lambda: a.sum(), 1), ("write-only (fill_)", lambda: a.fill_(7), 1)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%s of 2048 MiB: %.3f ms, %.0f GB/s" % (name, ms, factor * a.numel() * 8 / ms / 1e6), flush=True)
