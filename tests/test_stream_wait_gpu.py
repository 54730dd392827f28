"""gpq_stream_wait (include/gpqhe_hip.h): stream-to-stream ordering without blocking the host -- what a plain-C host pipelines uploads, the two stages
and downloads with (tests/c/shard_host.c `pipe`, bench.py `with_host_scatter.pipelined`).  Here directly: a transform queued on stream A, a copy of
its output queued on stream B behind gpq_stream_wait(B, A); the copy must hold the TRANSFORMED words (src/ntt.c:37-52 per limb), and work queued on A
after the wait must not be waited for."""
import ctypes as C

import numpy as np
import pytest
import torch

from gpqhe_amd import _native, to_device, to_host

pytestmark = pytest.mark.gpu


def test_a_copy_on_another_stream_waits_for_the_transform(engine_ctx, oracle_ctx):
    logn, dim, batch = 15, 2, 48
    g, o = engine_ctx(logn, dim), oracle_ctx(logn, dim)
    lib = _native.load()
    slab_host = np.concatenate([o.gen(500 + k, dim) for k in range(batch)])
    want = o.ntt_slab(slab_host, dim)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for rnd in range(3):
        slab = to_device(slab_host)
        out = torch.zeros_like(slab)
        late = torch.zeros(1 << 22, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            g.poly_ntt(slab, dim)                                                  # four launches, ~100 us of work on stream A
        _native.check(lib.gpq_stream_wait(C.c_void_p(sb.cuda_stream), C.c_void_p(sa.cuda_stream)), "gpq_stream_wait")
        with torch.cuda.stream(sa):
            late.fill_(7)                                                          # queued on A AFTER the wait: B does not depend on it
        _native.check(lib.gpq_copy(C.c_void_p(out.data_ptr()), C.c_void_p(slab.data_ptr()), slab.numel() * 8, C.c_void_p(sb.cuda_stream)), "gpq_copy")
        sb.synchronize()
        assert np.array_equal(to_host(out), want), "round %d: the copy on stream B overtook the transform on stream A" % rnd
        torch.cuda.synchronize()
        assert int(late[0]) == 7


def test_null_arguments_mean_the_null_stream():
    lib = _native.load()
    _native.check(lib.gpq_stream_wait(None, None), "gpq_stream_wait")             # legal: the null stream waiting for itself
    torch.cuda.synchronize()
