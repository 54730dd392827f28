"""BASELINE configs[3]'s own batch -- 512 independent ciphertext multiplications at n = 2^16, 30 / 45 limbs -- once on the one
device of the box (88 GB of slabs): the N = 1 anchor of the strong-scaling line.  512 oracle evaluations are out of reach, so the
batch is a block of 8 distinct ciphertexts repeated 64 times and is checked through what that gives: every ciphertext equals the
first occurrence of its block position (independence: src/he-mult.c:116-138 and :58-66 carry no cross-ciphertext state, and no
launch group of the engine may either), and the first and the last ciphertext equal the oracle."""
import numpy as np
import pytest

from gpqhe_amd import to_device, to_host

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_batch_of_512_ciphertexts_on_one_device(engine_ctx, oracle_ctx):
    import torch
    logn, dim_a, dim_b, batch, block = 16, 30, 45, 512, 8
    free, _ = torch.cuda.mem_get_info()
    if free < 100 << 30:
        pytest.skip("needs 100 GB of free HBM")
    o, g = oracle_ctx(logn, dim_b), engine_ctx(logn, dim_b)
    per_a, per_b = dim_a * o.n, dim_b * o.n
    host_in = [np.concatenate([o.gen(1000 + 4 * k + i, dim_a) for k in range(block)]) for i in range(4)]
    host_x = np.concatenate([o.gen(2000 + k, dim_b) for k in range(block)])
    ev = [o.gen(3000, dim_b), o.gen(3001, dim_b)]
    ins = [to_device(v).repeat(batch // block) for v in host_in]
    x = to_device(host_x).repeat(batch // block)
    e0, e1 = to_device(ev[0]), to_device(ev[1])
    d = [torch.empty_like(ins[0]) for _ in range(3)]
    c = [torch.empty_like(x) for _ in range(2)]
    g.he_mul_tensor(d[0], d[1], d[2], *ins, dim_a)
    g.he_keyswitch(c[0], c[1], x, e0, e1, dim_b)
    torch.cuda.synchronize()
    for name, t, per in [("d0", d[0], per_a), ("d1", d[1], per_a), ("d2", d[2], per_a), ("c0", c[0], per_b), ("c1", c[1], per_b)]:
        v = t.view(batch // block, block * per)
        for r in range(1, batch // block):
            assert torch.equal(v[r], v[0]), "%s: repeat %d of the block differs from its first occurrence" % (name, r)
    for k, pos in ((0, 0), (block - 1, batch - 1)):                        # first and last ciphertext of the batch against the oracle
        want = list(o.he_mul_tensor(*[v[k * per_a:(k + 1) * per_a] for v in host_in], dim_a))
        want += list(o.keyswitch(host_x[k * per_b:(k + 1) * per_b], ev[0], ev[1], dim_b))
        for name, t, w in zip(("d0", "d1", "d2", "c0", "c1"), d + c, want):
            per = w.size
            assert np.array_equal(to_host(t[pos * per:(pos + 1) * per]), w), "%s of ciphertext %d differs from the oracle" % (name, pos)
