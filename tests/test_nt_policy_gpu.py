"""gpq_set_nt_policy (include/gpqhe_hip.h): the slab loads / stores of the transform kernels with the default cache policy, non-temporal, or
chosen by the launch group's working set.  A cache policy must never change a word: every entry point built on the transform kernels
(src/ntt.c:37-73 behind src/poly.c:poly_mul, src/he-mult.c:117-141 and :59-66, :179-185) gives the same words in all three modes; mode 0
is the path the oracle tests pin."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("logn,dim,batch", [(9, 3, 5), (12, 6, 3), (15, 10, 4), (16, 12, 3)])
def test_every_policy_gives_the_same_words(engine_ctx, logn, dim, batch):
    import torch
    from bench import rand_slab
    g = engine_ctx(logn, dim)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(900 + logn)
    src = [rand_slab(torch, g, dim, batch, gen) for _ in range(5)]
    evk = [rand_slab(torch, g, dim, 1, gen) for _ in range(2)]

    def run():
        out = []
        a = [t.clone() for t in src]
        g.poly_ntt(a[0], dim); out.append(a[0].clone())
        g.poly_invntt(a[1], dim); out.append(a[1].clone())
        r = torch.empty_like(a[2])
        g.poly_mul_rns(r, a[2], a[3], dim); out.append(r)
        b = [t.clone() for t in src]
        r0, r1 = torch.empty_like(b[0]), torch.empty_like(b[0])
        g.mulpt_rns(r0, r1, b[0], b[1], b[2], dim); out += [r0, r1]
        c = [t.clone() for t in src]
        d = [torch.empty_like(c[0]) for _ in range(3)]
        g.he_mul_tensor(d[0], d[1], d[2], c[0], c[1], c[2], c[3], dim); out += d
        sq = [torch.empty_like(c[0]) for _ in range(3)]
        g.he_mul_tensor(sq[0], sq[1], sq[2], c[0], c[1], c[0], c[1], dim); out += sq
        k = [torch.empty_like(c[0]) for _ in range(2)]
        g.he_keyswitch(k[0], k[1], c[4], evk[0], evk[1], dim); out += k
        torch.cuda.synchronize()
        return out

    try:
        g.set_nt_policy(0)
        want = run()
        g.set_nt_policy(1)
        always = run()
        g.set_nt_policy(-1)
        auto = run()
    finally:
        g.set_nt_policy(-1)
    for i, (w, a, b) in enumerate(zip(want, always, auto)):
        assert torch.equal(w, a), "result %d differs with non-temporal slab traffic" % i
        assert torch.equal(w, b), "result %d differs with the working-set policy" % i
    assert bool((want[2] != 0).any()) and bool((want[-1] != 0).any())


def test_policy_argument_is_checked(engine_ctx):
    import gpqhe_amd
    g = engine_ctx(9, 3)
    for bad in (-2, 2, 7):
        with pytest.raises(gpqhe_amd.GpqError):
            g.set_nt_policy(bad)
    g.set_nt_policy(-1)
