"""Inputs whose forward transform holds residues 0 -- the places where src/ntt.c:45-48 keeps its data in [0, p], not [0, p):
    a[j] = (a[j] <= q - t) ? a[j] + t : a[j] + t - q
stores q when a[j] + t == q, and a stored q survives a later stage whenever its partner's product t is 0.  Shared by the CPU
test that pins the oracle's behaviour on them and the GPU tests that compare the HIP path with it."""
import numpy as np


def limb_cases(o, d, rng):
    """[(name, canonical input limb)] for prime d of oracle context o; targets are built in the NTT domain and pulled back with
    the oracle's invntt (src/ntt.c:54-73: canonical output), so ntt(input) has exactly the target's residues."""
    n, p = o.n, o.p[d]
    rnd = lambda: rng.integers(1, p, size=n, dtype=np.uint64)       # no zero residue anywhere
    cases = []

    def add(name, target):
        cases.append((name, o.invntt(np.ascontiguousarray(target, dtype=np.uint64), d)))

    add("no zero", rnd())
    t = rnd(); t[0] = 0; add("zero at sum-leg position 0", t)
    if n >= 2:
        t = rnd(); t[1] = 0; add("zero at difference-leg position 1 (x == t)", t)
        t = rnd(); t[0] = t[1] = 0; add("both legs of the last butterfly zero", t)
        t = rnd(); t[n - 2] = 0; add("zero at the last sum-leg position", t)
    if n >= 8:
        t = rnd(); t[0:4] = 0; add("a block of four zeros (a stored p meets t == 0 in the next stage)", t)
        t = rnd(); t[0:n // 2] = 0; add("lower half zero", t)
        t = rnd(); t[0::2] = 0; add("every sum-leg position zero", t)
        t = rnd(); t[1::2] = 0; add("every difference-leg position zero", t)
        t = rnd(); t[rng.integers(0, n, size=max(2, n // 16))] = 0; add("scattered zeros", t)
    cases.append(("all-zero limb (0 + 0 stays 0)", np.zeros(n, dtype=np.uint64)))
    one = np.zeros(n, dtype=np.uint64); one[0] = 1
    cases.append(("constant polynomial 1", one))
    if n >= 2:
        x = np.zeros(n, dtype=np.uint64); x[n // 2] = p - 1
        cases.append(("-(X^(n/2))", x))
    t = np.zeros(n, dtype=np.uint64); t[n - 1] = 5; add("a single non-zero NTT coefficient", t)
    return cases


def slab_of_cases(o, dim, seed):
    """One polynomial per case; limb d of polynomial k carries case k built for prime d.  Returns (names, slab uint64[cases][dim][n])."""
    rng = np.random.default_rng(seed)
    per_limb = [limb_cases(o, d, rng) for d in range(dim)]
    names = [c[0] for c in per_limb[0]]
    slab = np.concatenate([per_limb[d][k][1] for k in range(len(names)) for d in range(dim)])
    return names, slab
