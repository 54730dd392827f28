#!/usr/bin/env python3
"""bench.py -- he_mul RNS-core throughput of the MI355X engine.

Workload (BASELINE.json configs[2]): n = 2^16, tensor stage over
dimA = 30 limbs (src/he-mult.c:116-138) + key-switch inner product over
dimB = 45 limbs (src/he-mult.c:58-66) for a batch of 64 independent
ciphertext multiplications per GPU, synthetic uniform residues already in RNS
form and resident in HBM.  One "step" = the whole batch once.  he_rescale has
no RNS-domain work in the reference (src/he-rescale.c:33-54 is big-integer
only), so it is not part of the timed RNS core.

Multi-GPU: independent ciphertexts, one shard per rank, no data-path collective
(weak scaling); the only collectives are the timing barrier and a MAX.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOGN, DIM_A, DIM_B = 16, 30, 45
ALGO_BYTES_PER_HE_MUL = (7 * DIM_A + 5 * DIM_B) * (8 << LOGN)  # 228,065,280
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

# own read+write bytes of one (limb, polynomial) unit of each kernel, in limbs of n*8 bytes
KERNEL_LIMB_PASSES = {"strided_fwd": 2, "strided_inv": 2, "tensor_mid": 7, "keyswitch_mid": 5,
                      "contig_fwd": 2, "contig_inv": 2, "pointwise": 3}


def rand_slab(torch, ctx, dim, batch, gen):
    """uniform residues in [0, p_d), limb-major [batch][dim][n]"""
    out = torch.empty((batch, dim, ctx.n), dtype=torch.int64, device="cuda")
    for d in range(dim):
        out[:, d, :] = torch.randint(0, ctx.p[d], (batch, ctx.n), dtype=torch.int64, device="cuda", generator=gen)
    return out.reshape(-1)


def pmc_traffic(kernel, chunk):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/, separate
    rocprofv3 --pmc runs of this same command): 2*FETCH_SIZE + WRITE_SIZE in KiB, the factor 2
    being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section).  The PMC run
    used launch groups of `pmc_chunk` polynomials; traffic scales with the group size."""
    import glob
    import re
    def order(path):   # profiles/rNN/vMM_pmc_summary.json: by round, then by version number (v10 after v9)
        m = re.search(r"r(\d+)[/\\]v(\d+)_", path)
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_summary.json")), key=order)
    if not files:
        return None
    try:
        data = json.load(open(files[-1]))
        for name, v in data.items():
            if kernel in name.replace("strided_pass<8, 4, false", "strided_fwd").replace("strided_pass<8, 4, true", "strided_inv"):
                return int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 * chunk / data.get("_chunk", 4))
    except Exception:
        return None
    return None


def power_state(torch, step, seconds=2.5):
    """Shader clock and package power while the hot loop runs (rocm-smi polled during an extra, untimed stretch of steps):
    on this part the loop sits at the package power cap, which is what sets its clock.  None if rocm-smi is unavailable."""
    import re
    import shutil
    import subprocess
    if shutil.which("rocm-smi") is None:
        return None
    try:
        dev = torch.cuda.current_device()
        t0, clk, pw = time.perf_counter(), [], []
        while time.perf_counter() - t0 < seconds:
            for _ in range(11):
                step()                                   # asynchronous: about as much work as one poll takes
            out = subprocess.run(["rocm-smi", "-d", str(dev), "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            w = re.search(r"Power \(W\): ([0-9.]+)", out)
            if m and w:
                clk.append(int(m.group(1))); pw.append(float(w.group(1)))
        torch.cuda.synchronize()
        if len(clk) < 2:
            return None
        clk, pw = sorted(clk[1:]), sorted(pw[1:])       # the first sample may precede the ramp
        return {"sclk_MHz": clk[len(clk) // 2], "package_W": pw[len(pw) // 2], "samples": len(clk),
                "how": "rocm-smi polled during an extra untimed stretch of the same steps"}
    except Exception:
        return None


def ntt_rate(torch, gpqhe_amd, logn, dim, batch, iters=20):
    """Second half of BASELINE's metric: NTT GB/s = 16*n bytes per limb per direction
    (read + write once) over a forward+inverse pair, HIP-event timed."""
    ctx = gpqhe_amd.PolyContext(logn, dim)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    slab = rand_slab(torch, ctx, dim, batch, gen)
    ref = slab.clone()
    ctx.poly_ntt(slab, dim)
    ctx.poly_invntt(slab, dim)          # the round trip must be the identity
    ok = bool(torch.equal(slab, ref))
    for _ in range(3):                  # warm-up: the first pairs after a synchronisation run ~10 % slower
        ctx.poly_ntt(slab, dim)
        ctx.poly_invntt(slab, dim)
    t = gpqhe_amd.StreamTimer()
    t.start()
    for _ in range(iters):
        ctx.poly_ntt(slab, dim)
        ctx.poly_invntt(slab, dim)
    t.stop()
    ms = t.elapsed_ms() / iters
    byts = 2 * 16 * (1 << logn) * dim * batch
    ctx.close()
    return {"shape": "n=2^%d, %d limbs, batch %d, forward+inverse" % (logn, dim, batch), "ms_per_pair": round(ms, 4),
            "GBps": round(byts / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roundtrip_identity": ok}


def he_mul_mpi_rate(torch, gpqhe_amd, ctx, batch, iters=6):
    """Whole he_mul of src/he-mult.c:88-156 on device big slabs (q = 2^850: decompose, tensor, CRT, relinearise
    with exact division by P, centre) -- SURVEY.md 8f rank 1-2, reported beside the RNS-core headline."""
    logq, W = 850, 14
    dimP, dimA, dimB, _ = ctx.he_dims(logq, logq)
    n = ctx.n
    gen = torch.Generator(device="cuda")
    gen.manual_seed(21)

    def centred():
        big = torch.randint(-(1 << 62), 1 << 62, (batch, W, n), dtype=torch.int64, device="cuda", generator=gen)
        big[:, W - 1] = torch.randint(-(1 << 16), 1 << 16, (batch, n), dtype=torch.int64, device="cuda", generator=gen)
        return big.reshape(-1).contiguous()

    cts = [centred() for _ in range(4)]
    rlk0, rlk1 = rand_slab(torch, ctx, dimB, 1, gen), rand_slab(torch, ctx, dimB, 1, gen)
    o0, o1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
    for _ in range(3):
        ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP)
    t = gpqhe_amd.StreamTimer()
    t.start()
    for _ in range(iters):
        ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP)
    t.stop()
    ms = t.elapsed_ms() / iters
    # BASELINE configs[2] is "he_mul + he_rescale": the same with he_rs (src/he-rescale.c:33-54, Delta = 2^50) after each product
    t.start()
    for _ in range(iters):
        ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP)
        ctx.he_rs(o0, o1, W, 50, logq - 50)
    t.stop()
    ms_rs = t.elapsed_ms() / iters
    return {"shape": "n=2^16, q=2^850 (W=14 words), dimA/dimB/dimP=%d/%d/%d, batch %d" % (dimA, dimB, dimP, batch),
            "ms_per_batch": round(ms, 3), "he_mul_per_s": round(batch / (ms * 1e-3), 1),
            "he_mul_plus_he_rescale_per_s": round(batch / (ms_rs * 1e-3), 1)}


def keyswitch_n17_rate(torch, gpqhe_amd, batch=16, iters=3):
    """BASELINE configs[4] shape: the key-switch inner product (he_swk loop, src/he-automorphism.c:59-67) at n = 2^17, 44 limbs."""
    logn, dim = 17, 44
    ctx = gpqhe_amd.PolyContext(logn, dim)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(17)
    x = rand_slab(torch, ctx, dim, batch, gen)
    e0, e1 = rand_slab(torch, ctx, dim, 1, gen), rand_slab(torch, ctx, dim, 1, gen)
    c0, c1 = torch.empty_like(x), torch.empty_like(x)
    ws = ctx.keyswitch_workspace(dim, batch)
    ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
    t = gpqhe_amd.StreamTimer()
    t.start()
    for _ in range(iters):
        ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
    t.stop()
    ms = t.elapsed_ms() / iters
    algo = 5 * dim * (8 << logn) * batch
    ctx.close()
    return {"shape": "n=2^17, 44 limbs, batch %d" % batch, "ms_per_batch": round(ms, 3), "keyswitch_per_s": round(batch / (ms * 1e-3), 1),
            "algo_GBps": round(algo / (ms * 1e-3) / 1e9, 1)}


def cpu_baseline(ctx, host_inputs, gpu_outputs, sample):
    """The oracle (CPU restatement of the reference loops) timed on this host, one
    thread like the reference, on `sample` ciphertexts of the same workload; its
    outputs double as a bit-exact check of the GPU results for those ciphertexts."""
    from oracle.oracle import OracleCtx, lib
    lib().orc_set_threads(1)
    o = OracleCtx(LOGN, DIM_B)
    assert o.p == ctx.p[:DIM_B]
    perA, perB = DIM_A * o.n, DIM_B * o.n
    a0, a1, b0, b1, x, e0, e1 = host_inputs
    t0 = time.perf_counter()
    ok = True
    for k in range(sample):
        d = o.he_mul_tensor(*[np.ascontiguousarray(v[k * perA:(k + 1) * perA]) for v in (a0, a1, b0, b1)], DIM_A)
        c = o.keyswitch(np.ascontiguousarray(x[k * perB:(k + 1) * perB]), e0, e1, DIM_B)
        for got, exp, per in zip(gpu_outputs, list(d) + list(c), (perA, perA, perA, perB, perB)):
            ok = ok and np.array_equal(got[k * per:(k + 1) * per], exp)
    dt = time.perf_counter() - t0
    out = {"value": sample / dt, "unit": "he_mul/s", "cores": 1, "kind": "port",
           "sample": "%d he_mul RNS cores (tensor %d limbs + key-switch %d limbs, n=2^%d) of the same batch, %.1f s" % (sample, DIM_A, DIM_B, LOGN, dt),
           "bit_exact_vs_gpu": bool(ok)}
    # the same loops with OpenMP over the limbs on every core this process may use (optional, core count stated)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)             # the CPU share of a one-GPU box; more threads than limbs per stage buy nothing anyway
    if cores > 1:
        lib().orc_set_threads(cores)
        t1 = time.perf_counter()
        for k in range(sample):
            o.he_mul_tensor(*[np.ascontiguousarray(v[k * perA:(k + 1) * perA]) for v in (a0, a1, b0, b1)], DIM_A)
            o.keyswitch(np.ascontiguousarray(x[k * perB:(k + 1) * perB]), e0, e1, DIM_B)
        out["all_cores"] = {"value": sample / (time.perf_counter() - t1), "cores": cores, "threads": "OpenMP over limbs"}
        lib().orc_set_threads(1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="ciphertext multiplications per GPU per step")
    ap.add_argument("--chunk", type=int, default=0, help="polynomials per fused launch group (0 = library default)")
    ap.add_argument("--cpu-sample", type=int, default=2, help="ciphertexts the CPU baseline replays (0 = skip)")
    ap.add_argument("--no-ntt", action="store_true", help="skip the secondary legs run after the timed region (NTT GB/s, MPI-level he_mul, n=2^17 key switch)")
    ap.add_argument("--streams", type=int, default=1, help="2: tensor stage and key-switch stage on separate HIP streams")
    ap.add_argument("--scatter-gather", action="store_true", help="N>1 only: also time a step with the input slabs scattered from rank 0 "
                    "and the outputs gathered back (grouped isend/irecv = one RCCL group over xGMI); SURVEY.md 8d config 4")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path with several ranks on one GPU)")
    args = ap.parse_args()

    import torch
    import gpqhe_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        if args.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            torch.cuda.set_device(local % ndev)
            dist.init_process_group(backend=args.backend)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world:
        if rank == 0 and world > 1:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    ctx = gpqhe_amd.PolyContext(LOGN, DIM_B)
    if args.chunk:
        ctx.set_chunk(args.chunk)
    B = args.batch
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1000 + rank)
    a0, a1, b0, b1 = (rand_slab(torch, ctx, DIM_A, B, gen) for _ in range(4))
    x = rand_slab(torch, ctx, DIM_B, B, gen)                   # stands for decompose(d2), SURVEY.md 8d
    gk = torch.Generator(device="cuda")
    gk.manual_seed(3000)                                       # one relinearisation key shared by every rank
    e0, e1 = rand_slab(torch, ctx, DIM_B, 1, gk), rand_slab(torch, ctx, DIM_B, 1, gk)
    d0, d1, d2 = (torch.empty_like(a0) for _ in range(3))
    c0, c1 = torch.empty_like(x), torch.empty_like(x)
    wsA, wsB = ctx.tensor_workspace(DIM_A, B), ctx.keyswitch_workspace(DIM_B, B)

    streams = [torch.cuda.Stream() for _ in range(2)] if args.streams == 2 else None

    def step():
        if streams is None:
            ctx.he_mul_tensor(d0, d1, d2, a0, a1, b0, b1, DIM_A, wsA)
            ctx.he_keyswitch(c0, c1, x, e0, e1, DIM_B, wsB)
        else:  # the two stages of different ciphertexts are independent: let their launch tails overlap
            with torch.cuda.stream(streams[0]):
                ctx.he_mul_tensor(d0, d1, d2, a0, a1, b0, b1, DIM_A, wsA)
            with torch.cuda.stream(streams[1]):
                ctx.he_keyswitch(c0, c1, x, e0, e1, DIM_B, wsB)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    prof = ctx.profile_collect()
    if dist is not None:
        from gpqhe_amd.dist import max_over_ranks
        dt = max_over_ranks(dt)

    sg = None
    if dist is not None and args.scatter_gather:
        # BASELINE configs[3] with the transfers inside the timed region: rank 0 owns the whole batch, every rank works on its
        # shard, results return to rank 0.  A root-GPU scatter is bound by one xGMI link per peer.
        from gpqhe_amd.dist import scatter_slab, gather_slab, max_over_ranks
        Bs = min(B, 16)
        dev = torch.device("cuda", torch.cuda.current_device())
        per_a, per_b = DIM_A * ctx.n, DIM_B * ctx.n
        full_in = [torch.cat([v[: Bs * per_a]] * world) if rank == 0 else None for v in (a0, a1, b0, b1)]
        full_x = torch.cat([x[: Bs * per_b]] * world) if rank == 0 else None
        barrier()
        t1 = time.perf_counter()
        sa = [scatter_slab(f, per_a, Bs * world, 0, dev) for f in full_in]
        sx = scatter_slab(full_x, per_b, Bs * world, 0, dev)
        torch.cuda.synchronize()
        o = [torch.empty_like(sa[0]) for _ in range(3)] + [torch.empty_like(sx) for _ in range(2)]
        ctx.he_mul_tensor(o[0], o[1], o[2], sa[0], sa[1], sa[2], sa[3], DIM_A, wsA)
        ctx.he_keyswitch(o[3], o[4], sx, e0, e1, DIM_B, wsB)
        torch.cuda.synchronize()      # the results must exist before a backend without stream semantics (gloo rehearsal) reads them
        back = [gather_slab(o[i], per_a if i < 3 else per_b, Bs * world, 0) for i in range(5)]
        barrier()
        dsg = max_over_ranks(time.perf_counter() - t1)
        moved = (4 * per_a + per_b + 3 * per_a + 2 * per_b) * 8 * Bs * (world - 1)
        sg = {"batch_per_gpu": Bs, "he_mul_per_s": round(Bs * world / dsg, 1), "ms": round(dsg * 1e3, 2), "bytes_over_links": moved,
              "GBps_root": round(moved / dsg / 1e9, 1)}
        if rank == 0:   # the shards are copies of the first ciphertexts: every shard's result equals rank 0's own
            assert all(torch.equal(back[i][: o[i].numel()], back[i][o[i].numel(): 2 * o[i].numel()]) for i in range(5))
        del full_in, full_x, sa, sx, o, back

    if rank == 0:
        total_he_mul = world * B * args.steps
        value = total_he_mul / dt
        # dominant kernel by accumulated device time
        kname, (kms, kcnt) = max(prof.items(), key=lambda kv: kv[1][0])
        # strided kernels run for both stages: average units per launch from the launch mix
        chunk = min(B, args.chunk or 32)
        units = {"tensor_mid": DIM_A * chunk, "keyswitch_mid": DIM_B * chunk,
                 "strided_fwd": (4 * DIM_A + 1 * DIM_B) * chunk / 2.0, "strided_inv": (3 * DIM_A + 2 * DIM_B) * chunk / 2.0}
        kernels = {}
        for name, (ms, cnt) in prof.items():
            avg_ms = ms / cnt
            byts = KERNEL_LIMB_PASSES[name] * units.get(name, 0) * (8 << LOGN)
            kernels[name] = {"avg_ms": round(avg_ms, 4), "launches": int(cnt), "share": round(ms / sum(v[0] for v in prof.values()), 3),
                             "algo_GBps": round(byts / (avg_ms * 1e-3) / 1e9, 1)}
        kavg = kms / kcnt
        kbytes = KERNEL_LIMB_PASSES[kname] * units[kname] * (8 << LOGN)
        achieved = kbytes / (kavg * 1e-3) / 1e9
        traffic = pmc_traffic(kname, chunk)
        out = {
            "metric": "ciphertext he_mul/sec (RNS core: tensor 30 limbs + key-switch 45 limbs), N=2^16",
            "value": round(value, 2), "unit": "he_mul/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "he_mul RNS core, n=2^16, dimA=30, dimB=45, batch=%d ciphertexts per GPU (BASELINE configs[2]); "
                                   "he_rescale has no RNS-domain work in the reference" % B,
                       "batch_per_gpu": B, "chunk": chunk, "parallelism": "ciphertext-per-GPU x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": int(kbytes), "avg_launch_ms": round(kavg, 4)},
            "he_mul_e2e": {"algo_bytes_per_he_mul": ALGO_BYTES_PER_HE_MUL,
                           "achieved_GBps_per_gpu": round(ALGO_BYTES_PER_HE_MUL * value / world / 1e9, 1),
                           "hbm_frac_per_gpu": round(ALGO_BYTES_PER_HE_MUL * value / world / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels": kernels,
        }
        if sg is not None:
            out["with_scatter_gather"] = sg
        if world == 1 and args.cpu_sample > 0:
            s = args.cpu_sample
            host_in = [gpqhe_amd.to_host(v[: s * DIM_A * ctx.n]) for v in (a0, a1, b0, b1)] + \
                      [gpqhe_amd.to_host(x[: s * DIM_B * ctx.n]), gpqhe_amd.to_host(e0), gpqhe_amd.to_host(e1)]
            gpu_out = [gpqhe_amd.to_host(v[: s * DIM_A * ctx.n]) for v in (d0, d1, d2)] + \
                      [gpqhe_amd.to_host(v[: s * DIM_B * ctx.n]) for v in (c0, c1)]
            out["cpu_baseline"] = cpu_baseline(ctx, host_in, gpu_out, s)
        if world == 1 and not args.no_ntt:
            pw = power_state(torch, step)
            if pw is not None:
                out["power"] = pw
            # NTT GB/s at the headline ring (n=2^16, 30 limbs) and at BASELINE configs[1] (n=2^15, 10 limbs)
            out["ntt"] = [ntt_rate(torch, gpqhe_amd, 16, DIM_A, B), ntt_rate(torch, gpqhe_amd, 15, 10, 64)]
            del a0, a1, b0, b1, x, d0, d1, d2, c0, c1, wsA, wsB
            torch.cuda.empty_cache()
            out["ntt"].append(ntt_rate(torch, gpqhe_amd, 16, DIM_A, 4 * B))   # launch size matters: 4 GiB slab
            out["he_mul_mpi_level"] = he_mul_mpi_rate(torch, gpqhe_amd, ctx, B, iters=3)   # BASELINE configs[2]: he_mul + he_rescale, batch 64
            out["keyswitch_n17"] = keyswitch_n17_rate(torch, gpqhe_amd)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
