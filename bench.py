#!/usr/bin/env python3
"""bench.py -- he_mul RNS-core throughput of the MI355X engine.

Workload (BASELINE.json configs[2]): n = 2^16, tensor stage over
dimA = 30 limbs (src/he-mult.c:116-138) + key-switch inner product over
dimB = 45 limbs (src/he-mult.c:58-66) for a batch of 64 independent
ciphertext multiplications per GPU, synthetic uniform residues already in RNS
form and resident in HBM.  One "step" = the whole batch once.  he_rescale has
no RNS-domain work in the reference (src/he-rescale.c:33-54 is big-integer
only), so it is not part of the timed RNS core.

Multi-GPU: independent ciphertexts, one shard per rank, no data-path collective;
the only collectives of the timed region are the barrier and a MAX.  One process per
GPU over RCCL.  Either start the ranks yourself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
or run `python bench.py --gpus N ...` alone: with WORLD_SIZE unset the parent (which never
touches a GPU) starts the N rank processes itself and exits with their status.
  --batch B         ciphertexts per GPU (weak scaling, the default: 64 per GPU)
  --total-batch T   T ciphertexts block-partitioned over the ranks (strong scaling;
                    BASELINE configs[3] = --total-batch 512 at N = 2/4/8)
With N > 1 the line also carries `with_host_scatter` (every rank uploads its own shard from
page-locked host memory and downloads its results: SURVEY.md 8e's alternative) and
`with_scatter_gather` (the input slabs sent from rank 0 over xGMI and the results returned to it),
both with the transfers inside the timed region.

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
SG_DEADLINE_S = 150    # watchdog of the secondary scatter/gather leg (N > 1)
SG_FAILED_STATUS = 3   # exit status of every rank when that leg stalls or fails (the line is still printed first)

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
    used launch groups of `_chunk` polynomials; traffic scales with the group size.
    Returns (bytes or None, provenance): the counters are NOT collected by the run that prints
    them (gpurun refuses counter collection next to other tracing), so the line says which
    committed summary the figure comes from and which commit that summary profiled."""
    import glob
    import re
    def order(path):   # profiles/rNN/vMM_pmc_summary.json: by round, then by version number (v10 after v9)
        m = re.search(r"r(\d+)[/\\]v(\d+)_", path)
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_summary.json")), key=order)
    if not files:
        return None, "no profiles/r*/*pmc_summary.json in the tree"
    rel = os.path.relpath(files[-1], ROOT)
    try:
        data = json.load(open(files[-1]))
    except (OSError, ValueError) as exc:
        return None, "%s unreadable: %s" % (rel, exc)
    src = {"file": rel, "profiled_head": data.get("_head"), "profiled_chunk": data.get("_chunk", 4),
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB per launch, scaled by chunk/profiled_chunk"}
    for name, v in data.items():
        if name.startswith("_"):
            continue
        if kernel in name.replace("strided_pass<8, 4, false", "strided_fwd").replace("strided_pass<8, 4, true", "strided_inv"):
            src["kernel"] = name
            return int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 * chunk / data.get("_chunk", 4)), src
    src["error"] = "no kernel matching %r in the summary" % kernel
    return None, src


def valu_issue(value_per_gpu, sclk_mhz):
    """How fast the he_mul core issues integer-VALU instructions against the rate at which the SAME instruction mix issues with no
    memory traffic -- the bound that actually binds it.  Nothing here is a literal: the instruction count per he_mul comes
    from the newest committed PMC pass (SQ_INSTS_VALU of the four kernels of the core, profiles/r*/v*_pmc_summary.json), the bound from
    the newest committed probe evaluation (profiles/r*/v*_valu_bound.json: tools/issue_probe runs the library's own butterfly groups in a
    register-resident loop on every SIMD; tools/valu_bound.py turns its run, the probe's assembly and the rocm-smi samples into VALU
    wave-instructions per second, cycles per instruction, clock and package power).  Both files name the commit they were taken at; the
    probe file also carries a hash of the kernel headers, and the object says `stale` when the sources of this run differ from it."""
    import glob
    import hashlib
    import re
    def order(path):
        m = re.search(r"r(\d+)[/\\]v(\d+)_", path)
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_summary.json")), key=order)
    bounds = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*valu_bound.json")), key=order)
    if not pmcs or not bounds:
        return None
    try:
        data = json.load(open(pmcs[-1]))
        bound = json.load(open(bounds[-1]))
        per_group = 0.0
        for name, v in data.items():
            if name.startswith("_"):
                continue
            launches_per_group = 2 if "strided_pass" in name else 1      # a strided pass runs once for the tensor stage and once for the key switch
            per_group += v["SQ_INSTS_VALU"] * launches_per_group
        insts = per_group / data.get("_chunk", 16)
        ib = bound["issue_bound"]
        csrc = os.path.join(ROOT, "gpqhe_amd", "csrc")
        now = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in ("modarith.hpp", "ntt_kernels.hpp"))).hexdigest()[:16]
    except (OSError, ValueError, KeyError) as exc:
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    rate = insts * value_per_gpu                                           # VALU wave-instructions per second, whole chip
    out = {"valu_wave_insts_per_he_mul": int(insts), "valu_wave_insts_per_s": round(rate, -7),
           "insts_source": {"file": os.path.relpath(pmcs[-1], ROOT), "profiled_head": data.get("_head")},
           "bound_valu_wave_insts_per_s": ib["valu_wave_insts_per_s"],
           "bound_source": {"file": os.path.relpath(bounds[-1], ROOT), "probe_head": bound.get("_head"), "probe_sclk_MHz": ib.get("sclk_MHz"),
                            "probe_package_W": ib.get("package_W"), "probe_cycles_per_valu_inst": ib.get("cycles_per_valu_inst"),
                            "kernel_source_sha16": bound.get("_kernel_source_sha16")},
           "frac_of_valu_issue_rate": round(rate / ib["valu_wave_insts_per_s"], 3),
           "stale": now != bound.get("_kernel_source_sha16"),
           "note": "integer-VALU instructions per second of the core against the same instruction mix with no memory traffic (the probe runs un-capped at "
                   "its clock; the core sits at the package power cap and a lower clock: `power`); `roofline` above prices the same run against HBM"}
    if sclk_mhz:
        cpi = sclk_mhz * 1e6 * 256 * 4 / rate
        out.update({"sclk_MHz": sclk_mhz, "cycles_per_valu_inst": round(cpi, 3),
                    "frac_of_probe_cycles_per_inst": round(ib["cycles_per_valu_inst"] / cpi, 3) if ib.get("cycles_per_valu_inst") else None})
    return out


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


def copy_rate(torch, gpqhe_amd, mib=2048, iters=8):
    """What this device's memory system delivers to a PLAIN stream with the library's own access shape (16 bytes per lane, persistent
    workgroups: gpq_probe_stream), measured in this process: the best read+write copy over a few launch shapes, and the best read-only and
    write-only streams.  `GBps` (the copy) is the yardstick for `of_copy_rate`: a kernel that reads and writes cannot beat the best plain
    copy, so every `of_copy_rate` must come out <= 1 -- tests/test_bench_contract.py asserts it, with the 3 % by which two measurements of the
    yardstick itself differ (round 3 used torch's Tensor.copy_, which two kernels beat: a soft yardstick).  Since the transform kernels' slab
    traffic is non-temporal (round 4) the forward strided pass runs at 0.93-0.97 of it."""
    from gpqhe_amd import _native
    lib = _native.load()
    nbytes = mib * 1024 * 1024
    a = torch.empty(nbytes // 8, dtype=torch.int64, device="cuda")
    a.random_()
    b = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    shapes = {0: [(256, 256, 4), (256, 256, 8), (512, 256, 4), (256, 512, 4), (256, 512, 2), (512, 512, 2), (256, 1024, 2), (1024, 256, 2), (1024, 512, 1),
                  (2048, 256, 1), (512, 1024, 1)],
              1: [(256, 256, 8), (256, 512, 4), (512, 256, 4), (2048, 1024, 1), (2048, 256, 1)],
              2: [(256, 256, 1), (512, 256, 1), (256, 512, 1), (1024, 256, 1)]}
    best = {}
    for kind, cfgs in shapes.items():
        for blocks, threads, unroll in cfgs:
            for _ in range(2):
                _native.check(lib.gpq_probe_stream(b.data_ptr(), a.data_ptr(), nbytes, kind, blocks, threads, unroll, st), "gpq_probe_stream")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                _native.check(lib.gpq_probe_stream(b.data_ptr(), a.data_ptr(), nbytes, kind, blocks, threads, unroll, st), "gpq_probe_stream")
            e1.record()
            torch.cuda.synchronize()
            rate = (2 if kind == 0 else 1) * nbytes / (e0.elapsed_time(e1) / iters) / 1e6
            if rate > best.get(kind, (0, None))[0]:
                best[kind] = (rate, (blocks, threads, unroll))
    del a, b
    return {"GBps": round(best[0][0], 1), "read_only_GBps": round(best[1][0], 1), "write_only_GBps": round(best[2][0], 1),
            "best_shapes_blocks_threads_unroll": {"copy": best[0][1], "read": best[1][1], "write": best[2][1]},
            "how": "gpq_probe_stream (the library's own kernels: 16 B per lane, persistent workgroups, nt accesses) over %d MiB, best of %d launch shapes per kind; "
                   "GBps = read + written bytes of the copy" % (mib, sum(len(v) for v in shapes.values()))}


def valu_floor(value_per_gpu, sclk_mhz):
    """The plain issue floor of the he_mul core at THIS run's clock: SQ_INSTS_VALU per he_mul (newest committed PMC pass) x 4 cycles per
    wave-instruction on a SIMD-32 / (1024 SIMDs x the shader clock rocm-smi showed during this run).  frac = floor time / measured time:
    how much of the step the integer VALUs are busy issuing at that clock; the rest is memory latency in the strided passes and launch tails."""
    import glob
    import re
    def order(path):
        m = re.search(r"r(\d+)[/\\]v(\d+)_", path)
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_summary.json")), key=order)
    if not pmcs or not sclk_mhz:
        return None
    try:
        data = json.load(open(pmcs[-1]))
        per_group = sum(v["SQ_INSTS_VALU"] * (2 if "strided_pass" in name else 1) for name, v in data.items() if not name.startswith("_"))
        insts = per_group / data.get("_chunk", 16)
    except (OSError, ValueError, KeyError) as exc:
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    floor_s = insts * 4 / (1024 * sclk_mhz * 1e6)
    return {"valu_wave_insts_per_he_mul": int(insts), "cycles_per_wave_inst": 4, "simds": 1024, "sclk_MHz": sclk_mhz,
            "floor_us_per_he_mul": round(floor_s * 1e6, 2), "floor_he_mul_per_s": round(1 / floor_s, 1),
            "frac": round(floor_s * value_per_gpu, 4),
            "insts_source": {"file": os.path.relpath(pmcs[-1], ROOT), "profiled_head": data.get("_head")},
            "note": "frac = (instructions x 4 cycles / (1024 SIMDs x this run's clock)) x he_mul/s: the share of the step the integer VALUs spend issuing; "
                    "this, not HBM, is what binds the RNS core (DESIGN.md 5)"}


def ntt_rate(torch, gpqhe_amd, logn, dim, batch, iters=20):
    """Second half of BASELINE's metric: NTT GB/s = 16*n bytes per limb per direction
    (read + write once) over a forward+inverse pair, HIP-event timed.  The pairs are warmed for 0.15 s first: the leg
    starts behind host-side table construction, and the first ~10 ms after an idle stretch run at a ramping clock (up to 13 % slower at
    configs[1], profiles/r04/v5_ntt_streams.txt round 0 vs rounds 1-2); three timed repetitions, the median reported, all three listed."""
    ctx = gpqhe_amd.PolyContext(logn, dim)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    slab = rand_slab(torch, ctx, dim, batch, gen)
    ref = slab.clone()
    ctx.poly_ntt(slab, dim)
    ctx.poly_invntt(slab, dim)          # the round trip must be the identity
    ok = bool(torch.equal(slab, ref))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:      # steady clock before anything is timed
        for _ in range(5):
            ctx.poly_ntt(slab, dim)
            ctx.poly_invntt(slab, dim)
        torch.cuda.synchronize()
    reps = []
    for _ in range(3):
        t = gpqhe_amd.StreamTimer()
        t.start()
        for _ in range(iters):
            ctx.poly_ntt(slab, dim)
            ctx.poly_invntt(slab, dim)
        t.stop()
        reps.append(t.elapsed_ms() / iters)
    ms = sorted(reps)[1]
    byts = 2 * 16 * (1 << logn) * dim * batch
    ctx.close()
    return {"shape": "n=2^%d, %d limbs, batch %d, forward+inverse" % (logn, dim, batch), "ms_per_pair": round(ms, 4),
            "GBps": round(byts / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "ms_per_pair_repetitions": [round(v, 4) for v in reps], "roundtrip_identity": ok}


def he_mul_mpi_rate(torch, gpqhe_amd, ctx, batch, iters=6, logq=850, two_lanes=True, checked=None, sample_clocks=False, brief=False, restore_overlap=-1, group=32):
    """Whole he_mul of src/he-mult.c:88-156 on device big slabs (q = 2^logq: decompose, tensor, CRT, relinearise
    with exact division by P, centre) -- SURVEY.md 8f rank 1-2, reported beside the RNS-core headline."""
    W = logq // 64 + 1
    dimP, dimA, dimB, _ = ctx.he_dims(logq, logq)
    n = ctx.n
    gen = torch.Generator(device="cuda")
    gen.manual_seed(21)

    def centred():
        big = torch.randint(-(1 << 62), 1 << 62, (batch, W, n), dtype=torch.int64, device="cuda", generator=gen)
        top = logq - 2 - 64 * (W - 1)
        big[:, W - 1] = torch.randint(-(1 << top), 1 << top, (batch, n), dtype=torch.int64, device="cuda", generator=gen)
        return big.reshape(-1).contiguous()

    cts = [centred() for _ in range(4)]
    rlk0, rlk1 = rand_slab(torch, ctx, dimB, 1, gen), rand_slab(torch, ctx, dimB, 1, gen)
    o0, o1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
    ctx.set_overlap(-1)                         # what the library picks for this shape by itself (gpq_set_overlap's default)
    ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP)
    torch.cuda.synchronize()
    default_lanes = ctx.last_lanes()
    ctx.set_overlap(two_lanes)                  # (False: tools/mpi_profile.py under rocprofv3 -- kernel durations of one lane, nothing running beside them)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:      # steady clock (see ntt_rate)
        ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP)
        torch.cuda.synchronize()
    t = gpqhe_amd.StreamTimer()
    t.start()
    for _ in range(iters):
        ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP)
    t.stop()
    ms = t.elapsed_ms() / iters
    lanes_this_run = ctx.last_lanes()                # what the timed calls really ran on (a declined peer means 1 whatever was asked for)
    if brief:                                        # bench.py --quick: the two rates, nothing else
        ctx.he_mul_rs(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP, 50)
        t.start()
        for _ in range(iters):
            ctx.he_mul_rs(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP, 50)
        t.stop()
        ctx.set_overlap(restore_overlap)
        return {"shape": "n=2^%d, q=2^%d (W=%d words), dimA/dimB/dimP=%d/%d/%d, batch %d" % (ctx.logn, logq, W, dimA, dimB, dimP, batch),
                "ms_per_batch": round(ms, 3), "he_mul_per_s": round(batch / (ms * 1e-3), 1),
                "he_mul_plus_he_rescale_per_s": round(batch / (t.elapsed_ms() / iters * 1e-3), 1), "lanes": {"default": default_lanes, "this_run": lanes_this_run}}
    ctx.set_overlap(False)                           # the same on the caller's stream alone (gpq_set_overlap(ctx, 0)): what the two lanes buy
    t.start()
    for _ in range(iters):
        ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP)
    t.stop()
    ms_one = t.elapsed_ms() / iters
    # the clock under each mode (the lanes share one power cap: where two lanes buy nothing, this says whether the clock gave the gain back)
    clocks = {}
    if sample_clocks:
        for lanes in (1, 0):
            ctx.set_overlap(bool(lanes))
            pw = power_state(torch, lambda: ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP), seconds=1.2)
            if pw is not None:
                clocks["two_lanes" if lanes else "one_lane"] = {"sclk_MHz": pw["sclk_MHz"], "package_W": pw["package_W"]}
    ctx.set_overlap(two_lanes)
    ctx.profile(True)                                # a second pass with HIP events around every launch of this leg (on the launch stream) for the
    for _ in range(iters):                           # breakdown: the events themselves cost a few per cent at small shapes, so the rate above is timed without them
        ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP)
    torch.cuda.synchronize()
    ctx.profile(False)
    prof = ctx.profile_collect()
    # BASELINE configs[2] is "he_mul + he_rescale": the same with he_rs (src/he-rescale.c:33-54, Delta = 2^50) after each product
    t.start()
    for _ in range(iters):
        ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP)
        ctx.he_rs(o0, o1, W, 50, logq - 50)
    t.stop()
    ms_rs_two = t.elapsed_ms() / iters
    # ... as ONE call (gpq_he_mul_rs: the rounding division rides in the relinearisation tail; same words, tests/test_he_mul_rs_gpu.py)
    ctx.he_mul_rs(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP, 50)
    t.start()
    for _ in range(iters):
        ctx.he_mul_rs(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP, 50)
    t.stop()
    ms_rs = t.elapsed_ms() / iters
    check = None
    if checked:
        # Outside every timed region: one more product + rescale at this launch shape, and the first and last ciphertext of EACH launch group
        # packed up for the restated reference (oracle/expect.py: src/he-mult.c:88-156, src/he-rescale.c:33-54) -- main() runs the workers at the
        # end of the run and writes `bit_exact_vs_restated_reference` into this leg
        ctx.he_mul(o0, o1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP)
        r0, r1 = torch.empty_like(o0), torch.empty_like(o1)
        ctx.he_mul_rs(r0, r1, *cts, rlk0, rlk1, W, logq, dimA, dimB, dimP, 50)      # the fused call is what the rate above timed: check ITS words
        torch.cuda.synchronize()
        per = W * n                                  # `group` = the context's launch group (--chunk; the library's default is 32)
        picks = sorted({k for g0 in range(0, batch, group) for k in (g0, min(g0 + group, batch) - 1)})
        host = gpqhe_amd.to_host
        k0h, k1h = host(rlk0), host(rlk1)
        check = {"leg": checked, "picks": picks, "names": ("c0", "c1", "rs0", "rs1"),
                 "tasks": [dict(kind="he_mul", logn=ctx.logn, dimP=dimP, dimA=dimA, dimB=dimB, W=W, logq=logq, rs=50, rlk0=k0h, rlk1=k1h,
                                ct=[host(v[k * per:(k + 1) * per]) for v in cts]) for k in picks],
                 "got": [{nm: host(v[k * per:(k + 1) * per]) for nm, v in zip(("c0", "c1", "rs0", "rs1"), (o0, o1, r0, r1))} for k in picks]}
    # Algorithmic bytes of the bridge per he_mul, each slab read or written once (words of 8 bytes per coefficient):
    #   rns_decompose     4 x (W in, dimA out) + (W in, dimB out)                                   src/he-mult.c:117-120, :59
    #   CRT (poly_rns2mpi) 3 x (dimA in, W out) for d0, d1, d2 + 2 x (cnt in, W addend in, W out)    :139-141, the tail's CRT of Q (:67-77)
    #   relin front       2 x (dimB in, cnt out)                                                    exact division by P, :70
    #   one-product tail  2 x (dimB in, W addend in, W out): replaces the relin front and the tail's CRT
    #   streaming bridge (gpqhe_amd/csrc/bridge_stream.hpp): CRT(d2hat) -> rns_decompose in one kernel (dimA in, dimB out);
    #                     the one-product tail with its addend's limbs as rows of the same product, 2 x (dimB + dimA in, W out)
    cnt = dimB - dimP
    if "bridge_tail_stream" in prof:
        words = {"bridge_decompose": 4 * (W + dimA), "bridge_crt_decompose": dimA + dimB, "bridge_tail_stream": 2 * (dimB + dimA + W)}
    elif "bridge_relin_tail_direct" in prof:
        words = {"bridge_decompose": 4 * (W + dimA) + (W + dimB), "bridge_reconstruct": 3 * (dimA + W), "bridge_relin_tail_direct": 2 * (dimB + 2 * W)}
    else:
        words = {"bridge_decompose": 4 * (W + dimA) + (W + dimB), "bridge_reconstruct": 3 * (dimA + W) + 2 * (cnt + 2 * W), "bridge_relin_front": 2 * (dimB + cnt)}
    core = {"strided_fwd": 2 * (4 * dimA + dimB), "strided_inv": 2 * (3 * dimA + 2 * dimB), "tensor_mid": 7 * dimA, "keyswitch_mid": 5 * dimB}
    total_ms = sum(v[0] for v in prof.values())
    kernels = {}
    for name, (kms, kcnt) in prof.items():
        per_call = kms / iters
        rec = {"ms_per_batch": round(per_call, 4), "launches_per_batch": round(kcnt / iters, 2), "share": round(kms / total_ms, 3)}
        w = words.get(name, core.get(name))
        if w:
            rec["algo_words_per_coefficient"] = w
            rec["algo_GBps"] = round(w * 8 * n * batch / (per_call * 1e-3) / 1e9, 1)
        kernels[name] = rec
    bridge = {k: v for k, v in kernels.items() if k.startswith("bridge_") and "algo_GBps" in v}
    roof = None
    if bridge:
        kname = max(bridge, key=lambda k: bridge[k]["ms_per_batch"])
        ach = bridge[kname]["algo_GBps"]
        roof = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                "bytes_per_batch": int(words[kname] * 8 * n * batch), "ms_per_batch": bridge[kname]["ms_per_batch"],
                "note": "the bridge kernel with the largest share of this leg; all launches of the kind together; "
                        "PMC of these kernels: profiles/r04/v17_mpi_pmc.txt"}
    bridge_ms = sum(v["ms_per_batch"] for k, v in kernels.items() if k.startswith("bridge_"))
    ctx.set_overlap(restore_overlap)                 # the caller's setting (--lanes) back: the legs behind this one run on the shared context
    return {"shape": "n=2^%d, q=2^%d (W=%d words), dimA/dimB/dimP=%d/%d/%d, batch %d" % (ctx.logn, logq, W, dimA, dimB, dimP, batch),
            "ms_per_batch": round(ms, 3), "he_mul_per_s": round(batch / (ms * 1e-3), 1),
            "he_mul_plus_he_rescale_per_s": round(batch / (ms_rs * 1e-3), 1),
            "he_mul_then_he_rescale_two_calls_per_s": round(batch / (ms_rs_two * 1e-3), 1),
            "lanes": {"default": default_lanes, "this_run": lanes_this_run, "one_lane_ms_per_batch": round(ms_one, 3), "one_lane_he_mul_per_s": round(batch / (ms_one * 1e-3), 1),
                      "gain_this_device": round(ms_one / ms - 1, 4), "clocks": clocks or None,
                      "note": "`default` = the lanes the library picked by itself at this shape (gpq_last_lanes after a call under gpq_set_overlap(ctx, -1): two while the peer's "
                              "per-group workspace costs <= 16 GiB).  With two lanes every other launch group (32 ciphertexts) runs on a second internal stream through a peer "
                              "context.  The gain is launch tails and depends on the device at this shape (driver's round-4 box +0.15 %, builder's boxes +3.5 ... +3.9 %; never a "
                              "loss); the kernel breakdown below is a profiled ONE-lane pass, so its kernel times add up to the one-lane figure"},
            "bridge_ms_per_batch": round(bridge_ms, 3), "core_ms_per_batch": round(sum(v["ms_per_batch"] for k, v in kernels.items() if not k.startswith("bridge_")), 3),
            "bridge_algo_bytes_per_he_mul": int(sum(words.values()) * 8 * n), "kernels": kernels, "roofline": roof, "_check": check}


def he_swk_mpi_rate(torch, gpqhe_amd, batch=64, iters=3, logn=17, logq=835, checked="he_swk_mpi_level"):
    """BASELINE configs[4] as the function the reference runs: whole he_swk (src/he-automorphism.c:40-85 -- rns_decompose of d1, the key-switch
    inner product over dimB limbs, CRT, exact division by P, + d0, centring) at n = 2^17, q = 2^835 on device big slabs, batch 64."""
    W = logq // 64 + 1
    dimP = (logq + 1 + logn) // 59 + 1                                   # hectx.dim, src/precomp.c:401
    probe = gpqhe_amd.PolyContext(logn, dimP)
    P = 1
    for p in probe.p[:dimP]:
        P *= p
    probe.close()
    dimB = (logq + 1 + (P << logq).bit_length() + logn) // 59 + 1        # src/he-automorphism.c:52
    ctx = gpqhe_amd.PolyContext(logn, dimB)
    n = ctx.n
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1717)

    def centred():
        big = torch.randint(-(1 << 62), 1 << 62, (batch, W, n), dtype=torch.int64, device="cuda", generator=gen)
        top = logq - 2 - 64 * (W - 1)
        big[:, W - 1] = torch.randint(-(1 << top), 1 << top, (batch, n), dtype=torch.int64, device="cuda", generator=gen)
        return big.reshape(-1).contiguous()

    d0, d1 = centred(), centred()
    swk0, swk1 = rand_slab(torch, ctx, dimB, 1, gen), rand_slab(torch, ctx, dimB, 1, gen)
    o0, o1 = torch.empty_like(d0), torch.empty_like(d0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:      # steady clock (see ntt_rate)
        ctx.he_swk(o0, o1, d0, d1, swk0, swk1, W, logq, dimB, dimP)
        torch.cuda.synchronize()
    default_lanes = ctx.last_lanes()             # (the warm-up calls ran under the library's own choice)
    t = gpqhe_amd.StreamTimer()
    res = {}
    ran = {}
    for lanes in (1, 0):
        ctx.set_overlap(bool(lanes))
        ctx.he_swk(o0, o1, d0, d1, swk0, swk1, W, logq, dimB, dimP)
        ran[lanes] = ctx.last_lanes()
        t.start()
        for _ in range(iters):
            ctx.he_swk(o0, o1, d0, d1, swk0, swk1, W, logq, dimB, dimP)
        t.stop()
        res[lanes] = t.elapsed_ms() / iters
    ctx.profile(True)                            # one-lane pass with HIP events around every launch for the breakdown (gpq_profile takes no lane)
    for _ in range(iters):
        ctx.he_swk(o0, o1, d0, d1, swk0, swk1, W, logq, dimB, dimP)
    torch.cuda.synchronize()
    ctx.profile(False)
    prof = ctx.profile_collect()
    ctx.set_overlap(-1)
    # algorithmic words per coefficient, each slab read or written once: rns_decompose of d1 (W in, dimB out); the key switch (SURVEY.md 8d Stage B:
    # 3 reads + 2 writes per limb, here split over its three kernels); the tail: c0hat, c1hat (2 dimB) + d0 (W) in, c0, c1 (2 W) out
    words = {"bridge_decompose": W + dimB, "strided_fwd": 2 * dimB, "keyswitch_mid": 5 * dimB, "strided_inv": 4 * dimB,
             "bridge_tail_stream": 2 * dimB + 3 * W, "bridge_relin_tail_direct": 2 * dimB + 3 * W}
    total_ms = sum(v[0] for v in prof.values())
    kernels = {}
    for name, (kms, kcnt) in prof.items():
        per_call = kms / iters
        rec = {"ms_per_batch": round(per_call, 4), "launches_per_batch": round(kcnt / iters, 2), "share": round(kms / total_ms, 3)}
        if name in words:
            rec["algo_words_per_coefficient"] = words[name]
            rec["algo_GBps"] = round(words[name] * 8 * n * batch / (per_call * 1e-3) / 1e9, 1)
            rec["hbm_frac"] = round(rec["algo_GBps"] / HBM_PEAK_GBS, 4)
        kernels[name] = rec
    check = None
    if checked:
        ctx.he_swk(o0, o1, d0, d1, swk0, swk1, W, logq, dimB, dimP)
        torch.cuda.synchronize()
        group, per = 32, W * n
        picks = [0, batch - 1] if batch > group else [0]            # first polynomial pair of the first launch group, last of the last
        host = gpqhe_amd.to_host
        k0h, k1h = host(swk0), host(swk1)
        check = {"leg": checked, "picks": picks, "names": ("c0", "c1"),
                 "tasks": [dict(kind="he_swk", logn=logn, dimP=dimP, dimB=dimB, W=W, logq=logq, swk0=k0h, swk1=k1h,
                                d0=host(d0[k * per:(k + 1) * per]), d1=host(d1[k * per:(k + 1) * per])) for k in picks],
                 "got": [{nm: host(v[k * per:(k + 1) * per]) for nm, v in zip(("c0", "c1"), (o0, o1))} for k in picks]}
    ctx.close()
    ms = res[1]
    whole = (W + 2 * dimB) + 5 * dimB + (2 * dimB + 3 * W)          # decompose out + key switch + tail, the slabs between them counted once each way
    return {"shape": "n=2^%d, q=2^%d (W=%d words), dimB/dimP=%d/%d, batch %d: BASELINE configs[4] on one GPU" % (logn, logq, W, dimB, dimP, batch),
            "ms_per_batch": round(ms, 3), "he_swk_per_s": round(batch / (ms * 1e-3), 1),
            "lanes": {"default": default_lanes, "this_run": ran[1], "one_lane_ms_per_batch": round(res[0], 3), "one_lane_he_swk_per_s": round(batch / (res[0] * 1e-3), 1),
                      "gain_this_device": round(res[0] / ms - 1, 4)},
            "bridge_ms_per_batch": round(sum(v["ms_per_batch"] for k, v in kernels.items() if k.startswith("bridge_")), 3),
            "core_ms_per_batch": round(sum(v["ms_per_batch"] for k, v in kernels.items() if not k.startswith("bridge_")), 3),
            "algo_bytes_per_he_swk": int(whole * 8 * n), "algo_GBps": round(whole * 8 * n * batch / (ms * 1e-3) / 1e9, 1),
            "hbm_frac": round(whole * 8 * n * batch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "kernels": kernels, "_check": check}


def run_checks(out, checks):
    """The restated reference (oracle/expect.py: bigint_ref.he_mul / he_swk, the C oracle's limb loops + Python integers for everything libgcrypt
    does) on the ciphertexts the whole-function legs picked at their own launch shapes, all in parallel worker processes after every timed leg
    has finished; writes `bit_exact_vs_restated_reference` into each leg.  The oracle is the checker here, never the thing measured."""
    from oracle import expect
    checks = [c for c in checks if c]
    if not checks:
        return
    t0 = time.perf_counter()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    want = expect.expect_many([t for c in checks for t in c["tasks"]], workers=max(1, min(8, cores)))
    dt = time.perf_counter() - t0
    i = 0
    for c in checks:
        bad = []
        for k, got in zip(c["picks"], c["got"]):
            for nm in c["names"]:
                if not np.array_equal(got[nm], want[i][nm]):
                    bad.append("%s of ciphertext %d" % (nm, k))
            i += 1
        out[c["leg"]]["bit_exact_vs_restated_reference"] = {
            "ok": not bad, "ciphertexts": c["picks"], "outputs": list(c["names"]), "mismatches": bad,
            "how": "first and last ciphertext of each launch group of 32 at this leg's own launch shape (lanes on), every coefficient, against "
                   "oracle/bigint_ref (src/he-mult.c:88-156 / src/he-automorphism.c:40-85, src/he-rescale.c:33-54 restated); outside the timed regions; "
                   "%.0f s of CPU for all checked legs together" % dt}


def keyswitch_n17_rate(torch, gpqhe_amd, batch=64, iters=3):
    """BASELINE configs[4] shape: the key-switch inner product (he_swk loop, src/he-automorphism.c:59-67) at n = 2^17, 44 limbs."""
    logn, dim = 17, 44
    ctx = gpqhe_amd.PolyContext(logn, dim)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(17)
    x = rand_slab(torch, ctx, dim, batch, gen)
    e0, e1 = rand_slab(torch, ctx, dim, 1, gen), rand_slab(torch, ctx, dim, 1, gen)
    c0, c1 = torch.empty_like(x), torch.empty_like(x)
    ws = ctx.keyswitch_workspace(dim, batch)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:      # steady clock (see ntt_rate): this leg used to time three calls straight after context construction
        ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
        torch.cuda.synchronize()
    t = gpqhe_amd.StreamTimer()
    reps = []
    for _ in range(3):
        t.start()
        for _ in range(iters):
            ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
        t.stop()
        reps.append(t.elapsed_ms() / iters)
    ms = sorted(reps)[1]
    algo = 5 * dim * (8 << logn) * batch
    ctx.close()
    return {"shape": "n=2^17, 44 limbs, batch %d" % batch, "ms_per_batch": round(ms, 3), "keyswitch_per_s": round(batch / (ms * 1e-3), 1),
            "algo_GBps": round(algo / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "ms_per_batch_repetitions": [round(v, 3) for v in reps]}


def reference_signature_latency(timeout_s=150, logn=16, logq=850):
    """Batch-1 wall times through GPQHE's own signatures with real libgcrypt integers (the reference's only calling pattern): builds
    tests/c/mpi_host.c against the in-tree library and parses its `hemultime 16 850` report.  A child process (this one holds the GPU
    already; nothing is exec'ed over it).  None when gcc / libgcrypt's runtime are not there."""
    import re
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(root, "gpqhe_amd")
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(lib, "libgpqhe_hip_ctx.so")):
        return None
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "mpi_host")
        cc = ["gcc", "-O1", "-std=gnu11", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "mpi_host.c"), "-L", lib,
              "-lgpqhe_hip", "-lgpqhe_hip_ctx", "-l:libgcrypt.so.20", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
        try:
            if subprocess.run(cc, capture_output=True, text=True, timeout=120).returncode != 0:
                return None
            r = subprocess.run([exe, "hemultime", str(logn), str(logq)], capture_output=True, text=True, timeout=timeout_s)
        except (OSError, subprocess.TimeoutExpired):
            return None
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-300:]}
    t = r.stdout
    out = {"shape": "n=2^%d, q=2^%d, Delta=2^50 (%d levels), one ciphertext per call, libgcrypt integers in and out" % (logn, logq, logq // 50), "unit": "ms per call, p50 of 50",
           "note": "host-bound (16 conversion threads over scattered libgcrypt integers, PCIe): these move by +-30 % with the CPU load of the box; profiles/r03/v13_hemultime_resident.txt",
           "parity_lines": [ln.strip() for ln in t.splitlines() if ln.startswith(("key cache:", "resident polynomials:", "direct mpi access:"))]}

    def grab(key, pattern, cast=float):
        m = re.search(pattern, t)
        if m:
            out[key] = cast(m.group(1)) if m.lastindex == 1 else [cast(g) for g in m.groups()]
    grab("new_operands[he_mul,squaring,he_rescale]", r"50 calls each: he_mul p50 ([0-9.]+) .*?squaring p50 ([0-9.]+) .*?he_rescale p50 ([0-9.]+)")
    grab("chained[he_mul,he_mul(ct,ct,ct),he_rescale]", r"chained .*?he_mul p50 ([0-9.]+) .*?he_mul\(&ct, &ct, &ct\) p50 ([0-9.]+) .*?he_rescale of a product p50 ([0-9.]+)")
    grab("chained[he_add,he_addpt,he_neg]", r"additive calls in a chain .*?he_add p50 ([0-9.]+) .*?he_addpt p50 ([0-9.]+) .*?he_neg p50 ([0-9.]+)")
    grab("chained[he_rot,he_mulpt_new_plaintext]", r"per-diagonal calls in a chain: he_rot p50 ([0-9.]+) .*?new plaintext p50 ([0-9.]+)")
    grab("chained_he_copy_ct", r"he_copy_ct of a chained ciphertext p50 ([0-9.]+)")
    grab("he_inv_sequence_45_calls_ms[resident,every_call_uploads]", r"he_inv's call sequence.*?operands resident: ([0-9.]+) ms\n.*?every call uploads: ([0-9.]+) ms")
    grab("ladder_17x(he_mul+he_rescale)_ms[resident,every_call_uploads]", r"ladder of 17 x .*?operands resident: ([0-9.]+) ms\n(?:.*\n)?.*?every call uploads: ([0-9.]+) ms")
    grab("libgcrypt_host_he_add_ms", r"one he_add ([0-9.]+) ms")
    return out


def squaring_rate(torch, gpqhe_amd, ctx, batch, iters=5):
    """The RNS core of a SQUARING, he_mul(&ct, &ct, &ct, rlk) (src/he-algo.c:151; he_exp / he_inv square repeatedly): the tensor stage
    with both operands the same slabs runs two forward transforms instead of four (tensor_sq_mid8); key switch unchanged."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(151)
    a0, a1 = rand_slab(torch, ctx, DIM_A, batch, gen), rand_slab(torch, ctx, DIM_A, batch, gen)
    x = rand_slab(torch, ctx, DIM_B, batch, gen)
    e0, e1 = rand_slab(torch, ctx, DIM_B, 1, gen), rand_slab(torch, ctx, DIM_B, 1, gen)
    d = [torch.empty_like(a0) for _ in range(3)]
    c = [torch.empty_like(x) for _ in range(2)]
    wsA, wsB = ctx.tensor_workspace(DIM_A, batch), ctx.keyswitch_workspace(DIM_B, batch)

    def step():
        ctx.he_mul_tensor(d[0], d[1], d[2], a0, a1, a0, a1, DIM_A, wsA)
        ctx.he_keyswitch(c[0], c[1], x, e0, e1, DIM_B, wsB)

    for _ in range(2):
        step()
    t = gpqhe_amd.StreamTimer()
    t.start()
    for _ in range(iters):
        step()
    t.stop()
    ms = t.elapsed_ms() / iters
    return {"shape": "n=2^16, dimA=30 (2 forward + 3 inverse transforms per limb), dimB=45, batch %d" % batch, "ms_per_batch": round(ms, 3),
            "he_mul_per_s": round(batch / (ms * 1e-3), 1)}


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
           "bit_exact_vs_gpu": bool(ok),
           "build": "oracle/gpqhe_oracle.c, gcc -O2 (oracle/Makefile; the survey timed the reference itself at its own -Og: this port is the more generous baseline)"}
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


ROCPROF_NAMES = {   # how rocprofv3 names the kernels of the default n = 2^16 path (every limb wide-split; the last `true`: the non-temporal
    # instantiation gpq_set_nt_policy picks for launch groups beyond the Infinity Cache, which the bench's groups are)
    "tensor_mid": "gpq::tensor_mid8<gpq::TwW, 8, true>",
    "keyswitch_mid": "gpq::keyswitch_mid8x2<gpq::TwW, 8, true, true>",
    "strided_fwd": "gpq::strided_pass<8, 4, false, false, gpq::TwW, 8, true>",
    "strided_inv": "gpq::strided_pass<8, 4, true, false, gpq::TwW, 8, true>",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="ciphertext multiplications per GPU per step (weak scaling)")
    ap.add_argument("--total-batch", type=int, default=0, help="ciphertext multiplications per step over ALL GPUs, block-partitioned "
                    "(strong scaling; BASELINE configs[3] = 512); overrides --batch")
    ap.add_argument("--chunk", type=int, default=0, help="polynomials per fused launch group (0 = library default)")
    ap.add_argument("--limb-block", type=int, default=0, help="limbs per launch group (0 = library default: all)")
    ap.add_argument("--cpu-sample", type=int, default=16, help="ciphertexts the CPU baseline replays (0 = skip)")
    ap.add_argument("--no-ntt", action="store_true", help="skip the secondary legs run after the timed region (NTT GB/s, MPI-level he_mul, n=2^17 key switch)")
    ap.add_argument("--variant", default=None, help="dev: path of another build of libgpqhe_hip.so (make -C gpqhe_amd/csrc variant ...) for interleaved A/B timing")
    ap.add_argument("--quick", action="store_true", help="of the secondary legs keep only the clock / power sample, the VALU floor and the copy yardstick (tests)")
    ap.add_argument("--lanes", choices=("auto", "1", "2"), default="auto", help="RNS-core steps: gpq_set_overlap of the context (auto = the library's default; 1 for kernel traces "
                    "and PMC passes, whose durations should be those of kernels with nothing running beside them)")
    ap.add_argument("--host-pipeline-sub", type=int, default=2, help="with_host_scatter: ciphertexts per sub-batch of the pipelined (three-stream) variant; 0 = skip it")
    ap.add_argument("--no-affinity", action="store_true", help="leave the process's CPU mask alone (default: bind to the CPUs of the GPU's NUMA node before the first HIP call)")
    ap.add_argument("--no-check", action="store_true", help="skip the restated-reference check of the whole-function legs (about 40 s of CPU after the timed legs)")
    ap.add_argument("--streams", type=int, default=1, help="2: tensor stage and key-switch stage on separate HIP streams")
    ap.add_argument("--no-scatter-gather", action="store_true", help="N>1: skip the extra step that has the input slabs scattered from "
                    "rank 0 and the outputs gathered back inside the timed region (SURVEY.md 8d config 4)")
    ap.add_argument("--backend", default="nccl", help="N>1: nccl = slab transfers over RCCL/xGMI, one rank per GPU (the control plane -- barriers, MAX of "
                    "the timing -- is gloo either way); gloo = everything over gloo, to rehearse several ranks on one GPU")
    ap.add_argument("--sg-deadline", type=float, default=SG_DEADLINE_S, help="seconds the scatter/gather leg may take before every rank gives up (exit status %d)" % SG_FAILED_STATUS)
    ap.add_argument("--share-gpu", action="store_true", help="testing: with --backend nccl on a node with fewer GPUs than ranks, let ranks share "
                    "a device -- RCCL refuses that at the first transfer, which rehearses a fabric failure inside the scatter/gather leg")
    ap.add_argument("--sg-stall-rank", type=int, default=-1, help="testing: this rank never enters the scatter/gather leg (rehearses a stalled transfer)")
    return ap.parse_args(argv)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: this parent makes no torch / HIP call at all and starts the N rank
    processes as children (a process that has touched the GPU must never exec another program on this pool), one per GPU,
    rendezvous on 127.0.0.1.  Returns the first non-zero exit status of a rank (0 if all succeed)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print("bench.py: rank %d exited with status %d; stopping the other ranks" % (r, code), file=sys.stderr)
                for o in alive:
                    procs[o].terminate()                           # exactly the children started above
        time.sleep(0.05)
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:      # never run another rank count than the one asked for and report it as if it were
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))

    # Placement first: confine this rank to the CPUs of ITS GPU's NUMA node -- in-process, from sysfs, before torch is
    # imported and before the first HIP call (the runtime's helper threads inherit the mask; `with_host_scatter` first-touches its page-locked
    # staging memory on that node).  The single-process N = 1 run does the same for device 0, so `--gpus 1` under a launcher and the plain run are
    # one code path.  Never an error: `affinity` in the line says what was done.
    from gpqhe_amd import affinity
    try:
        nvis = len(affinity.visible_nodes(affinity.gpu_nodes(), os.environ))
    except (OSError, ValueError):
        nvis = 0
    aff = affinity.bind_to_gpu(local % nvis if (world > 1 and nvis) else 0) if not args.no_affinity else {"bound": False, "why_not": "--no-affinity"}

    import torch
    import gpqhe_amd
    from gpqhe_amd.dist import shard_range
    if args.variant:
        from gpqhe_amd import _native
        _native.use_variant(args.variant)

    dist = None
    ndev = torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            if local >= ndev and not args.share_gpu:
                sys.exit("bench.py: rank %d needs cuda:%d but this node shows %d device(s); one rank per GPU over RCCL "
                         "(--backend gloo rehearses several ranks on one GPU)" % (rank, local, ndev))
            torch.cuda.set_device(local % ndev)
            # Control plane (barriers, MAX of the timing, a few bytes) on gloo; the slabs of the scatter/gather leg on an RCCL group
            # that is only set up at its first transfer, inside that leg's watchdog: the path has no data-path collective
            # (SURVEY.md 8e), so a fabric problem must not be able to cost the compute-only line.
            from gpqhe_amd.dist import use_data_group
            dist.init_process_group(backend="gloo")
            use_data_group(dist.new_group(backend="nccl"))
        else:
            torch.cuda.set_device(local % ndev)
            dist.init_process_group(backend=args.backend)
    else:
        torch.cuda.set_device(0)
    dev_index = torch.cuda.current_device()

    if args.total_batch:
        if args.total_batch < world:
            sys.exit("bench.py: --total-batch %d is smaller than the %d ranks" % (args.total_batch, world))
        lo, hi = shard_range(args.total_batch, world, rank)
        B, total_batch, scaling = hi - lo, args.total_batch, "strong"
    else:
        B, total_batch, scaling = args.batch, args.batch * world, "weak"

    ctx = gpqhe_amd.PolyContext(LOGN, DIM_B)
    ctx.set_overlap({"auto": -1, "1": 0, "2": 1}[args.lanes])
    if args.chunk:
        ctx.set_chunk(args.chunk)
    if args.limb_block:
        ctx.set_limb_block(args.limb_block)
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
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0          # this rank's own K steps (what a slow device shows in); the reported time is the one behind the barrier
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    prof = ctx.profile_collect()
    ranks_seen, devices = 1, ["cuda:%d" % dev_index]
    mine = {"rank": rank, "device": "cuda:%d" % dev_index, "batch": B, "he_mul_per_s": round(B * args.steps / dt_own, 1),
            "ms_per_step": round(dt_own / args.steps * 1e3, 3), "affinity": aff}
    from gpqhe_amd.dist import gather_rank_records, summarize_ranks
    per_rank = gather_rank_records(mine)
    if dist is not None:
        from gpqhe_amd.dist import max_over_ranks
        dt = max_over_ranks(dt)
        ranks_seen = dist.get_world_size()
        devices = [r["device"] for r in per_rank]

    if rank == 0:
        total_he_mul = total_batch * args.steps
        value = total_he_mul / dt
        # Dominant kernel by accumulated device time.  tensor_mid8 (10 launches) and the inverse strided pass (20 launches: it runs for
        # both stages) are within a few per cent of each other, and which one leads changes with the device: among the kernels within
        # 10 % of the largest share the one with the LOWER fraction of the roofline is reported (the conservative reading, and the same
        # kernel from run to run); `kernels` carries the rate of every kernel either way.
        # strided kernels run for both stages: average units per launch from the launch mix
        chunk = min(B, args.chunk or 32)
        lb = args.limb_block or 0
        la, lbb = (min(lb, DIM_A), min(lb, DIM_B)) if lb else (DIM_A, DIM_B)
        na, nb = -(-DIM_A // la), -(-DIM_B // lbb)               # launch groups per stage: ceil(limbs / limb block)
        units = {"tensor_mid": DIM_A * chunk / na, "keyswitch_mid": DIM_B * chunk / nb,
                 "strided_fwd": (4 * DIM_A + 1 * DIM_B) * chunk / (na + nb),
                 "strided_inv": (3 * DIM_A + 2 * DIM_B) * chunk / (na + nb)}
        top = max(v[0] for v in prof.values())
        kname, (kms, kcnt) = min(((n, v) for n, v in prof.items() if v[0] >= 0.90 * top),
                                 key=lambda nv: KERNEL_LIMB_PASSES[nv[0]] * units[nv[0]] / (nv[1][0] / nv[1][1]))
        kernels = {}
        for name, (ms, cnt) in prof.items():
            avg_ms = ms / cnt
            byts = KERNEL_LIMB_PASSES[name] * units.get(name, 0) * (8 << LOGN)
            kernels[name] = {"avg_ms": round(avg_ms, 4), "launches": int(cnt), "share": round(ms / sum(v[0] for v in prof.values()), 3),
                             "algo_GBps": round(byts / (avg_ms * 1e-3) / 1e9, 1)}
        kavg = kms / kcnt
        kbytes = KERNEL_LIMB_PASSES[kname] * units[kname] * (8 << LOGN)
        achieved = kbytes / (kavg * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(kname, chunk)
        out = {
            "metric": "ciphertext he_mul/sec (RNS core: tensor 30 limbs + key-switch 45 limbs), N=2^16",
            "value": round(value, 2), "unit": "he_mul/s", "n_gpus": world, "ranks_seen": ranks_seen, "devices": devices,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "he_mul RNS core, n=2^16, dimA=30, dimB=45, %s; he_rescale has no RNS-domain work in the reference"
                                   % ("batch=%d ciphertexts per GPU (BASELINE configs[2])" % B if scaling == "weak" else
                                      "batch=%d ciphertexts block-partitioned over %d GPU(s) (BASELINE configs[3])" % (total_batch, world)),
                       "batch_per_gpu": B, "total_batch": total_batch, "chunk": chunk,
                       "parallelism": "ciphertext-per-GPU x%d, no data-path collective" % world,
                       "launcher": "torchrun / external" if os.environ.get("TORCHELASTIC_RUN_ID") else ("self-launched ranks" if world > 1 else "single process"),
                       "backend": (None if world == 1 else "gloo (control plane) + nccl/RCCL (slab transfers)" if args.backend == "nccl"
                                   else args.backend)},
            "roofline": {"bound": "hbm", "kernel": kname, "kernel_rocprof": ROCPROF_NAMES.get(kname, kname),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": int(kbytes), "avg_launch_ms": round(kavg, 4)},
            "he_mul_e2e": {"algo_bytes_per_he_mul": ALGO_BYTES_PER_HE_MUL,
                           "achieved_GBps_per_gpu": round(ALGO_BYTES_PER_HE_MUL * value / world / 1e9, 1),
                           "hbm_frac_per_gpu": round(ALGO_BYTES_PER_HE_MUL * value / world / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels": kernels,
            # every rank's OWN rate over the same K steps (before the closing barrier), beside the aggregate that the MAX of the wall time gives:
            # a slow device (the pool's parts differ by +-4 %) or a badly placed rank shows here, not only as a lower `value`
            "per_rank": summarize_ranks(per_rank),
            "affinity": aff,
            # BASELINE's metric is "he_mul/sec + NTT GB/s": the second half and the whole-function rate as TOP-LEVEL scalars (filled in by the legs
            # below; None when a leg was not run: N > 1, --no-ntt)
            "ntt_GBps": None, "ntt_hbm_frac": None, "he_mul_whole_per_s": None, "he_mul_plus_he_rescale_whole_per_s": None,
        }
        if world == 1 and args.cpu_sample > 0:
            s = min(args.cpu_sample, B)
            host_in = [gpqhe_amd.to_host(v[: s * DIM_A * ctx.n]) for v in (a0, a1, b0, b1)] + \
                      [gpqhe_amd.to_host(x[: s * DIM_B * ctx.n]), gpqhe_amd.to_host(e0), gpqhe_amd.to_host(e1)]
            gpu_out = [gpqhe_amd.to_host(v[: s * DIM_A * ctx.n]) for v in (d0, d1, d2)] + \
                      [gpqhe_amd.to_host(v[: s * DIM_B * ctx.n]) for v in (c0, c1)]
            out["cpu_baseline"] = cpu_baseline(ctx, host_in, gpu_out, s)
        if world == 1 and not args.no_ntt:
            pw = power_state(torch, step)
            if pw is not None:
                out["power"] = pw
                vi = valu_issue(value / world, pw["sclk_MHz"])
                vf = valu_floor(value / world, pw["sclk_MHz"])
                if vf is not None:
                    out["valu_floor"] = vf
                    if "frac" in vf:                      # what the evidence says binds the headline kernel; the HBM figures stay (north_star's declared roofline)
                        out["roofline"]["bound"] = "valu_issue"
                        out["roofline"]["declared_bound"] = "hbm"
                        out["roofline"]["valu_floor_frac"] = vf["frac"]
                        # ONE block with the three denominators the same measurement can be put over: they differ by what they
                        # charge to "issue" -- and the last line says which term is the largest.  The two probe-derived ones are only given when
                        # the probe was evaluated on THESE kernel headers (a stale probe is omitted, never shipped with a flag).
                        den = {"four_cycles_per_wave_instruction_at_this_clock": vf["frac"]}
                        fresh = vi is not None and "error" not in vi and not vi.get("stale") and vi.get("frac_of_probe_cycles_per_inst")
                        if fresh:
                            den["own_mix_probe_cycles_per_instruction_at_this_clock"] = vi["frac_of_probe_cycles_per_inst"]
                            den["own_mix_probe_at_its_uncapped_clock"] = vi["frac_of_valu_issue_rate"]
                            clk = vi["bound_source"].get("probe_sclk_MHz")
                            if clk:
                                den["clock_this_run_over_probe_clock"] = round(pw["sclk_MHz"] / clk, 3)
                            den["reading"] = ("the core issues at %.2f of the 4-cycle convention and %.2f of what its own instruction mix reaches per cycle with no memory "
                                              "traffic; against that probe at the clock it runs at un-capped the core is at %.2f: the package power cap (this run %d MHz at %.0f W) "
                                              "is the largest single term, not issue slots" % (den["four_cycles_per_wave_instruction_at_this_clock"],
                                                                                               den["own_mix_probe_cycles_per_instruction_at_this_clock"],
                                                                                               den["own_mix_probe_at_its_uncapped_clock"], pw["sclk_MHz"], pw["package_W"]))
                            vi.pop("stale", None)
                            out["valu_issue"] = vi
                        else:
                            den["probe_denominators"] = "omitted: tools/issue_probe has not been evaluated on the kernel headers of this tree (tools/valu_bound.py)"
                        vf["denominators"] = den
            cr = copy_rate(torch, gpqhe_amd)
            out["copy_rate"] = cr
            for rec in out["kernels"].values():               # every kernel's algorithmic rate against that of the best plain copy
                rec["of_copy_rate"] = round(rec["algo_GBps"] / cr["GBps"], 3)
        if world == 1 and not args.no_ntt and not args.quick:
            # The PCIe-inclusive rate of the same core on this one GPU (never `value`): the shard uploaded from page-locked host memory -- first-touched on
            # the GPU's own NUMA node, see `affinity` -- and the five output slabs downloaded, inside the timed region (at N > 1 every rank does this)
            out["with_host_scatter"] = host_scatter_step(torch, ctx, (a0, a1, b0, b1), x, (e0, e1), (wsA, wsB), min(B, 16), 1, barrier, repeats=3, pipeline_sub=args.host_pipeline_sub)
        if world == 1 and not args.no_ntt and args.quick:
            # the quick line still carries both halves of the metric: one NTT leg (the headline ring) and the whole function, timed briefly
            head = ntt_rate(torch, gpqhe_amd, 16, DIM_A, B, iters=8)
            out["ntt"] = [head]
            out["ntt_GBps"], out["ntt_hbm_frac"] = head["GBps"], head["hbm_frac"]
            del a0, a1, b0, b1, x, d0, d1, d2, c0, c1, wsA, wsB
            torch.cuda.empty_cache()
            whole = he_mul_mpi_rate(torch, gpqhe_amd, ctx, B, iters=3, brief=True, two_lanes=args.lanes != "1", restore_overlap={"auto": -1, "1": 0, "2": 1}[args.lanes], group=chunk)
            out["he_mul_mpi_level"] = whole
            out["he_mul_whole_per_s"], out["he_mul_plus_he_rescale_whole_per_s"] = whole["he_mul_per_s"], whole["he_mul_plus_he_rescale_per_s"]
        if world == 1 and not args.no_ntt and not args.quick:
            # NTT GB/s at the headline ring (n=2^16, 30 limbs) and at BASELINE configs[1] (n=2^15, 10 limbs)
            out["ntt"] = [ntt_rate(torch, gpqhe_amd, 16, DIM_A, B), ntt_rate(torch, gpqhe_amd, 15, 10, 64)]
            out["ntt_GBps"], out["ntt_hbm_frac"] = out["ntt"][0]["GBps"], out["ntt"][0]["hbm_frac"]     # n = 2^16 x 30 limbs x batch, forward + inverse
            del a0, a1, b0, b1, x, d0, d1, d2, c0, c1, wsA, wsB
            torch.cuda.empty_cache()
            out["ntt"].append(ntt_rate(torch, gpqhe_amd, 16, DIM_A, 4 * B))   # launch size matters: 4 GiB slab
            out["ntt"].append(ntt_rate(torch, gpqhe_amd, 15, 10, 2048))       # configs[1]'s ring at a launch that fills the chip (5 GiB)
            out["ntt"].append(ntt_rate(torch, gpqhe_amd, 17, 44, 64, iters=6))      # configs[4]'s ring and limb count (27 wide-split + 17 split limbs)
            checks = []
            out["he_mul_mpi_level"] = he_mul_mpi_rate(torch, gpqhe_amd, ctx, B, iters=5, checked="he_mul_mpi_level", sample_clocks=True, two_lanes=args.lanes != "1",
                                                      restore_overlap={"auto": -1, "1": 0, "2": 1}[args.lanes], group=chunk)   # BASELINE configs[2]: he_mul + he_rescale, batch 64
            checks.append(out["he_mul_mpi_level"].pop("_check"))
            out["he_mul_whole_per_s"] = out["he_mul_mpi_level"]["he_mul_per_s"]                              # src/he-mult.c:88-156 as a whole, device slabs
            out["he_mul_plus_he_rescale_whole_per_s"] = out["he_mul_mpi_level"]["he_mul_plus_he_rescale_per_s"]   # BASELINE configs[2]'s pair as one call
            out["keyswitch_n17"] = keyswitch_n17_rate(torch, gpqhe_amd)
            out["he_swk_mpi_level"] = he_swk_mpi_rate(torch, gpqhe_amd)             # BASELINE configs[4] as the reference's function, one GPU
            checks.append(out["he_swk_mpi_level"].pop("_check"))
            out["squaring_core"] = squaring_rate(torch, gpqhe_amd, ctx, B)
            rs = reference_signature_latency()
            if rs is not None:
                out["reference_signature"] = rs
            # The reference's own default parameters (tests/gpqhe.c:1296-1299: logn 14, q = 2^438, Delta = 2^50 -- the only shape GPQHE itself
            # ever runs; BASELINE.md: 1.12 s per he_mul on one CPU core): whole he_mul on device slabs at batch 64, and its own signature, batch 1
            ctx.close()
            c14 = gpqhe_amd.PolyContext(14, 24)
            rd = he_mul_mpi_rate(torch, gpqhe_amd, c14, 64, iters=25, logq=438, checked="reference_default", sample_clocks=True)
            c14.close()
            checks.append(rd.pop("_check"))
            rd = {k: rd[k] for k in ("shape", "ms_per_batch", "he_mul_per_s", "he_mul_plus_he_rescale_per_s", "he_mul_then_he_rescale_two_calls_per_s", "lanes", "bridge_ms_per_batch", "core_ms_per_batch")}
            rd["lanes"] = {k: v for k, v in rd["lanes"].items() if k != "note"}
            rd["reference_cpu_he_mul_per_s"] = round(1 / 1.12, 3)        # SURVEY.md 6 (survey probe, one core)
            sig = reference_signature_latency(logn=14, logq=438)
            if sig is not None:
                rd["reference_signature"] = {k: v for k, v in sig.items() if k not in ("note", "parity_lines")}
            out["reference_default"] = rd
            if not args.no_check:
                run_checks(out, checks)
    if dist is not None and not args.no_scatter_gather:
        # BASELINE configs[3] with the transfers inside the timed region: rank 0 owns the whole batch, every rank works on its
        # shard, results return to rank 0.  A root-GPU scatter is bound by one xGMI link per peer.  This leg is
        # secondary: a failure is reported in the line, and if the transfers stall a watchdog prints the (complete) compute-only
        # line and ends the ranks -- it can never cost the headline.
        import threading
        finished = threading.Event()
        progress = {"stage": "not started"}                  # what this rank was doing when the leg stalled or failed

        def abandon(reason):
            """The leg did not finish on this rank: rank 0 prints the (complete) compute-only line with the reason, then EVERY rank
            leaves with a non-zero status -- a stalled or failed transfer must never look like success to launch_ranks / the driver."""
            if rank == 0:
                out["with_scatter_gather"] = {"error": reason, "rank": rank, "stage": progress["stage"],
                                              "note": "the compute-only figures of this line are complete"}
                print(json.dumps(out), flush=True)
            else:
                print("bench.py: rank %d gives up on the scatter/gather leg at stage %r: %s" % (rank, progress["stage"], reason), file=sys.stderr, flush=True)
            os._exit(SG_FAILED_STATUS)

        def give_up():
            if not finished.is_set():
                abandon("no result within %g s" % args.sg_deadline)

        watchdog = threading.Timer(args.sg_deadline, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            if rank == args.sg_stall_rank:
                progress["stage"] = "stalled on purpose (--sg-stall-rank)"
                threading.Event().wait()
            # SURVEY.md 8e's alternative first: every rank sources its own shard from page-locked host memory (PCIe, all GPUs in parallel)
            hs = host_scatter_step(torch, ctx, (a0, a1, b0, b1), x, (e0, e1), (wsA, wsB), min(total_batch // world, 16), world, barrier, progress,
                                   repeats=2, pipeline_sub=args.host_pipeline_sub)
            sg = scatter_gather_step(torch, dist, ctx, (a0, a1, b0, b1), x, (e0, e1), (wsA, wsB), min(total_batch // world, 16), world, rank, barrier, progress)
        except Exception as exc:           # noqa: BLE001
            # the peers may be inside a transfer with this rank: no barrier any more (it could never complete); their own watchdogs
            # (or launch_ranks, which stops the other ranks at the first non-zero exit) end them
            finished.set()
            abandon("%s: %s" % (type(exc).__name__, exc))
        finished.set()
        watchdog.cancel()
        if rank == 0:
            out["with_host_scatter"] = hs
            out["with_scatter_gather"] = sg
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def host_scatter_step(torch, ctx, ins, x, evk, wss, Bs, world, barrier, progress=None, repeats=1, pipeline_sub=0):
    """One step over Bs ciphertexts per rank with every rank sourcing ITS OWN shard from page-locked host memory and returning its five
    output slabs there, transfers inside the timed region: SURVEY.md 8e's alternative to the root-GPU scatter (which one 153 GB/s xGMI
    link per peer bounds) -- all GPUs load over their own PCIe links in parallel, no GPU-to-GPU traffic at all.  Needs no process group
    beyond the barrier and the MAX of the timing."""
    from gpqhe_amd.dist import max_over_ranks
    a0, a1, b0, b1 = ins
    per_a, per_b = DIM_A * ctx.n, DIM_B * ctx.n
    progress = progress if progress is not None else {}
    progress["stage"] = "host scatter: staging the shard in page-locked memory"
    host_in = [v[: Bs * per_a].cpu().pin_memory() for v in (a0, a1, b0, b1)] + [x[: Bs * per_b].cpu().pin_memory()]
    host_out = [torch.empty(Bs * per_a, dtype=torch.int64).pin_memory() for _ in range(3)] + [torch.empty(Bs * per_b, dtype=torch.int64).pin_memory() for _ in range(2)]
    dev_in = [torch.empty(h.numel(), dtype=torch.int64, device="cuda") for h in host_in]
    o = [torch.empty(Bs * per_a, dtype=torch.int64, device="cuda") for _ in range(3)] + [torch.empty(Bs * per_b, dtype=torch.int64, device="cuda") for _ in range(2)]
    times = []
    for rep in range(repeats):          # (the first pass also pays the first use of the freshly pinned pages; every pass is listed)
        progress["stage"] = "host scatter: barrier before the uploads (pass %d)" % rep
        barrier()
        t1 = time.perf_counter()
        progress["stage"] = "host scatter: uploads, compute, downloads (pass %d)" % rep
        for d, h in zip(dev_in, host_in):
            d.copy_(h, non_blocking=True)
        ctx.he_mul_tensor(o[0], o[1], o[2], dev_in[0], dev_in[1], dev_in[2], dev_in[3], DIM_A, wss[0])
        ctx.he_keyswitch(o[3], o[4], dev_in[4], evk[0], evk[1], DIM_B, wss[1])
        for h, d in zip(host_out, o):
            h.copy_(d, non_blocking=True)
        progress["stage"] = "host scatter: barrier after the downloads (pass %d)" % rep
        barrier()
        dt = time.perf_counter() - t1
        times.append(max_over_ranks(dt) if world > 1 else dt)
    dt = min(times)
    moved = (4 * per_a + per_b + 3 * per_a + 2 * per_b) * 8 * Bs
    # the results that came back over PCIe are the ones a resident run computes
    ref = [torch.empty_like(t) for t in o]
    ctx.he_mul_tensor(ref[0], ref[1], ref[2], a0[: Bs * per_a], a1[: Bs * per_a], b0[: Bs * per_a], b1[: Bs * per_a], DIM_A, wss[0])
    ctx.he_keyswitch(ref[3], ref[4], x[: Bs * per_b], evk[0], evk[1], DIM_B, wss[1])
    torch.cuda.synchronize()
    ref_host = [r.cpu() for r in ref]
    same = bool(all(torch.equal(h, r) for h, r in zip(host_out, ref_host)))
    # The same as a three-stage PIPELINE over sub-batches -- upload of sub-batch k+1, compute of k and download of k-1 on three streams tied by events: the
    # link is full duplex, so uploads and downloads overlap each other and the compute ("overlap copies with compute on separate HIP streams").
    piped = None
    if pipeline_sub and Bs % pipeline_sub == 0 and Bs // pipeline_sub >= 2:
        progress["stage"] = "host scatter: pipelined passes"
        nsub = Bs // pipeline_sub
        s_up, s_cmp, s_dn = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        for t in o:
            t.zero_()
        for h in host_out:
            h.zero_()
        ptimes = []
        for rep in range(repeats):
            barrier()
            t1 = time.perf_counter()
            for k in range(nsub):
                ia, ib = slice(k * pipeline_sub * per_a, (k + 1) * pipeline_sub * per_a), slice(k * pipeline_sub * per_b, (k + 1) * pipeline_sub * per_b)
                with torch.cuda.stream(s_up):
                    for j in range(4):
                        dev_in[j][ia].copy_(host_in[j][ia], non_blocking=True)
                    dev_in[4][ib].copy_(host_in[4][ib], non_blocking=True)
                    up = torch.cuda.Event(); up.record()
                with torch.cuda.stream(s_cmp):
                    s_cmp.wait_event(up)
                    ctx.he_mul_tensor(o[0][ia], o[1][ia], o[2][ia], dev_in[0][ia], dev_in[1][ia], dev_in[2][ia], dev_in[3][ia], DIM_A, wss[0])
                    ctx.he_keyswitch(o[3][ib], o[4][ib], dev_in[4][ib], evk[0], evk[1], DIM_B, wss[1])
                    done = torch.cuda.Event(); done.record()
                with torch.cuda.stream(s_dn):
                    s_dn.wait_event(done)
                    for j in range(3):
                        host_out[j][ia].copy_(o[j][ia], non_blocking=True)
                    for j in (3, 4):
                        host_out[j][ib].copy_(o[j][ib], non_blocking=True)
            barrier()                                           # (torch.cuda.synchronize(): all three streams)
            dtp = time.perf_counter() - t1
            ptimes.append(max_over_ranks(dtp) if world > 1 else dtp)
        dtp = min(ptimes)
        piped = {"sub_batch": pipeline_sub, "equals_resident_run": bool(all(torch.equal(h, r) for h, r in zip(host_out, ref_host))), "he_mul_per_s": round(Bs * world / dtp, 1), "ms": round(dtp * 1e3, 2), "GBps_per_gpu_both_directions": round(moved / dtp / 1e9, 1),
                 "ms_every_pass": [round(t * 1e3, 2) for t in ptimes]}
    progress["stage"] = "host scatter: done"
    return {"batch_per_gpu": Bs, "he_mul_per_s": round(Bs * world / dt, 1), "ms": round(dt * 1e3, 2), "bytes_per_gpu_over_pcie": moved,
            "GBps_per_gpu": round(moved / dt / 1e9, 1), "equals_resident_run": same, "ms_every_pass": [round(t * 1e3, 2) for t in times], "pipelined": piped,
            "how": "every rank uploads its own shard from page-locked host memory and downloads its five output slabs, inside the timed region (SURVEY.md 8e)"}


def scatter_gather_step(torch, dist, ctx, ins, x, evk, wss, Bs, world, rank, barrier, progress=None):
    """One step over Bs ciphertexts per rank with the transfers timed: rank 0 holds the Bs*world inputs, every rank receives
    its shard (grouped isend/irecv = one RCCL group over xGMI), computes, and returns its five output slabs to rank 0."""
    from gpqhe_amd.dist import scatter_slab, gather_slab, max_over_ranks
    a0, a1, b0, b1 = ins
    dev = torch.device("cuda", torch.cuda.current_device())
    per_a, per_b = DIM_A * ctx.n, DIM_B * ctx.n
    full_in = [torch.cat([v[: Bs * per_a]] * world) if rank == 0 else None for v in (a0, a1, b0, b1)]
    full_x = torch.cat([x[: Bs * per_b]] * world) if rank == 0 else None
    progress = progress if progress is not None else {}
    progress["stage"] = "barrier before the scatter"
    barrier()
    t1 = time.perf_counter()
    sa = []
    for i, f in enumerate(full_in):
        progress["stage"] = "scatter of input slab %d of 5 (rank 0 -> all)" % (i + 1)
        sa.append(scatter_slab(f, per_a, Bs * world, 0, dev))
    progress["stage"] = "scatter of input slab 5 of 5 (rank 0 -> all)"
    sx = scatter_slab(full_x, per_b, Bs * world, 0, dev)
    progress["stage"] = "compute"
    o = [torch.empty_like(sa[0]) for _ in range(3)] + [torch.empty_like(sx) for _ in range(2)]
    ctx.he_mul_tensor(o[0], o[1], o[2], sa[0], sa[1], sa[2], sa[3], DIM_A, wss[0])
    ctx.he_keyswitch(o[3], o[4], sx, evk[0], evk[1], DIM_B, wss[1])
    back = []
    for i in range(5):
        progress["stage"] = "gather of output slab %d of 5 (all -> rank 0)" % (i + 1)
        back.append(gather_slab(o[i], per_a if i < 3 else per_b, Bs * world, 0))
    progress["stage"] = "barrier after the gather"
    barrier()
    progress["stage"] = "done"
    dsg = max_over_ranks(time.perf_counter() - t1)
    moved = (4 * per_a + per_b + 3 * per_a + 2 * per_b) * 8 * Bs * (world - 1)
    sg = {"batch_per_gpu": Bs, "he_mul_per_s": round(Bs * world / dsg, 1), "ms": round(dsg * 1e3, 2), "bytes_over_links": moved,
          "GBps_root": round(moved / dsg / 1e9, 1)}
    if rank == 0:   # the shards are copies of the first ciphertexts: every shard's result equals rank 0's own
        sg["shards_identical"] = bool(all(torch.equal(back[i][: o[i].numel()], back[i][r * o[i].numel(): (r + 1) * o[i].numel()])
                                          for i in range(5) for r in range(1, world)))
    return sg


if __name__ == "__main__":
    main()
