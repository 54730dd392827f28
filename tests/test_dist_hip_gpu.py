"""The N>1 path with the HIP kernels: two ranks share the one GPU of the box over gloo (RCCL wants one device per
rank; the partition, the transfers and the kernels are the same).  Rank 0 owns a ragged batch of independent
ciphertext multiplications (src/he-mult.c:116-138 + :58-66 carry no cross-ciphertext state), scatters the input
slabs, every rank runs gpq_he_mul_tensor + gpq_keyswitch on its shard through the C ABI, the outputs are gathered and
every ciphertext is compared bit for bit with the oracle.  Also: `bench.py --gpus 2` must start two ranks by itself."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, logn, dim_a, dim_b, batch, q):
    import datetime
    import traceback
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # a rank that dies must not leave its peer waiting for long: short collective timeout, and every failure is reported
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        _run_rank(rank, world, logn, dim_a, dim_b, batch, q, torch, dist)
    except BaseException:
        q.put(("error", rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def _run_rank(rank, world, logn, dim_a, dim_b, batch, q, torch, dist):
    import gpqhe_amd
    from gpqhe_amd.dist import gather_slab, scatter_slab, shard_range
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = gpqhe_amd.PolyContext(logn, dim_b)
    n = g.n
    per_a, per_b = dim_a * n, dim_b * n
    full, o = None, None
    if rank == 0:
        from oracle.oracle import OracleCtx
        o = OracleCtx(logn, dim_b)
        assert o.p == g.p
        full = [torch.from_numpy(np.concatenate([o.gen(100 * s + k, dim_a) for k in range(batch)]).view(np.int64)) for s in range(4)]
        full.append(torch.from_numpy(np.concatenate([o.gen(900 + k, dim_b) for k in range(batch)]).view(np.int64)))
        evk = [o.gen(7000, dim_b), o.gen(7001, dim_b)]
        ev = [torch.from_numpy(e.view(np.int64)).clone() for e in evk]
    else:
        ev = [torch.empty(per_b, dtype=torch.int64) for _ in range(2)]
    for e in ev:                                   # the key is replicated, not sharded
        dist.broadcast(e, 0)
    ev = [e.to(dev) for e in ev]
    mine = [scatter_slab(full[s] if rank == 0 else None, per_a, batch, 0, dev) for s in range(4)]
    x = scatter_slab(full[4] if rank == 0 else None, per_b, batch, 0, dev)
    lo, hi = shard_range(batch, world, rank)
    assert all(m.is_cuda and m.numel() == (hi - lo) * per_a for m in mine) and x.numel() == (hi - lo) * per_b
    d = [torch.empty_like(mine[0]) for _ in range(3)]
    c = [torch.empty_like(x) for _ in range(2)]
    g.he_mul_tensor(d[0], d[1], d[2], *mine, dim_a)            # HIP kernels through the C ABI
    g.he_keyswitch(c[0], c[1], x, ev[0], ev[1], dim_b)
    back = [gather_slab(v, per_a, batch, 0) for v in d] + [gather_slab(v, per_b, batch, 0) for v in c]
    if rank == 0:
        bad = []
        got = [b.cpu().numpy().view(np.uint64) for b in back]
        host = [f.numpy().view(np.uint64) for f in full]
        for k in range(batch):
            sa, sb = slice(k * per_a, (k + 1) * per_a), slice(k * per_b, (k + 1) * per_b)
            exp = list(o.he_mul_tensor(*[np.ascontiguousarray(h[sa]) for h in host[:4]], dim_a))
            exp += list(o.keyswitch(np.ascontiguousarray(host[4][sb]), evk[0], evk[1], dim_b))
            for name, gv, ev_, sl in zip(("d0", "d1", "d2", "c0", "c1"), got, exp, (sa, sa, sa, sb, sb)):
                if not np.array_equal(gv[sl], ev_):
                    bad.append((k, name))
        q.put(("done", 0, bad))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("logn,dim_a,dim_b,batch", [(13, 3, 4, 5), (16, 30, 45, 3)])
def test_two_ranks_run_the_hip_core_on_their_shards(logn, dim_a, dim_b, batch):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, logn, dim_a, dim_b, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        kind, who, what = q.get(timeout=300)
    finally:
        for p in procs:
            p.join(150)
            if p.is_alive():
                p.terminate()
    assert kind == "done", "rank %d failed:\n%s" % (who, what)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert what == [], "ciphertexts whose gathered result differs from the oracle: %r" % what


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 4` with WORLD_SIZE unset: the parent spawns four ranks (gloo: all on the one GPU here -- the rehearsal of the driver's
    first multi-GPU run), the line says n_gpus = ranks_seen = 4 and carries both transfer legs; --total-batch shards 9 as 3 + 2 + 2 + 2; every rank
    reports its placement and its OWN rate beside the MAX-time aggregate (round 6)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--total-batch", "9",
           "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--no-ntt"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=840, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["ranks_seen"] == 4 and out["devices"] == ["cuda:0"] * 4
    assert out["scaling"] == "strong" and out["config"]["total_batch"] == 9 and out["config"]["batch_per_gpu"] == 3
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    pr = out["per_rank"]
    assert [x["rank"] for x in pr["ranks"]] == [0, 1, 2, 3] and [x["batch"] for x in pr["ranks"]] == [3, 2, 2, 2]
    assert 0 < pr["he_mul_per_s_min"] <= pr["he_mul_per_s_max"] and all(isinstance(x["affinity"]["bound"], bool) for x in pr["ranks"])
    assert out["value"] <= pr["sum_of_own_rates"] * 1.001                 # the aggregate waits for the slowest rank: never more than the own rates add up to
    for key in ("ntt_GBps", "ntt_hbm_frac", "he_mul_whole_per_s", "he_mul_plus_he_rescale_whole_per_s"):
        assert key in out and out[key] is None                            # single-GPU legs: present, not run at N > 1
    sg = out["with_scatter_gather"]
    assert "error" not in sg and sg["shards_identical"] is True and sg["he_mul_per_s"] > 0
    # SURVEY.md 8e's alternative, rehearsed by the same ranks: every rank uploads its own shard from page-locked host memory and
    # downloads its results inside the timed region; what comes back over PCIe is what a resident run computes
    hs = out["with_host_scatter"]
    assert hs["equals_resident_run"] is True and hs["he_mul_per_s"] > 0 and hs["batch_per_gpu"] == 2
    assert hs["bytes_per_gpu_over_pcie"] == (7 * 30 + 3 * 45) * 65536 * 8 * hs["batch_per_gpu"]


@pytest.mark.timeout(600)
def test_a_stalled_scatter_gather_leg_is_a_failure_not_a_success():
    """One rank never enters the transfers: every rank gives up at the deadline, rank 0 still prints the complete compute-only
    line (with the error, its rank and the stage it was stuck in), and the exit status that reaches the caller is non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--total-batch", "4",
           "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--no-ntt", "--sg-stall-rank", "1", "--sg-deadline", "10"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540, cwd=ROOT)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["n_gpus"] == 2
    sg = out["with_scatter_gather"]
    assert "no result within" in sg["error"] and sg["rank"] == 0 and "scatter" in sg["stage"]


@pytest.mark.timeout(600)
def test_a_failing_rccl_group_costs_the_transfer_leg_only():
    """--backend nccl keeps barriers and timing on gloo and sets the RCCL group up at the first slab transfer.  Two ranks on the one
    GPU of this box are something RCCL refuses (or never completes): the compute-only line is still printed in full, the
    scatter/gather entry carries the error, the exit status is non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl", "--share-gpu", "--total-batch", "4",
           "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--no-ntt", "--sg-deadline", "40"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540, cwd=ROOT)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert "nccl" in out["config"]["backend"] and "gloo" in out["config"]["backend"]
    assert out["with_scatter_gather"]["error"] and "scatter" in out["with_scatter_gather"]["stage"]


@pytest.mark.timeout(400)
def test_rccl_first_contact_on_one_gpu():
    """RCCL doing something other than failing: one rank on cuda:0, control plane gloo, data plane
    `new_group(backend="nccl")` as bench.py builds it, in a CHILD started before this process has touched the GPU for it: an all_reduce, an
    all_gather, a grouped self send/recv of three 45-limb polynomials (the call shape of the scatter / gather leg), the library's own
    scatter_slab / gather_slab over that group.  No N > 1 number exists or is claimed (the driver's 8-GPU run is the first)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_first_contact.py"), str(_free_port()), "240"], env=env, capture_output=True, text=True,
                       timeout=330, cwd=ROOT)
    steps = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    names = [s["step"] for s in steps]
    assert r.returncode == 0 and names[-1] == "done", (r.returncode, names, r.stderr[-1500:])
    assert steps[0] == {"step": "groups", "data_backend": "nccl"}
    assert all(s.get("ok", True) for s in steps), steps
    assert names == ["groups", "all_reduce", "all_gather", "self_sendrecv", "scatter_gather", "max_over_ranks", "done"]
