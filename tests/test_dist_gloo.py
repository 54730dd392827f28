"""The N>1 path on CPU: world_size 2 over gloo.  Independent ciphertexts are
block-partitioned over ranks, each rank runs the RNS core on its shard (here the
oracle stands in for the device op -- this is a test), outputs are gathered and
must equal the single-process result; timing is MAX-reduced as bench.py does."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpqhe_amd.dist import gather_slab, max_over_ranks, scatter_slab, shard_range, use_data_group

LOGN, DIM, BATCH = 7, 3, 5  # ragged on purpose: 5 ciphertexts over 2 ranks


def test_shard_range_partitions_every_batch():
    for batch in (0, 1, 5, 64, 511, 512):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(batch, world, r)
                assert 0 <= lo <= hi <= batch and hi - lo in (batch // world, batch // world + 1)
                cover += list(range(lo, hi))
            assert cover == list(range(batch))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, two_planes=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if two_planes:      # bench.py's arrangement: control plane = default group, slabs over a group of their own
            use_data_group(dist.new_group(backend="gloo"))
        from oracle.oracle import OracleCtx
        o = OracleCtx(LOGN, DIM)
        per = DIM * o.n
        full = None
        if rank == 0:
            full = [torch.from_numpy(np.concatenate([o.gen(10 * s + k, DIM) for k in range(BATCH)]).view(np.int64)) for s in range(4)]
        # the shard lands on the device asked for, on the source rank too (here: the host; on the GPU box: cuda)
        mine = [scatter_slab(full[s] if rank == 0 else None, per, BATCH, 0, torch.device("cpu") if s % 2 else None) for s in range(4)]
        assert all(m.device.type == "cpu" for m in mine)
        lo, hi = shard_range(BATCH, world, rank)
        outs = [np.empty((hi - lo) * per, dtype=np.uint64) for _ in range(3)]
        for k in range(hi - lo):
            d = o.he_mul_tensor(*[np.ascontiguousarray(m.numpy().view(np.uint64)[k * per:(k + 1) * per]) for m in mine], DIM)
            for out, v in zip(outs, d):
                out[k * per:(k + 1) * per] = v
        gathered = [gather_slab(torch.from_numpy(v.view(np.int64)), per, BATCH) for v in outs]
        slowest = max_over_ranks(1.0 + rank)
        if rank == 0:
            ok = True
            for k in range(BATCH):
                d = o.he_mul_tensor(*[np.ascontiguousarray(f.numpy().view(np.uint64)[k * per:(k + 1) * per]) for f in full], DIM)
                for g, v in zip(gathered, d):
                    ok = ok and np.array_equal(g.numpy().view(np.uint64)[k * per:(k + 1) * per], v)
            q.put((ok, slowest))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("two_planes", [False, True])
def test_two_rank_scatter_compute_gather(two_planes):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, two_planes)) for r in range(world)]
    for p in procs:
        p.start()
    ok, slowest = q.get(timeout=150)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert ok, "gathered multi-rank result differs from the single-process result"
    assert slowest == 2.0  # MAX over ranks of (1.0, 2.0)


def _placed_worker(rank, world, port, q, root):
    """what a bench.py rank does around its timed region, without a GPU: place itself (fake two-socket sysfs tree; the mask change is recorded, not
    applied), take its own time, gather every rank's record over the control plane, MAX the wall time"""
    from gpqhe_amd import affinity
    from gpqhe_amd.dist import gather_rank_records, summarize_ranks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    asked = []
    aff = affinity.bind_to_gpu(rank * 5, root, environ={}, setaffinity=asked.append, getaffinity=lambda: set(range(192)))   # devices 0 and 5: one per socket
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        own = 0.5 * (1 + rank)                                            # rank 1 is the slow device
        mine = {"rank": rank, "device": "cuda:%d" % (rank * 5), "batch": 64, "he_mul_per_s": round(64 / own, 1), "ms_per_step": own * 1e3, "affinity": aff,
                "asked_cpus": sorted(asked[0])[:1] + sorted(asked[0])[-1:] if asked else None}
        records = gather_rank_records(mine)
        wall = max_over_ranks(own)
        if rank == 0:
            q.put((summarize_ranks(records), wall))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_ranks_place_themselves_and_report_their_own_rates(tmp_path):
    """per-rank placement (each rank on the socket of ITS GPU) and per-rank rates beside the MAX-time aggregate, over gloo"""
    from tests.test_affinity import _tree
    root = _tree(tmp_path)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_placed_worker, args=(r, world, port, q, root)) for r in range(world)]
    for p in procs:
        p.start()
    summary, wall = q.get(timeout=150)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert wall == 1.0                                                    # the aggregate is priced on the slowest rank ...
    assert summary["he_mul_per_s_min"] == 64.0 and summary["he_mul_per_s_max"] == 128.0 and summary["sum_of_own_rates"] == 192.0   # ... and the line shows who that was
    r0, r1 = summary["ranks"]
    assert (r0["rank"], r1["rank"]) == (0, 1) and (r0["device"], r1["device"]) == ("cuda:0", "cuda:5")
    assert r0["affinity"]["bound"] and r0["affinity"]["numa_node"] == 0 and r0["asked_cpus"] == [0, 143]
    assert r1["affinity"]["bound"] and r1["affinity"]["numa_node"] == 1 and r1["asked_cpus"] == [48, 191]
