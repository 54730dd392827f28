"""Several devices from ONE plain-C process through the C ABI alone (tests/c/shard_host.c): a batch of independent ciphertext
multiplications (src/he-mult.c:116-138 + :58-66: no cross-ciphertext state) is block-partitioned over a list of devices, one
context / stream / buffer set per shard, every shard launched before any is waited for.  The box has one GPU, so the list is
"0" (one shard) and "0,0,0" (three shards with their own contexts and streams on that GPU, ragged 3 + 2 + 2); on an 8-GPU node the
same binary takes "0,1,...,7".  Every ciphertext's five outputs must equal the oracle's, whatever the sharding."""
import os
import subprocess

import numpy as np
import pytest

from oracle.oracle import fnv

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shard_host(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shard") / "shard_host")
    lib_dir = os.path.join(ROOT, "gpqhe_amd")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "shard_host.c"),
                           "-L", lib_dir, "-lgpqhe_hip", "-pthread", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


@pytest.mark.timeout(600)
@pytest.mark.parametrize("logn,dim_a,dim_b,batch", [(13, 3, 4, 7), (16, 30, 45, 3)])
def test_batch_sharded_over_a_device_list_from_plain_c(shard_host, oracle_ctx, logn, dim_a, dim_b, batch):
    o = oracle_ctx(logn, dim_b)
    n = o.n
    expect = {}
    ev = [o.gen(3000, dim_b), o.gen(3001, dim_b)]
    for k in range(batch):
        ins = [o.gen(1000 + 4 * k + i, dim_a) for i in range(4)]
        d = o.he_mul_tensor(*ins, dim_a)
        c = o.keyswitch(o.gen(2000 + k, dim_b), ev[0], ev[1], dim_b)
        expect[k] = [fnv(v) for v in list(d) + list(c)]
    for devs, extra in (("0", []), ("0,0,0", []), ("0", ["0", "2"]), ("0,0,0", ["0", "2"])):     # plain, and pipelined in sub-batches of 2 (gpq_stream_wait)
        res = subprocess.run([shard_host, str(logn), str(dim_a), str(dim_b), str(batch), devs] + extra, capture_output=True, text=True, timeout=500)
        assert res.returncode == 0, res.stderr
        lines = res.stdout.strip().split("\n")
        assert lines[0].split()[:3] == ["devices", "visible", lines[0].split()[2]] and lines[0].endswith("shards %d" % len(devs.split(",")))
        got = {}
        placed = [ln for ln in lines[1:] if ln.startswith("shard ")]       # one worker thread per shard, each reporting its placement
        assert len(placed) == len(devs.split(",")) and all("worker confined to" in ln for ln in placed)
        for ln in lines[1:]:
            if ln.startswith("shard "):
                continue
            f = ln.split()
            assert f[0] == "ct" and f[2] == "dev" and f[3] == "0"
            got[int(f[1])] = f[4:]
        assert sorted(got) == list(range(batch))
        for k in range(batch):
            assert got[k] == expect[k], (devs, extra, k)
    if logn == 16:
        assert expect[0] == ["99655f317c50d5c1", "e7658c5a00ed9eac", "12600b1bac18b63e", "466a17f24e0654d2", "578cf3337a189ad0"]   # SURVEY.md 8c, k = 0


@pytest.mark.timeout(900)
def test_eight_shards_with_their_own_contexts_on_one_device(shard_host, oracle_ctx):
    """BASELINE configs[3]'s layout rehearsed on the one GPU of the box: eight shards ("0,0,0,0,0,0,0,0" stands for "0,1,...,7"),
    each with its own context, stream and buffers, at the headline shape; 32 ciphertexts (4 per shard) carrying the inputs of
    ciphertext k mod 4, so four oracle evaluations price all of them.  The full batch of 512 (64 per shard) is the same command with
    512 in place of 32: profiles/r03/ keeps that run."""
    logn, dim_a, dim_b, batch, period = 16, 30, 45, 32, 4
    o = oracle_ctx(logn, dim_b)
    ev = [o.gen(3000, dim_b), o.gen(3001, dim_b)]
    expect = []
    for k in range(period):
        d = o.he_mul_tensor(*[o.gen(1000 + 4 * k + i, dim_a) for i in range(4)], dim_a)
        c = o.keyswitch(o.gen(2000 + k, dim_b), ev[0], ev[1], dim_b)
        expect.append([fnv(v) for v in list(d) + list(c)])
    res = subprocess.run([shard_host, str(logn), str(dim_a), str(dim_b), str(batch), "0,0,0,0,0,0,0,0", str(period)], capture_output=True, text=True, timeout=800)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().split("\n")
    assert lines[0].endswith("shards 8")
    assert sum(ln.startswith("shard ") for ln in lines) == 8          # eight worker threads, each placed before its first device call
    got = {int(f[1]): f[4:] for f in (ln.split() for ln in lines[1:] if ln.startswith("ct "))}
    assert sorted(got) == list(range(batch))
    for k in range(batch):
        assert got[k] == expect[k % period], k
