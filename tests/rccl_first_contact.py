"""Child process of tests/test_dist_hip_gpu.py::test_rccl_first_contact_on_one_gpu (not a test module): the first time this code base lets RCCL
do anything but fail.  One rank on cuda:0, control plane gloo, data plane `new_group(backend="nccl")` exactly as bench.py sets it up; prints one
JSON line per step so that a hang names the step it hung in (a watchdog ends the process: nothing may wait on a fabric forever)."""
import datetime
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def say(step, **kw):
    print(json.dumps(dict(step=step, **kw)), flush=True)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    deadline = threading.Timer(float(sys.argv[2]) if len(sys.argv) > 2 else 90.0, lambda: (say("watchdog"), os._exit(3)))
    deadline.daemon = True
    deadline.start()
    import torch
    import torch.distributed as dist
    from gpqhe_amd.dist import data_backend, gather_slab, max_over_ranks, scatter_slab, use_data_group
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1, timeout=datetime.timedelta(seconds=60))
    group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=60))
    use_data_group(group)
    say("groups", data_backend=data_backend())
    t = torch.arange(1 << 20, dtype=torch.int64, device="cuda")
    want = int(t.sum().item())
    dist.all_reduce(t, group=group)                               # first RCCL collective: communicator set-up + one kernel
    torch.cuda.synchronize()
    say("all_reduce", ok=int(t.sum().item()) == want)
    out = torch.empty_like(t)
    dist.all_gather_into_tensor(out, t, group=group)
    torch.cuda.synchronize()
    say("all_gather", ok=bool(torch.equal(out, t)))
    # the slab transfers of the scatter / gather leg are grouped ncclSend / ncclRecv (dist.batch_isend_irecv): with one rank the only peer is
    # the rank itself -- a self send/recv inside one group call
    src = torch.randint(-(1 << 62), 1 << 62, (3 * 45 * 65536,), dtype=torch.int64, device="cuda")      # three 45-limb polynomials of the headline ring
    dst = torch.zeros_like(src)
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0, group=group), dist.P2POp(dist.irecv, dst, 0, group=group)]):
        w.wait()
    torch.cuda.synchronize()
    say("self_sendrecv", ok=bool(torch.equal(src, dst)), bytes=src.numel() * 8)
    # the library's own scatter / gather entry points over the nccl data group (one rank: a copy on the device, no fabric)
    mine = scatter_slab(src, 45 * 65536, 3, 0, torch.device("cuda", 0))
    back = gather_slab(mine, 45 * 65536, 3, 0)
    say("scatter_gather", ok=bool(torch.equal(back, src)) and mine.is_cuda)
    say("max_over_ranks", ok=max_over_ranks(1.5) == 1.5)
    dist.destroy_process_group()
    say("done")


if __name__ == "__main__":
    main()
