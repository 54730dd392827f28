"""Host placement of ranks / worker threads: HIP device -> KFD GPU node -> render node -> NUMA CPUs -> sched_setaffinity,
from sysfs alone.  Both implementations -- gpqhe_amd/affinity.py (bench.py's rank processes, before their first HIP call) and the library's
gpq_device_local_cpus / gpq_bind_thread_to_device (C hosts, tests/c/shard_host.c's worker threads) -- are driven over the same FAKE sysfs tree of
a two-socket, eight-GPU node, plus the degenerate trees (no topology, one memory domain, a render node the cgroup hides)."""
import ctypes as C
import os

import pytest

from gpqhe_amd import affinity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tree(tmp, gpus=8, numa=None, hidden=(), cpus_per_node=48):
    """nodes 0, 1 = CPU sockets (simd_count 0); nodes 2.. = GPUs, render minors 128.., four per socket"""
    root = str(tmp)
    base = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    for node in range(2):
        os.makedirs(os.path.join(base, str(node)))
        with open(os.path.join(base, str(node), "properties"), "w") as f:
            f.write("cpu_cores_count %d\nsimd_count 0\ndrm_render_minor -1\n" % cpus_per_node)
    os.makedirs(os.path.join(root, "dev/dri"), exist_ok=True)
    for g in range(gpus):
        node, minor = 2 + g, 128 + g
        os.makedirs(os.path.join(base, str(node)))
        with open(os.path.join(base, str(node), "properties"), "w") as f:
            f.write("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\ndrm_render_minor %d\nunique_id 1234\n" % (0x500 + g, minor))
        if g not in hidden:
            open(os.path.join(root, "dev/dri/renderD%d" % minor), "w").close()
        dev = os.path.join(root, "sys/class/drm/renderD%d/device" % minor)
        os.makedirs(dev)
        socket = g // 4 if numa is None else numa
        with open(os.path.join(dev, "numa_node"), "w") as f:
            f.write("%d\n" % socket)
        lo = max(socket, 0) * cpus_per_node
        with open(os.path.join(dev, "local_cpulist"), "w") as f:       # hyperthreads as a second range, like a real host
            f.write("%d-%d,%d-%d\n" % (lo, lo + cpus_per_node - 1, 2 * cpus_per_node + lo, 2 * cpus_per_node + lo + cpus_per_node - 1))
    return root


def test_parse_cpulist():
    assert affinity.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert affinity.parse_cpulist("") == set()


def test_python_mapping_over_a_two_socket_eight_gpu_tree(tmp_path):
    root = _tree(tmp_path)
    for dev in range(8):
        cpus, rep = affinity.cpus_of_gpu(dev, root, environ={})
        lo = (dev // 4) * 48
        assert cpus == set(range(lo, lo + 48)) | set(range(96 + lo, 96 + lo + 48)), rep
        assert rep["numa_node"] == dev // 4 and rep["render_minor"] == 128 + dev and rep["kfd_node"] == 2 + dev
    # visibility lists re-index the devices: HIP device 0 of this rank is the sixth GPU of the node
    cpus, rep = affinity.cpus_of_gpu(0, root, environ={"HIP_VISIBLE_DEVICES": "5,1"})
    assert rep["render_minor"] == 133 and rep["numa_node"] == 1
    cpus, rep = affinity.cpus_of_gpu(1, root, environ={"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "3,0"})
    assert rep["render_minor"] == 132                                   # ROCr layer first, then the HIP layer
    assert affinity.cpus_of_gpu(8, root, environ={})[0] == set()        # not there
    assert affinity.cpus_of_gpu(0, root, environ={"HIP_VISIBLE_DEVICES": "GPU-abcdef"})[0] == set()   # UUID form: gives up, no error


def test_python_bind_intersects_with_the_allowed_mask_and_never_raises(tmp_path):
    root = _tree(tmp_path)
    calls = []
    rep = affinity.bind_to_gpu(5, root, environ={}, setaffinity=calls.append, getaffinity=lambda: set(range(192)))
    assert rep["bound"] and rep["cpus"] == 96 and calls == [set(range(48, 96)) | set(range(144, 192))]
    calls.clear()
    rep = affinity.bind_to_gpu(5, root, environ={}, setaffinity=calls.append, getaffinity=lambda: {0, 1, 2, 50, 51})
    assert rep["bound"] and calls == [{50, 51}]                          # a cgroup's 5 CPUs: only the two on the GPU's socket
    calls.clear()
    rep = affinity.bind_to_gpu(5, root, environ={}, setaffinity=calls.append, getaffinity=lambda: {0, 1, 2})
    assert not rep["bound"] and not calls and "mask" in rep["why_not"]
    rep = affinity.bind_to_gpu(5, root, environ={}, setaffinity=calls.append, getaffinity=lambda: {50, 51})
    assert not rep["bound"] and not calls and "already" in rep["why_not"]
    # degenerate trees
    rep = affinity.bind_to_gpu(0, str(tmp_path / "nothing"), environ={}, setaffinity=calls.append, getaffinity=lambda: {0, 1})
    assert not rep["bound"] and "why_not" in rep and not calls
    one = _tree(tmp_path / "one", gpus=1, numa=-1)
    rep = affinity.bind_to_gpu(0, one, environ={}, setaffinity=calls.append, getaffinity=lambda: {0, 1})
    assert not rep["bound"] and rep["numa_node"] == -1 and not calls
    # a GPU whose render node this cgroup cannot open is skipped exactly as the runtime skips it: device 3 is the fifth GPU
    hid = _tree(tmp_path / "hid", hidden=(3,))
    assert affinity.cpus_of_gpu(3, hid, environ={})[1]["render_minor"] == 132


def test_this_process_can_bind_itself_and_restore():
    """the real call on this machine's own sysfs: whatever it decides, it must not raise and must leave a non-empty mask"""
    before = os.sched_getaffinity(0)
    try:
        rep = affinity.bind_to_gpu(0)
        assert isinstance(rep["bound"], bool)
        assert os.sched_getaffinity(0) and os.sched_getaffinity(0) <= before
    finally:
        os.sched_setaffinity(0, before)


def test_c_side_reads_the_same_tree(tmp_path, monkeypatch):
    """gpq_device_local_cpus (gpqhe_amd/csrc/affinity.hip) over the fake tree: no GPU and no HIP call involved"""
    from gpqhe_amd import _native
    lib = _native.load()
    root = _tree(tmp_path)
    buf = C.create_string_buffer(256)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    for dev in range(8):
        count = lib.gpq_device_local_cpus(dev, root.encode(), buf, len(buf))
        lo = (dev // 4) * 48
        assert count == 96 and buf.value.decode() == "%d-%d,%d-%d" % (lo, lo + 47, 96 + lo, 96 + lo + 47)
        assert affinity.parse_cpulist(buf.value.decode()) == affinity.cpus_of_gpu(dev, root, environ={})[0]
    assert lib.gpq_device_local_cpus(8, root.encode(), buf, len(buf)) == 0 and buf.value == b""
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5,1")
    assert lib.gpq_device_local_cpus(0, root.encode(), buf, len(buf)) == 96 and buf.value.decode().startswith("48-95")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4,5,6,7")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3,0")
    assert lib.gpq_device_local_cpus(1, root.encode(), buf, len(buf)) == 96 and buf.value.decode().startswith("48-95")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-abcdef")
    assert lib.gpq_device_local_cpus(0, root.encode(), buf, len(buf)) == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES"); monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    one = _tree(tmp_path / "one", gpus=1, numa=-1)
    assert lib.gpq_device_local_cpus(0, one.encode(), buf, len(buf)) == 0
    assert lib.gpq_device_local_cpus(0, str(tmp_path / "nothing").encode(), buf, len(buf)) == 0
    # the binding call on the real sysfs of this machine: any answer but a crash, and the mask stays usable
    before = os.sched_getaffinity(0)
    try:
        n = lib.gpq_bind_thread_to_device(0)
        assert n >= 0 and os.sched_getaffinity(0)
    finally:
        os.sched_setaffinity(0, before)
