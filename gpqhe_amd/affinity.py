"""Host-side placement of a rank: bind the process to the CPUs of its GPU's NUMA node, in-process, BEFORE the first HIP call.

One rank per GPU on a two-socket host: the rank's launch thread, the HIP runtime's helper threads (they inherit the mask at
runtime initialisation) and the page-locked staging memory it first-touches (`with_host_scatter`: eight PCIe uploads in parallel) should sit on
the socket the GPU hangs off.  No `taskset` / `numactl` hop -- a launcher that re-execs is exactly what this pool forbids once the GPU is
initialised -- and no HIP call: everything comes from sysfs.

  HIP device i  ->  the i-th GPU node of /sys/class/kfd/kfd/topology/nodes (simd_count > 0, render node openable by this process), after
                    ROCR_VISIBLE_DEVICES and then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (integer lists) have been applied
                ->  drm_render_minor  ->  /sys/class/drm/renderD<minor>/device/{numa_node, local_cpulist}
                ->  os.sched_setaffinity(0, local CPUs that the current mask allows)

Anything missing or odd (no KFD topology, a UUID in a visibility list, numa_node -1, an empty intersection) means "no binding", reported, never
an error: placement is an optimisation.  The C hosts get the same through `gpq_bind_thread_to_device` (include/gpqhe_hip.h).
"""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11}"""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.update(range(int(lo), int(hi) + 1))
        else:
            cpus.add(int(part))
    return cpus


def _int_list(text):
    out = []
    for tok in text.split(","):
        tok = tok.strip()
        if tok == "":
            continue
        out.append(int(tok))            # ValueError on a UUID form: the caller gives up
    return out


def gpu_nodes(root="/", can_open=None):
    """[(kfd node id, drm_render_minor)] of the GPU nodes this process could open, in the order the HIP runtime enumerates them."""
    base = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    if can_open is None:
        def can_open(minor):
            path = os.path.join(root, "dev/dri/renderD%d" % minor)
            return os.access(path, os.R_OK | os.W_OK)
    nodes = []
    for name in sorted(os.listdir(base), key=int):
        props = {}
        try:
            with open(os.path.join(base, name, "properties")) as f:
                for line in f:
                    kv = line.split()
                    if len(kv) == 2:
                        props[kv[0]] = kv[1]
        except OSError:
            continue                     # a node this cgroup may not read: the runtime cannot use it either
        if int(props.get("simd_count", "0")) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1"))
        if minor < 0 or not can_open(minor):
            continue
        nodes.append((int(name), minor))
    return nodes


def visible_nodes(nodes, environ):
    """apply ROCR_VISIBLE_DEVICES (the ROCr layer), then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (the HIP layer)"""
    for var in ("ROCR_VISIBLE_DEVICES", ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        names = var if isinstance(var, tuple) else (var,)
        text = next((environ[n] for n in names if environ.get(n, "") != ""), None)
        if text is None:
            continue
        picked = []
        for i in _int_list(text):
            if i < 0 or i >= len(nodes):
                break                    # the runtimes stop at the first invalid index
            picked.append(nodes[i])
        nodes = picked
    return nodes


def cpus_of_gpu(index, root="/", environ=None, can_open=None):
    """(set of CPUs local to HIP device `index`, report dict).  The set is empty when nothing can be said."""
    environ = os.environ if environ is None else environ
    rep = {"device": index}
    try:
        nodes = visible_nodes(gpu_nodes(root, can_open), environ)
        if index < 0 or index >= len(nodes):
            rep["why_not"] = "device %d not among the %d GPU node(s) sysfs shows" % (index, len(nodes))
            return set(), rep
        node, minor = nodes[index]
        dev = os.path.join(root, "sys/class/drm/renderD%d/device" % minor)
        rep.update(kfd_node=node, render_minor=minor)
        with open(os.path.join(dev, "numa_node")) as f:
            numa = int(f.read().strip())
        rep["numa_node"] = numa
        if numa < 0:
            rep["why_not"] = "numa_node is -1 (one memory domain, or the platform does not say)"
            return set(), rep
        with open(os.path.join(dev, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
        rep["local_cpus"] = len(cpus)
        return cpus, rep
    except (OSError, ValueError) as exc:
        rep["why_not"] = "%s: %s" % (type(exc).__name__, exc)
        return set(), rep


def bind_to_gpu(index, root="/", environ=None, can_open=None, setaffinity=None, getaffinity=None):
    """Restrict the calling process to the CPUs of HIP device `index`'s NUMA node (intersected with what it is allowed already).
    Call before the first HIP call of the process.  Returns the report that goes into bench.py's line."""
    setaffinity = setaffinity or (lambda cpus: os.sched_setaffinity(0, cpus))
    getaffinity = getaffinity or (lambda: os.sched_getaffinity(0))
    cpus, rep = cpus_of_gpu(index, root, environ, can_open)
    rep["bound"] = False
    try:
        allowed = set(getaffinity())
    except (AttributeError, OSError) as exc:
        rep.setdefault("why_not", "sched_getaffinity: %s" % exc)
        return rep
    rep["allowed_cpus_before"] = len(allowed)
    if not cpus:
        return rep
    target = cpus & allowed
    if not target:
        rep["why_not"] = "none of the node's %d CPUs is in this process's mask (cgroup / launcher restriction)" % len(cpus)
        return rep
    if target == allowed:
        rep["why_not"] = "already confined to the GPU's node"
        rep["cpus"] = len(target)
        return rep
    try:
        setaffinity(target)
    except OSError as exc:
        rep["why_not"] = "sched_setaffinity: %s" % exc
        return rep
    rep.update(bound=True, cpus=len(target))
    return rep
