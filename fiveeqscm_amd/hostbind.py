"""Bind a rank's host thread to the CPUs next to its GPU — BEFORE anything touches the GPU.

A rank of the sharded run is one host thread that enqueues two kernel launches per 35 us; on a two-socket
node a thread that wanders to the other socket pays a cross-socket hop on every doorbell write and every queue-full wait.  The
binding is read from sysfs alone, so that it can be applied before the HIP runtime exists (its helper threads then inherit the
mask, and nothing re-executes the process):

    KFD topology  /sys/class/kfd/kfd/topology/nodes/*/properties   -> the GPUs in the order ROCr (and so HIP) enumerates them,
                                                                      each with its PCI domain / location_id
    visible-devices filters (ROCR_VISIBLE_DEVICES, then HIP_ / CUDA_VISIBLE_DEVICES; integer forms)  -> HIP's device indices
    /sys/bus/pci/devices/<bdf>/local_cpulist (or numa_node -> /sys/devices/system/node/nodeK/cpulist) -> the CPUs of that GPU
    intersected with the mask the process already has (a container's share), and — when several local ranks have the same CPU
    list — cut into contiguous, DISJOINT slices of whole cores (SMT siblings stay together) in local-rank order, so that eight
    ranks end up on eight different CPU sets.

`bind_rank` applies the mask with os.sched_setaffinity and returns a report; `verify` compares the PCI address guessed from
sysfs with the one the runtime reports after initialisation and re-binds if the enumeration order was not what sysfs suggested
(UUID-style filters, a runtime that reorders) — the report says which of the two happened.  Nothing here imports torch.
"""
import os

_SYS = "/sys"


def parse_cpulist(text):
    """'0-3,8,10-11' -> sorted list of ints (the kernel's cpulist format); empty or garbage -> []."""
    out = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        try:
            if "-" in part:
                lo, hi = part.split("-", 1)
                out.update(range(int(lo), int(hi) + 1))
            else:
                out.add(int(part))
        except ValueError:
            return []
    return sorted(out)


def format_cpulist(cpus):
    """[0,1,2,3,8] -> '0-3,8'."""
    cpus = sorted(set(int(c) for c in cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(runs)


def _read(path):
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return None


def kfd_gpus(sys_root=_SYS):
    """PCI addresses ('0000:c1:00.0') of the GPU nodes of the KFD topology, in node order — the order ROCr enumerates its GPU
    agents in.  [] when the topology is not readable (no amdgpu driver, a sandbox without /sys/class/kfd)."""
    base = os.path.join(sys_root, "class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    out = []
    for n in nodes:
        text = _read(os.path.join(base, str(n), "properties"))
        if text is None:
            continue
        props = {}
        for line in text.splitlines():
            kv = line.split()
            if len(kv) == 2:
                props[kv[0]] = kv[1]
        try:
            if int(props.get("simd_count", "0")) <= 0:           # a CPU node
                continue
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        except (KeyError, ValueError):
            continue
        out.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7))
    return out


def _filter_indices(items, spec):
    """Apply one *_VISIBLE_DEVICES value of integer indices to a list; None = the value cannot be applied from here (UUID forms)."""
    picked = []
    for tok in spec.split(","):
        tok = tok.strip()
        if not tok:
            continue
        try:
            i = int(tok)
        except ValueError:
            return None
        if not 0 <= i < len(items):
            break                                                # the runtimes stop at the first invalid index
        picked.append(items[i])
    return picked


def visible_gpus(sys_root=_SYS, env=None):
    """(PCI addresses in HIP device-index order, exact) — exact=False when a filter could not be interpreted from sysfs."""
    env = os.environ if env is None else env
    gpus, exact = kfd_gpus(sys_root), True
    stages = [env.get("ROCR_VISIBLE_DEVICES")]
    hip = env.get("HIP_VISIBLE_DEVICES")
    stages.append(hip if hip is not None else env.get("CUDA_VISIBLE_DEVICES"))
    for spec in stages:
        if spec is None:
            continue
        got = _filter_indices(gpus, spec)
        if got is None:
            exact = False
            continue
        gpus = got
    return gpus, exact


def gpu_cpus(bdf, sys_root=_SYS):
    """(CPUs local to the PCI device, its NUMA node or None, which file said so)."""
    dev = os.path.join(sys_root, "bus/pci/devices", bdf)
    node = _read(os.path.join(dev, "numa_node"))
    try:
        node = int(node.strip()) if node is not None else None
    except ValueError:
        node = None
    text = _read(os.path.join(dev, "local_cpulist"))
    cpus = parse_cpulist(text) if text else []
    if cpus:
        return cpus, node, "local_cpulist"
    if node is not None and node >= 0:
        text = _read(os.path.join(sys_root, "devices/system/node", f"node{node}", "cpulist"))
        cpus = parse_cpulist(text) if text else []
        if cpus:
            return cpus, node, "numa_node"
    return [], node, None


def _cores(cpus, sys_root=_SYS):
    """`cpus` grouped into physical cores (thread_siblings_list), ordered by their first CPU; without the topology files every
    CPU is its own core."""
    left, out = set(cpus), []
    for c in sorted(cpus):
        if c not in left:
            continue
        text = _read(os.path.join(sys_root, "devices/system/cpu", f"cpu{c}", "topology/thread_siblings_list"))
        sib = [x for x in (parse_cpulist(text) if text else []) if x in left] or [c]
        if c not in sib:
            sib.append(c)
        left -= set(sib)
        out.append(sorted(sib))
    return out


def plan(local_rank, local_world, allowed, sys_root=_SYS, env=None, bdf=None):
    """The CPU set for `local_rank` of `local_world` ranks on this host, without applying it.  `allowed`: the CPUs the process
    may use now.  `bdf`: the GPU's PCI address when already known (verify()); else device index local_rank % visible GPUs.
    Every rank computes every rank's list from the same files, so the disjoint slices need no communication."""
    gpus, exact = visible_gpus(sys_root, env)
    rep = {"local_rank": int(local_rank), "pci_bus_id_from_sysfs": None, "numa_node": None, "cpus": None, "n_cpus": 0,
           "source": None, "applied": False, "sysfs_order_exact": exact}
    if bdf is None:
        if not gpus:
            rep["reason"] = "no KFD topology readable from sysfs"
            return rep, None
        bdf = gpus[local_rank % len(gpus)]
    rep["pci_bus_id_from_sysfs"] = bdf
    allowed = sorted(set(allowed))
    mine, node, src = gpu_cpus(bdf, sys_root)
    rep["numa_node"], rep["source"] = node, src
    mine = [c for c in mine if c in set(allowed)]
    if not mine:
        rep["reason"] = ("the device has no local_cpulist / numa_node" if src is None else
                         "the GPU's local CPUs are outside the mask this process was given")
        return rep, None
    # ranks whose GPUs have the same CPU list share it in disjoint contiguous slices (local-rank order)
    sharing = []
    for r in range(int(local_world)):
        b = bdf if r == local_rank else (gpus[r % len(gpus)] if gpus else None)
        if b is None:
            continue
        theirs = [c for c in gpu_cpus(b, sys_root)[0] if c in set(allowed)]
        if theirs == mine:
            sharing.append(r)
    if local_rank not in sharing:
        sharing = sorted(sharing + [local_rank])
    k, n = sharing.index(local_rank), len(sharing)
    cores = _cores(mine, sys_root)                               # hardware threads of one core stay with one rank
    if n > 1 and len(cores) >= n:
        lo, hi = k * len(cores) // n, (k + 1) * len(cores) // n
        mine = sorted(c for core in cores[lo:hi] for c in core)
        rep["sliced"] = f"{k + 1} of {n} ranks on this CPU list"
    rep["cpus"], rep["n_cpus"] = format_cpulist(mine), len(mine)
    return rep, mine


def bind_rank(local_rank, local_world=1, sys_root=_SYS, env=None, apply=True):
    """Bind the calling thread (every thread it starts afterwards — the HIP runtime's helpers — inherits the mask) to the CPUs of its GPU.  Call BEFORE the
    first GPU call.  Returns the report bench.py prints as config.devices[].cpu_binding.  FIVEEQ_BIND_CPUS=0 disables it."""
    env = os.environ if env is None else env
    if not hasattr(os, "sched_getaffinity"):
        return {"applied": False, "reason": "no sched_setaffinity on this platform"}
    allowed = sorted(os.sched_getaffinity(0))
    rep, mine = plan(local_rank, local_world, allowed, sys_root, env)
    rep["cpus_before"] = format_cpulist(allowed)
    if env.get("FIVEEQ_BIND_CPUS", "1") == "0":
        rep["reason"] = "FIVEEQ_BIND_CPUS=0"
        return rep
    if mine and apply:
        try:
            os.sched_setaffinity(0, mine)
            rep["applied"] = True
        except OSError as exc:
            rep["reason"] = f"sched_setaffinity: {exc}"
    return rep


def verify(rep, actual_bdf, local_rank, local_world=1, sys_root=_SYS, env=None):
    """After GPU initialisation: `actual_bdf` ('0000:c1:00.0' as the runtime reports the rank's device).  If sysfs had guessed
    another device, re-plan for the real one from the ORIGINAL mask and re-bind the calling thread.  Returns the report
    with `verified` / `rebound_after_init`."""
    rep = dict(rep)
    guess = rep.get("pci_bus_id_from_sysfs")
    rep["pci_bus_id_runtime"] = actual_bdf
    same = guess is not None and guess.lower().split(".")[0] == actual_bdf.lower().split(".")[0]
    rep["verified"] = bool(same)
    if same or (os.environ if env is None else env).get("FIVEEQ_BIND_CPUS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return rep
    allowed = parse_cpulist(rep.get("cpus_before", "")) or sorted(os.sched_getaffinity(0))
    full = actual_bdf if "." in actual_bdf else actual_bdf + ".0"
    rep2, mine = plan(local_rank, local_world, allowed, sys_root, env, bdf=full)
    rep.update({k: rep2[k] for k in ("numa_node", "cpus", "n_cpus", "source") if k in rep2})
    rep["rebound_after_init"] = False
    if mine:
        try:
            os.sched_setaffinity(0, mine)
            rep["applied"], rep["rebound_after_init"] = True, True
        except OSError as exc:
            rep["reason"] = f"sched_setaffinity: {exc}"
    return rep
