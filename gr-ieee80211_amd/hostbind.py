"""Where a rank's host side runs: the CPUs and the memory of the NUMA node its GPU hangs off, and ONE HIP runtime in the process.

One process per GPU feeds its device from pinned host memory (the block mirrors' scheduler buffers, the raw chain's ingest rings):
on a two-socket node a buffer pinned on the other socket crosses the socket interconnect on every PCIe read, and "every GPU has its own
link"  stops being true.  bind_rank() pins the process — and every thread it starts later: the engines' stage threads,
the fronts' timers — to the CPUs of the GPU's node and makes that node the preferred one for every allocation that follows
(hipHostMalloc included: pinned pages are allocated where the calling thread's policy says).  Nothing here is specific to the engine;
no arithmetic."""
import ctypes
import os
import time
import re


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(txt):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in (txt or "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def gpu_pci_address(local, torch=None):
    """'0000:c1:00.0' of HIP device `local` (None when it cannot be told: no GPU, or a runtime that does not say)"""
    try:
        if torch is None:
            import torch
        p = torch.cuda.get_device_properties(local)
        dom, bus, dev = getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
        if bus is None or dev is None:
            return None
        return "%04x:%02x:%02x.0" % (dom, bus, dev)
    except Exception:
        return None


def numa_node_of_pci(addr, sysfs="/sys"):
    """the device's NUMA node as the kernel reports it; -1 / None: the platform has (or tells of) only one"""
    if not addr:
        return None
    v = _read(os.path.join(sysfs, "bus", "pci", "devices", addr, "numa_node"))
    if v is None and addr.endswith(".0"):         # (torch tells domain:bus:device, not the function: any function of the device will do)
        import glob
        for d in sorted(glob.glob(os.path.join(sysfs, "bus", "pci", "devices", addr[:-1] + "*"))):
            v = _read(os.path.join(d, "numa_node"))
            if v is not None:
                break
    try:
        return int(v)
    except (TypeError, ValueError):
        return None


def node_cpus(node, sysfs="/sys"):
    return parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")))


_MPOL_PREFERRED = 1


def _set_preferred_node(node):
    """set_mempolicy(MPOL_PREFERRED, {node}): allocations of this thread (and of threads it starts) come from `node` while it has pages"""
    libc = ctypes.CDLL(None, use_errno=True)
    mask = (ctypes.c_ulong * 16)()
    mask[node // (8 * ctypes.sizeof(ctypes.c_ulong))] |= 1 << (node % (8 * ctypes.sizeof(ctypes.c_ulong)))
    # the syscall number is per architecture (238 is migrate_pages on aarch64): known machines only, anything else keeps its policy
    import platform
    nr = {"x86_64": 238, "amd64": 238, "aarch64": 237, "arm64": 237}.get(platform.machine().lower())
    if nr is None:
        return False
    rc = libc.syscall(nr, _MPOL_PREFERRED, ctypes.byref(mask), 16 * 8 * ctypes.sizeof(ctypes.c_ulong))
    return rc == 0


def bind_rank(local, torch=None, sysfs="/sys", apply=True):
    """Bind this process to the NUMA node of GPU `local`.  Returns what was found and done, for the bench line:
    {"gpu", "pci", "numa_node", "cpus", "cpu_count", "affinity": "set" | "kept (...)", "mempolicy": "preferred node N" | "kept (...)"}.
    A platform with one node (numa_node -1, or a container whose sysfs hides it) is left as it is — and says so."""
    info = {"gpu": int(local), "pci": gpu_pci_address(local, torch)}
    node = numa_node_of_pci(info["pci"], sysfs)
    info["numa_node"] = node
    if node is None or node < 0:
        info["affinity"] = info["mempolicy"] = "kept (the platform reports no NUMA node for the device)"
        return info
    cpus = node_cpus(node, sysfs)
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = []
    use = [c for c in cpus if c in allowed] if allowed else cpus
    info["cpus"] = _compact(use)
    info["cpu_count"] = len(use)
    if not use:
        info["affinity"] = info["mempolicy"] = "kept (none of the node's CPUs is in this process's cpuset)"
        return info
    if not apply:
        info["affinity"] = info["mempolicy"] = "not applied (dry run)"
        return info
    try:
        os.sched_setaffinity(0, use)
        info["affinity"] = "set"
    except OSError as e:
        info["affinity"] = "kept (sched_setaffinity: %s)" % e
    try:
        info["mempolicy"] = "preferred node %d" % node if _set_preferred_node(node) else "kept (set_mempolicy refused: errno %d)" % ctypes.get_errno()
    except Exception as e:
        info["mempolicy"] = "kept (%s)" % e
    return info


def _compact(cpus):
    """[0, 1, 2, 3, 8] -> '0-3,8'"""
    out, i = [], 0
    cpus = sorted(cpus)
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append("%d" % cpus[i] if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def _cpu_jiffies(procfs):
    """per logical CPU: (busy, total) jiffies since boot, from <procfs>/stat; {} when it cannot be read"""
    out = {}
    try:
        with open(os.path.join(procfs, "stat")) as f:
            for line in f:
                m = re.match(r"cpu(\d+)\s+(.*)", line)
                if not m:
                    continue
                v = [int(x) for x in m.group(2).split()]
                idle = v[3] + (v[4] if len(v) > 4 else 0)
                out[int(m.group(1))] = (sum(v) - idle, sum(v))
    except (OSError, ValueError):
        return {}
    return out


def cpu_pair_sharing_l3(sysfs="/sys", procfs="/proc", sample_s=0.05, allowed=None):
    """Two logical CPUs of this process's cpuset on DIFFERENT cores that share a last-level cache: where two scheduler threads that hand
    items and tags to each other (demod -> decode) belong — GNU Radio: block::set_processor_affinity / the blocks' "Core Affinity" in
    GRC.  On a two-socket, multi-CCD host threads left to the OS end up on different CCDs or sockets and every hand-over is a
    cross-fabric cache-line transfer (measured: the block chain at 250-350 k packets/s floating, 440-450 k on two cores of one CCD).
    The hosts of this pool are shared (load averages of 30-50 on 256 CPUs): of all such pairs the one whose cores were the least busy
    during `sample_s` seconds is taken (per core: the busier of its hardware threads, <procfs>/stat); sample_s = 0 or an unreadable
    stat file: the first pair.  None when the topology cannot be read."""
    if allowed is None:
        try:
            allowed = sorted(os.sched_getaffinity(0))
        except AttributeError:
            return None
    allowed = sorted(allowed)
    aset = set(allowed)
    groups, seen = [], set()                     # per last-level cache: its cores, each as the tuple of its allowed hardware threads
    for c0 in allowed:
        if c0 in seen:
            continue
        base = os.path.join(sysfs, "devices", "system", "cpu", "cpu%d" % c0)
        l3 = [c for c in parse_cpulist(_read(os.path.join(base, "cache", "index3", "shared_cpu_list"))) if c in aset]
        if not l3:
            continue
        cores, in_core = [], set()
        for c in l3:
            if c in in_core:
                continue
            every = parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "cpu", "cpu%d" % c, "topology", "thread_siblings_list"))) or [c]
            sib = tuple(x for x in every if x in aset) or (c,)
            in_core.update(sib)
            cores.append((sib, tuple(every)))              # (ours to run on, all of the core's: a neighbour's thread loads the core too)
        seen.update(l3)
        if len(cores) >= 2:
            groups.append(cores)
    if not groups:
        return None
    busy = {}
    if sample_s > 0:
        a = _cpu_jiffies(procfs)
        if a:
            time.sleep(sample_s)
            b = _cpu_jiffies(procfs)
            for c, (b1, t1) in b.items():
                if c in a and t1 > a[c][1]:
                    busy[c] = (b1 - a[c][0]) / (t1 - a[c][1])
    best, best_load = None, None
    for cores in groups:
        ranked = sorted(cores, key=lambda c: (max(busy.get(x, 0.0) for x in c[1]), c[0][0]))
        load = sum(max(busy.get(x, 0.0) for x in c[1]) for c in ranked[:2])
        if best is None or load < best_load - 1e-9:
            best, best_load = tuple(sorted((ranked[0][0][0], ranked[1][0][0]))), load
    return best


def hip_runtimes_mapped():
    """the distinct libamdhip64 files mapped into this process"""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                m = re.search(r"(/\S*libamdhip64\.so[^\s]*)", line)
                if m:
                    paths.add(os.path.realpath(m.group(1)))
    except OSError:
        pass
    return sorted(paths)


def assert_single_hip_runtime():
    """PyTorch-ROCm wheels carry their own libamdhip64.so and libc8p_rx.so is linked against the system's: same soname, so the one
    that is loaded first serves both (engine.prefer_torch_hip_runtime loads torch's).  TWO of them in one process means two sets of
    streams, events and device state that do not see each other — an event of the engine's waited for on a torch / RCCL stream is then
    undefined behaviour, and RCCL's IPC handles come from the runtime torch was built for.  Checked once at start-up by bench.py and
    shard.py; raises with both paths."""
    got = hip_runtimes_mapped()
    if len(got) > 1:
        raise RuntimeError("two HIP runtimes are mapped into this process: %s — import gr_ieee80211_amd.engine (or call "
                           "engine.prefer_torch_hip_runtime()) before anything else loads libamdhip64" % ", ".join(got))
    return got
