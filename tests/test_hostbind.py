"""gr-ieee80211_amd/hostbind.py: a rank finds its GPU's NUMA node in sysfs and binds to it; one HIP runtime per process."""
import os

from c8p_testlib import ROOT  # noqa: F401  (puts the package on the path)
from gr_ieee80211_amd import hostbind


def fake_sysfs(tmp_path, addr, node, cpulists):
    d = tmp_path / "bus" / "pci" / "devices" / addr
    d.mkdir(parents=True)
    (d / "numa_node").write_text("%d\n" % node)
    for n, txt in cpulists.items():
        nd = tmp_path / "devices" / "system" / "node" / ("node%d" % n)
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(txt + "\n")
    return str(tmp_path)


def test_cpulist_round_trip():
    assert hostbind.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert hostbind._compact([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    assert hostbind.parse_cpulist("") == [] and hostbind.parse_cpulist(None) == []


def test_node_lookup_and_dry_binding(tmp_path, monkeypatch):
    sysfs = fake_sysfs(tmp_path, "0000:c1:00.0", 1, {0: "0-3", 1: "4-7"})
    assert hostbind.numa_node_of_pci("0000:c1:00.0", sysfs) == 1
    assert hostbind.node_cpus(1, sysfs) == [4, 5, 6, 7]
    monkeypatch.setattr(hostbind, "gpu_pci_address", lambda local, torch=None: "0000:c1:00.0")
    info = hostbind.bind_rank(0, sysfs=sysfs, apply=False)
    allowed = sorted(os.sched_getaffinity(0))
    assert info["numa_node"] == 1 and info["pci"] == "0000:c1:00.0"
    assert info["cpus"] == hostbind._compact([c for c in (4, 5, 6, 7) if c in allowed])
    assert info["affinity"].startswith("not applied") or info["affinity"].startswith("kept")


def test_single_node_platform_is_left_alone(tmp_path, monkeypatch):
    sysfs = fake_sysfs(tmp_path, "0000:05:00.0", -1, {0: "0-7"})
    monkeypatch.setattr(hostbind, "gpu_pci_address", lambda local, torch=None: "0000:05:00.0")
    before = sorted(os.sched_getaffinity(0))
    info = hostbind.bind_rank(0, sysfs=sysfs)
    assert info["numa_node"] == -1 and info["affinity"].startswith("kept")
    assert sorted(os.sched_getaffinity(0)) == before


def test_binding_applies_and_is_inherited_by_threads(tmp_path, monkeypatch):
    import threading
    allowed = sorted(os.sched_getaffinity(0))
    half = allowed[:max(1, len(allowed) // 2)]
    sysfs = fake_sysfs(tmp_path, "0000:c1:00.0", 0, {0: hostbind._compact(half)})
    monkeypatch.setattr(hostbind, "gpu_pci_address", lambda local, torch=None: "0000:c1:00.0")
    try:
        info = hostbind.bind_rank(0, sysfs=sysfs)
        assert info["affinity"] == "set" and sorted(os.sched_getaffinity(0)) == half
        got = []
        t = threading.Thread(target=lambda: got.append(sorted(os.sched_getaffinity(0))))
        t.start(); t.join()
        assert got == [half]
        assert info["mempolicy"].startswith("preferred node 0") or info["mempolicy"].startswith("kept")
    finally:
        os.sched_setaffinity(0, allowed)


def test_no_gpu_no_binding(monkeypatch):
    monkeypatch.setattr(hostbind, "gpu_pci_address", lambda local, torch=None: None)
    info = hostbind.bind_rank(0)
    assert info["numa_node"] is None and info["affinity"].startswith("kept")


def test_at_most_one_hip_runtime_here():
    assert len(hostbind.assert_single_hip_runtime()) <= 1


def fake_topology(tmp_path, l3_groups, smt):
    """cpuN/cache/index3/shared_cpu_list and cpuN/topology/thread_siblings_list for groups of cores; smt: {cpu: sibling}"""
    for grp in l3_groups:
        txt = ",".join(str(c) for c in grp)
        for c in grp:
            d = tmp_path / "devices" / "system" / "cpu" / ("cpu%d" % c)
            (d / "cache" / "index3").mkdir(parents=True)
            (d / "cache" / "index3" / "shared_cpu_list").write_text(txt + "\n")
            (d / "topology").mkdir(parents=True)
            sib = sorted({c, smt.get(c, c)})
            (d / "topology" / "thread_siblings_list").write_text(",".join(str(x) for x in sib) + "\n")
    return str(tmp_path)


def test_cpu_pair_shares_a_cache_and_avoids_busy_cores(tmp_path, monkeypatch):
    """two cores of one last-level cache, never two threads of one core; of all such pairs the least busy one (a shared host)"""
    sysfs = fake_topology(tmp_path / "sys", [[0, 1, 2, 3, 8, 9, 10, 11], [4, 5, 6, 7, 12, 13, 14, 15]], {c: c + 8 for c in range(8)} | {c + 8: c for c in range(8)})
    allowed = list(range(16))
    assert hostbind.cpu_pair_sharing_l3(sysfs, sample_s=0, allowed=allowed) == (0, 1)
    assert hostbind.cpu_pair_sharing_l3(sysfs, sample_s=0, allowed=[0, 8, 4, 5]) == (4, 5)          # 0 and 8 are one core
    assert hostbind.cpu_pair_sharing_l3(sysfs, sample_s=0, allowed=[0, 8]) is None
    # busy fractions between two readings of <procfs>/stat: core 0 (its sibling 8 is busy), cores 1 and 2 busy; 3 idle; the second cache idle
    snaps = [{c: (0, 1000) for c in range(16)}, {c: (0, 2000) for c in range(16)}]
    for c, b in ((8, 900), (1, 800), (2, 700)):
        snaps[1][c] = (b, 2000)
    it = iter(snaps)
    monkeypatch.setattr(hostbind, "_cpu_jiffies", lambda procfs: next(it))
    monkeypatch.setattr(hostbind.time, "sleep", lambda s: None)
    assert hostbind.cpu_pair_sharing_l3(sysfs, sample_s=0.01, allowed=allowed) == (4, 5)
    snaps[1][4] = snaps[1][5] = snaps[1][6] = snaps[1][7] = (1000, 2000)                            # the second cache all busy: the best of the first
    it = iter(snaps)
    assert hostbind.cpu_pair_sharing_l3(sysfs, sample_s=0.01, allowed=allowed) == (2, 3)           # (cores 3 and 2: 0 + 0.7; the second cache: 1.0 + 1.0)


def test_cpu_jiffies_reads_proc_stat(tmp_path):
    (tmp_path / "stat").write_text("cpu  10 0 10 100 0 0 0 0 0 0\ncpu0 5 0 5 80 10 0 0 0 0 0\ncpu1 50 0 20 20 0 0 10 0 0 0\nintr 1 2 3\n")
    j = hostbind._cpu_jiffies(str(tmp_path))
    assert j == {0: (10, 100), 1: (80, 100)}
    assert hostbind._cpu_jiffies(str(tmp_path / "nothing")) == {}


def test_eight_gpus_on_two_sockets_bind_four_ranks_to_each(tmp_path, monkeypatch):
    """the node the driver's scaling run uses: 8 GPUs, 4 per socket (an MI355X platform's usual PCIe addresses), 2 NUMA nodes of 8 CPUs in
    this fixture (the test's own cpuset is stood in for: every CPU allowed).  Rank r finds GPU r's node and the node's CPUs; ranks 0-3
    land on node 0, 4-7 on node 1; a device whose function 0 is not listed (SR-IOV: only .1) is still found."""
    addrs = ["0000:05:00.0", "0000:15:00.0", "0000:65:00.0", "0000:75:00.0", "0000:85:00.0", "0000:95:00.0", "0000:e5:00.0", "0000:f5:00.1"]
    for i, a in enumerate(addrs):
        d = tmp_path / "bus" / "pci" / "devices" / a
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % (i // 4))
    for n, txt in {0: "0-7", 1: "8-15"}.items():
        nd = tmp_path / "devices" / "system" / "node" / ("node%d" % n)
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(txt + "\n")
    monkeypatch.setattr(hostbind, "gpu_pci_address", lambda local, torch=None: addrs[local][:-1] + "0")      # (torch reports no function)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)))
    got = [hostbind.bind_rank(r, sysfs=str(tmp_path), apply=False) for r in range(8)]
    assert [g["numa_node"] for g in got] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert [g["cpus"] for g in got] == ["0-7"] * 4 + ["8-15"] * 4 and all(g["cpu_count"] == 8 for g in got)
    assert all(g["affinity"].startswith("not applied") for g in got)
