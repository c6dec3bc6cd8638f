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
