"""fiveeqscm_amd.hostbind: the CPU set of a rank, read from sysfs before the GPU is touched — on a fake sysfs tree (this
container has no GPU): an 8-GPU two-socket node, a one-GPU container share, filters, missing files."""
import os

import pytest

from fiveeqscm_amd import hostbind as hb


def _node(root, n, simd, loc=0, dom=0):
    d = root / "class/kfd/kfd/topology/nodes" / str(n)
    d.mkdir(parents=True)
    (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nlocation_id {loc}\ndomain {dom}\n")


def _pci(root, bdf, cpulist=None, numa=None):
    d = root / "bus/pci/devices" / bdf
    d.mkdir(parents=True)
    if cpulist is not None:
        (d / "local_cpulist").write_text(cpulist + "\n")
    if numa is not None:
        (d / "numa_node").write_text(f"{numa}\n")


@pytest.fixture
def node8(tmp_path):
    """Two CPU nodes, eight GPUs: four per socket; socket 0 = CPUs 0-63,128-191, socket 1 = 64-127,192-255."""
    _node(tmp_path, 0, 0), _node(tmp_path, 1, 0)
    buses = [0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xE5, 0xF5]
    for i, bus in enumerate(buses):
        _node(tmp_path, 2 + i, 1024, loc=bus << 8)
        _pci(tmp_path, "0000:%02x:00.0" % bus, "0-63,128-191" if i < 4 else "64-127,192-255", numa=0 if i < 4 else 1)
    return tmp_path, ["0000:%02x:00.0" % b for b in buses]


def test_cpulist_round_trip():
    assert hb.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and hb.parse_cpulist("") == [] and hb.parse_cpulist("x") == []
    assert hb.format_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11" and hb.format_cpulist([]) == ""
    assert hb.parse_cpulist(hb.format_cpulist(range(64, 128))) == list(range(64, 128))


def test_eight_ranks_get_eight_disjoint_cpu_sets_on_their_own_sockets(node8):
    root, bdfs = node8
    assert hb.kfd_gpus(str(root)) == bdfs
    allowed = list(range(256))
    sets = []
    for r in range(8):
        rep, mine = hb.plan(r, 8, allowed, str(root), env={})
        assert rep["pci_bus_id_from_sysfs"] == bdfs[r] and rep["numa_node"] == (0 if r < 4 else 1) and rep["source"] == "local_cpulist"
        assert rep["n_cpus"] == 32 == len(mine) and rep["sliced"] == f"{r % 4 + 1} of 4 ranks on this CPU list"
        socket0 = set(range(0, 64)) | set(range(128, 192))
        assert set(mine) <= (socket0 if r < 4 else set(range(256)) - socket0)
        sets.append(set(mine))
    assert all(not (sets[i] & sets[j]) for i in range(8) for j in range(i)) and set().union(*sets) == set(range(256))
    # with the CPU topology readable, SMT siblings (cpu c and c + 128 here) go to the same rank
    for c in range(256):
        d = root / "devices/system/cpu" / f"cpu{c}" / "topology"
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text(f"{c % 128},{c % 128 + 128}\n")
    smt = [set(hb.plan(r, 8, allowed, str(root), env={})[1]) for r in range(8)]
    assert all(len(m) == 32 and {c % 128 for c in m} == {c % 128 for c in m if c < 128} and len({c % 128 for c in m}) == 16 for m in smt)
    assert all(not (smt[i] & smt[j]) for i in range(8) for j in range(i)) and smt[0] == set(range(0, 16)) | set(range(128, 144))
    # one rank alone keeps its GPU's whole list; a mask that holds none of the GPU's CPUs is left alone
    rep, mine = hb.plan(5, 1, allowed, str(root), env={})
    assert rep["n_cpus"] == 128 and "sliced" not in rep
    rep, mine = hb.plan(5, 8, list(range(0, 16)), str(root), env={})
    assert mine is None and "outside the mask" in rep["reason"]


def test_visible_device_filters_and_fallbacks(node8, tmp_path):
    root, bdfs = node8
    assert hb.visible_gpus(str(root), {"HIP_VISIBLE_DEVICES": "3,1"}) == ([bdfs[3], bdfs[1]], True)
    assert hb.visible_gpus(str(root), {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "1"}) == ([bdfs[5]], True)
    assert hb.visible_gpus(str(root), {"CUDA_VISIBLE_DEVICES": "2"}) == ([bdfs[2]], True)
    assert hb.visible_gpus(str(root), {"HIP_VISIBLE_DEVICES": "0", "CUDA_VISIBLE_DEVICES": "2"}) == ([bdfs[0]], True)     # HIP_ wins
    assert hb.visible_gpus(str(root), {"ROCR_VISIBLE_DEVICES": "GPU-abcdef"})[1] is False
    rep, mine = hb.plan(0, 1, list(range(256)), str(root), env={"HIP_VISIBLE_DEVICES": "6"})
    assert rep["pci_bus_id_from_sysfs"] == bdfs[6] and rep["numa_node"] == 1
    # more ranks than GPUs (the one-card rehearsals): device index = local_rank % visible, the sharers split the list
    one = {"HIP_VISIBLE_DEVICES": "0"}
    got = [hb.plan(r, 4, list(range(256)), str(root), env=one) for r in range(4)]
    assert [g[0]["n_cpus"] for g in got] == [32] * 4 and len(set(tuple(g[1]) for g in got)) == 4
    # numa_node only; nothing at all; no topology
    other = tmp_path / "other"
    _node(other, 0, 0), _node(other, 1, 256, loc=0x0300)
    _pci(other, "0000:03:00.0", numa=1)
    (other / "devices/system/node/node1").mkdir(parents=True)
    (other / "devices/system/node/node1/cpulist").write_text("8-15\n")
    rep, mine = hb.plan(0, 1, list(range(16)), str(other), env={})
    assert mine == list(range(8, 16)) and rep["source"] == "numa_node" and rep["cpus"] == "8-15"
    os.remove(other / "bus/pci/devices/0000:03:00.0/numa_node")
    assert hb.plan(0, 1, list(range(16)), str(other), env={})[1] is None
    assert hb.plan(0, 1, list(range(16)), str(tmp_path / "nothing"), env={})[0]["reason"].startswith("no KFD topology")


def test_bind_applies_the_mask_and_verify_rebinds_on_a_wrong_guess(node8):
    root, bdfs = node8
    before = sorted(os.sched_getaffinity(0))
    try:
        fake = root / "mine"
        _node(fake, 0, 0), _node(fake, 1, 1024, loc=0x0500), _node(fake, 2, 1024, loc=0x1500)
        half = before[:max(1, len(before) // 2)]
        _pci(fake, "0000:05:00.0", hb.format_cpulist(half), numa=0)
        _pci(fake, "0000:15:00.0", hb.format_cpulist(before[-1:]), numa=0)
        rep = hb.bind_rank(0, 1, str(fake), env={})
        assert rep["applied"] and sorted(os.sched_getaffinity(0)) == half and rep["cpus"] == hb.format_cpulist(half)
        assert rep["cpus_before"] == hb.format_cpulist(before)
        ok = hb.verify(rep, "0000:05:00", 0, 1, str(fake), env={})                     # torch reports domain:bus:device
        assert ok["verified"] and "rebound_after_init" not in ok
        moved = hb.verify(rep, "0000:15:00", 0, 1, str(fake), env={})                  # the runtime's device 0 is the OTHER card
        assert not moved["verified"] and moved["rebound_after_init"] and sorted(os.sched_getaffinity(0)) == before[-1:]
        assert moved["cpus"] == str(before[-1])
        os.sched_setaffinity(0, before)
        off = hb.bind_rank(0, 1, str(fake), env={"FIVEEQ_BIND_CPUS": "0"})
        assert not off["applied"] and sorted(os.sched_getaffinity(0)) == before and off["reason"] == "FIVEEQ_BIND_CPUS=0"
    finally:
        os.sched_setaffinity(0, before)
