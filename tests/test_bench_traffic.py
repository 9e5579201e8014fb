"""bench.py reports counter traffic (roofline.traffic) only from a counter file taken from the very sources the library
was built from, for the headline launch and for each secondary configuration's step kernel (tools/pmc_traffic.py)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def _write(tmp_path, digest):
    os.makedirs(tmp_path / "profiles", exist_ok=True)
    rec = {"kernel": "stretch_full_step_mfma_kernel<double, mcmcpp::DenseGaussianFn<double>, 2, 16, false>",
           "hbm_bytes_per_launch": 13.0e6, "source_sha256": digest,
           "secondary": {"C3": {"kernel": "stretch_half_step_kernel<double, mcmcpp::RosenbrockFn<double>, 2, 16, true, false>",
                                "hbm_bytes_per_launch": 24.0e6}}}
    json.dump(rec, open(tmp_path / "profiles" / "r99_pmc_traffic.json", "w"))


def test_traffic_only_from_matching_sources(tmp_path, monkeypatch):
    digest = bench.source_digest()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_digest", lambda: digest)
    _write(tmp_path, digest)
    assert bench.counter_traffic("stretch_full_step_mfma_kernel<double, DenseGaussianFn, EPL=2, LPW=16>")[0] == 13.0e6
    assert bench.counter_traffic("stretch_half_step_kernel<double, RosenbrockFn, EPL=2, LPW=16>", "C3")[0] == 24.0e6
    # another kernel under the same key, a key the passes did not cover, a kernel the headline record is not about: null
    assert bench.counter_traffic("stretch_half_step_mfma_kernel<double>", "C3") == (None, None)
    assert bench.counter_traffic("stretch_half_step_kernel<double, IsoGaussianFn>", "C5_one_gpu") == (None, None)
    assert bench.counter_traffic("stretch_half_step_kernel<double, RosenbrockFn>") == (None, None)
    _write(tmp_path, "0" * 64)  # taken from other sources: stale, not reported
    assert bench.counter_traffic("stretch_full_step_mfma_kernel<double>") == (None, None)
    assert bench.counter_traffic("stretch_half_step_kernel<double>", "C3") == (None, None)


def test_secondary_kernel_table_of_the_counter_tool():
    sys.path.insert(0, os.path.join(bench.ROOT, "tools"))
    import pmc_traffic
    src = open(os.path.join(bench.ROOT, "bench.py")).read()
    for key in pmc_traffic.SECONDARY:  # every key the tool records is one bench.py asks for
        assert 'traffic_key="%s"' % key in src


def test_numa_binding_helper(tmp_path, monkeypatch):
    """bench.py's placement helper on a made-up sysfs tree: binds to the GPU's node when enough permitted cores live
    there, leaves the affinity alone otherwise -- and never raises."""
    class Props:
        pci_domain_id, pci_bus_id, pci_device_id = 0, 0x65, 0

    class Cuda:
        @staticmethod
        def get_device_properties(_):
            return Props

    class Torch:
        cuda = Cuda

    dev = tmp_path / "bus/pci/devices/0000:65:00.0"
    os.makedirs(dev)
    os.makedirs(tmp_path / "devices/system/node/node1")
    (dev / "numa_node").write_text("1\n")
    (tmp_path / "devices/system/node/node1/cpulist").write_text("0-3,8-15,40\n")
    assert bench._parse_cpulist("0-3,8-15,40\n") == set(range(4)) | set(range(8, 16)) | {40}
    bound = []
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: set(range(64)))
    monkeypatch.setattr(os, "sched_setaffinity", lambda _pid, cpus: bound.append(set(cpus)))
    note = bench.bind_to_gpu_numa_node(Torch, 0, sysfs=str(tmp_path))
    assert bound == [set(range(4)) | set(range(8, 16)) | {40}] and note.startswith("NUMA node 1")
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: {0, 1, 2, 50})  # a container's share: too few cores on that node
    assert bench.bind_to_gpu_numa_node(Torch, 0, sysfs=str(tmp_path)).startswith("unchanged") and len(bound) == 1
    (dev / "numa_node").write_text("-1\n")
    assert "no NUMA node" in bench.bind_to_gpu_numa_node(Torch, 0, sysfs=str(tmp_path))
    assert bench.bind_to_gpu_numa_node(Torch, 0, sysfs=str(tmp_path / "nowhere")).startswith("unchanged")
