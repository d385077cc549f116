"""CPU tests: the C-ABI library loads and exports every symbol include/vgen_hip.h declares (no compute
without a GPU), fails loudly without a device, and the N>1 path (batch striping + host aggregation,
SURVEY.md §8(e)) is exercised with two gloo ranks, each rank's batches computed by the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vgen_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(vgen_[a-z0-9_]+)\s*\(", hdr))
    names -= {"vgen_progress_cb"}
    assert len(names) >= 20
    lib = ctypes.CDLL(os.path.join(ROOT, "vgen_amd", "libvgen_hip.so"))
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.vgen_abi_version() == 4
    # ... and INTEGRATION.md shows the reference-side binding of every one of them
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    undocumented = [n for n in sorted(names) if n not in doc]
    assert not undocumented, undocumented


def test_fault_injection_is_not_part_of_the_shipped_library():
    """vgen_debug_fail_after and the VGEN_DEBUG_GTAB_FAIL switch exist only in the test build (tests/native/
    libvgen_hip_hooks.so, the same sources with -DVGEN_TEST_HOOKS): the shipped library exports no debug symbol, reads no
    debug variable, and neither the header nor the reference-side binding in INTEGRATION.md declares one."""
    from conftest import HOOKS_SO
    lib = ctypes.CDLL(os.path.join(ROOT, "vgen_amd", "libvgen_hip.so"))
    assert not hasattr(lib, "vgen_debug_fail_after")
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "vgen_amd", "libvgen_hip.so")], capture_output=True, text=True).stdout
    assert "debug" not in syms.lower()
    blob = open(os.path.join(ROOT, "vgen_amd", "libvgen_hip.so"), "rb").read()
    assert b"VGEN_DEBUG" not in blob and b"injected device failure" not in blob
    assert "vgen_debug" not in open(os.path.join(ROOT, "include", "vgen_hip.h")).read()
    assert "vgen_debug" not in open(os.path.join(ROOT, "INTEGRATION.md")).read()
    # the environment variables the shipped library does read are the documented ones
    env = set(re.findall(rb"VGEN_[A-Z0-9_]+", blob))
    read = {b"VGEN_SEQ_S", b"VGEN_GTAB_BITS", b"VGEN_TRACE_CREATE", b"VGEN_LONE_VARIANT",                      # getenv'ed, once, at vgen_create (INTEGRATION.md lists them)
            b"VGEN_LONE_MAX_OTHERS", b"VGEN_SPLIT", b"VGEN_HASH_KPL"}                                            # ... the A/B switches of round 5 among them
    for name in read:
        assert name.decode() in open(os.path.join(ROOT, "INTEGRATION.md")).read(), name
    named_in_messages = {b"VGEN_FLAG_ENDO", b"VGEN_FLAG_TIMING", b"VGEN_SCAN_RANDOM_KEYS"}    # ABI constants quoted in error texts
    assert read <= env <= read | named_in_messages, env
    # ... and the test build has both hooks
    hooks = ctypes.CDLL(HOOKS_SO)
    assert hasattr(hooks, "vgen_debug_fail_after") and b"VGEN_DEBUG_GTAB_FAIL" in open(HOOKS_SO, "rb").read()
    assert hooks.vgen_abi_version() == lib.vgen_abi_version()


def test_no_device_is_a_loud_error_not_a_cpu_fallback():
    import vgen_amd as vg
    if vg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(vg.VgenError) as e:
        vg.GpuRunner()
    assert e.value.status == -2 and "no CPU fallback" in str(e.value)


def test_key_variants_of_endomorphism_contexts_are_lambda_multiples_mod_n():
    """vgen_key_variant (host/scalar.h: 256 x 256 -> 512-bit product folded mod n) against Python integers."""
    import random
    import vgen_amd as vg
    n = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    lam = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72
    assert pow(lam, 3, n) == 1
    rng = random.Random(1)
    keys = [1, 2, n - 1, n - 2, 2**255, lam, n - lam, (n + 1) // 2, 2**256 - n] + [rng.randrange(1, n) for _ in range(3000)]
    for k in keys:
        for v in range(6):
            want = pow(lam, v % 3, n) * k % n
            assert vg.key_variant(k, v) == (n - want if v >= 3 else want), (hex(k), v)
    with pytest.raises(vg.VgenError):
        vg.key_variant(0, 1)
    with pytest.raises(vg.VgenError):
        vg.key_variant(5, 6)


def test_random_key_stream_is_the_oracles():
    """vgen_random_key (core/rnd.h compiled for the host: the function the kernels run per lane) against the oracle's
    stream, restated with hashlib (oracle/pyoracle.py) and as walked by the C oracle's scan_random worker."""
    import random
    import vgen_amd as vg
    from oracle import pyoracle as vo
    n = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    rng = random.Random(3)
    cases = [(42, 0, 0), (42, 0, 1), (2**64 - 1, 2**32 - 1, 2**64 - 1), (0, 0, 0), (1, 2**31, 2**32), (7, 3, 2**40 + 5)]
    cases += [(rng.getrandbits(64), rng.getrandbits(32), rng.getrandbits(64)) for _ in range(2000)]
    # the 24-byte seeds of real (unseeded) searches — vgen_random_key_seed —, and their relation to the 64-bit ones
    cases += [(rng.randbytes(24), rng.getrandbits(32), rng.getrandbits(64)) for _ in range(2000)]
    cases += [(bytes(24), 0, 0), (b"\xff" * 24, 2**32 - 1, 2**64 - 1), (bytes(23) + b"\x01", 0, 0), (b"\x01" + bytes(23), 0, 0)]
    for seed, stream, index in cases:
        want = vo.random_key(seed, stream, index)
        assert vg.random_key(seed, stream, index) == (want if 0 < want < n else None)
    for s in (0, 1, 42, 2**64 - 1, rng.getrandbits(64)):
        assert vg.random_key(s, 3, 99) == vg.random_key(s.to_bytes(8, "little") + bytes(16), 3, 99)    # u64 = bytes 0..7, rest zero
    assert vg.random_key(bytes(23) + b"\x01", 0, 0) != vg.random_key(bytes(24), 0, 0)                    # every seed byte counts
    assert len({vg.random_key(bytes(i) + b"\x80" + bytes(23 - i), 1, 2) for i in range(24)}) == 24
    # known answer, computed by hand with hashlib: pins the message layout (tag, 24 seed bytes, stream, index; 52 bytes)
    import hashlib
    kat = hashlib.sha256(b"vgen-mi355x-rand" + bytes(range(24)) + (7).to_bytes(4, "little") + (2**40 + 5).to_bytes(8, "little")).digest()
    assert vg.random_key(bytes(range(24)), 7, 2**40 + 5) == int.from_bytes(kat, "big")
    # the C oracle's single worker walks stream 0 in index order: its matches are candidates of that stream, ascending
    res = vo.scan_random(0, "^1A", 42, count=4, threads=1)
    keys = [m["key"] for m in res["matches"]]
    idx = [i for i in range(2000) if vo.random_key(42, 0, i) in keys]
    assert len(idx) == 4 and [vo.random_key(42, 0, i) for i in idx] == keys


def test_host_side_helpers_match_oracle():
    import vgen_amd as vg
    from oracle import pyoracle as vo
    for k in (1, 2, 0xC0FFEE, 2**200 + 17):
        for fmt in (0, 1, 2, 3, 4, 5):
            g = vg.derive(fmt, k)
            o = vo.generate(fmt, k)
            assert (g.address, g.wif, g.hex) == (o["address"], o["wif"], o["hex"])
    n = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    assert vg.derive(0, 0) is None and vg.derive(0, n) is None
    # increment_key (gpu.rs:951-968)
    assert vg.key_add(1, 41) == 42 and vg.key_add(n - 2, 1) == n - 1 and vg.key_add(n - 1, 1) is None
    assert vg.key_add(2**256 - 1, 1) is None


WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import bench                                   # the product's own striping: bench.batch_key / bench.seed_key
import vgen_amd as vg                          # the product's host side of the boundary (no device needed for these calls)
from oracle import pyoracle as vo              # stands in for the device: payloads of each batch
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N, STEPS = 2048, 3                             # keys per batch, steps per rank  (global batches = STEPS * world)
k0 = bench.seed_key(42, 0)
assert k0 == vo.seed_key(42, 0)
pat = vg.Pattern("^1[A-F]", False, vg.AddressFormat.P2pkh)
mine, ops, starts = [], 0, []
for step in range(STEPS):
    start = bench.batch_key(k0, step, world, rank, N)          # what bench.py dispatches on this rank at this step
    starts.append(start)
    blob = vo.payload_seq(0, start, N, threads=1)
    for i in range(N):
        if pat.matches(vg.address_from_payload(0, blob[20 * i:20 * i + 20])):   # host confirm, as scan_shard does
            mine.append(vg.key_add(start, i))
    ops += N
dist.barrier()
gathered = [None] * world
dist.all_gather_object(gathered, (mine, ops, starts))   # host-side aggregation of match records and counters
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                # the max-over-ranks bench.py takes of the elapsed time
if rank == 0:
    merged = sorted(k for m, _, _ in gathered for k in m)
    print(json.dumps({"keys": [hex(k) for k in merged], "ops": sum(o for _, o, _ in gathered),
                      "starts": sorted(s for _, _, st in gathered for s in st), "max": t.item()}))
dist.destroy_process_group()
"""


def _torchrun(script, args, port, timeout=600, env_extra=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(env_extra or {}))
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)] + args,
                          capture_output=True, text=True, env=env, timeout=timeout)


def test_two_rank_batch_striping_equals_single_range_scan(tmp_path):
    """Two gloo ranks walk the batches bench.py's own batch_key() assigns them, confirm candidates with the
    product's host filter, and the merged result must be the oracle's scan of the whole range: the striping is
    disjoint, complete and in the order SURVEY.md 8(e) states."""
    import json
    from oracle import pyoracle as vo
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = _torchrun(script, [ROOT], 29517)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    k0 = vo.seed_key(42, 0)
    assert got["starts"] == [k0 + b * 2048 for b in range(6)]          # contiguous, disjoint, nothing skipped
    ref = vo.scan_range(0, "^1[A-F]", k0, k0 + 6 * 2048 - 1, count=10**9, threads=2)
    assert got["ops"] == 6 * 2048 == ref["operations"] and got["max"] == 2.0
    assert got["keys"] == [hex(m["key"]) for m in ref["matches"]]


RANK_REPORT_WORKER = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import bench
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
pin = bench.numa_pin(rank, sysfs=sys.argv[2])                  # a fake sysfs tree: two GPUs on two NUMA nodes
dist.barrier()
t0 = time.perf_counter()
time.sleep(0.05 * (rank + 1))                                  # rank 1 is the slow one
t1 = time.perf_counter()
mine = {"rank": rank, "device": rank, "value": 1000.0 / (rank + 1), "sustained": 1100.0, "region_mhz": 2300 + rank, "sustained_mhz": 2350,
        "host_dispatch_us": 10.5 + rank, "host_wait_us": 70.0, "numa_node": pin.get("numa_node"), "cpus": pin.get("cpus"), "pinned": pin.get("pinned"),
        "t0": t0, "t1": t1}
per_rank = [None] * world
dist.all_gather_object(per_rank, mine)
if rank == 0:
    cols, timing = bench.rank_report(per_rank)
    print(json.dumps({"per_rank": cols, "timing": timing, "affinity_now": sorted(os.sched_getaffinity(0))}))
dist.destroy_process_group()
"""


def _fake_sysfs(root, cpus_by_gpu):
    """A KFD topology with one CPU node and len(cpus_by_gpu) GPU nodes, each GPU on a NUMA node of its own."""
    nodes = root / "class/kfd/kfd/topology/nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 16\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, cpus in enumerate(cpus_by_gpu):
        (nodes / str(i + 1)).mkdir()
        bus = 0x05 + 0x20 * i
        (nodes / str(i + 1) / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
        dev = root / "bus/pci/devices" / ("0000:%02x:00.0" % bus)
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{i}\n")
        (dev / "local_cpulist").write_text(cpus + "\n")


def test_numa_pinning_reads_the_kfd_topology_and_never_fails(tmp_path):
    """bench.numa_pin: HIP device order from the KFD topology (no HIP call), the device's NUMA node and its cores from the PCI
    device's sysfs entry; anything unreadable leaves the affinity alone and says why."""
    import bench
    before = os.sched_getaffinity(0)
    try:
        cpus = sorted(before)
        half = max(1, len(cpus) // 2)
        a, b = bench.format_cpulist(cpus[:half]), bench.format_cpulist(cpus[half:] or cpus[:1])
        _fake_sysfs(tmp_path, [a, b])
        assert bench.gpu_pci_addresses(str(tmp_path)) == ["0000:05:00.0", "0000:25:00.0"]
        pin = bench.numa_pin(1, sysfs=str(tmp_path))
        if len(cpus) > 1:
            assert pin["pinned"] and pin["numa_node"] == 1 and pin["pci"] == "0000:25:00.0" and pin["cpus"] == b
            assert os.sched_getaffinity(0) == bench.parse_cpulist(b)
        os.sched_setaffinity(0, before)
        # a launcher's index list is honoured; UUIDs, a missing topology or a device without a node change nothing
        os.environ["HIP_VISIBLE_DEVICES"] = "1,0"
        assert bench.numa_pin(0, sysfs=str(tmp_path))["pci"] == "0000:25:00.0"
        os.sched_setaffinity(0, before)
        os.environ["HIP_VISIBLE_DEVICES"] = "GPU-deadbeef"
        pin = bench.numa_pin(0, sysfs=str(tmp_path))
        assert not pin["pinned"] and "ValueError" in pin["why_not"] and os.sched_getaffinity(0) == before
        del os.environ["HIP_VISIBLE_DEVICES"]
        pin = bench.numa_pin(0, sysfs=str(tmp_path / "nowhere"))
        assert not pin["pinned"] and "why_not" in pin and os.sched_getaffinity(0) == before
        (tmp_path / "bus/pci/devices/0000:05:00.0/numa_node").write_text("-1\n")
        pin = bench.numa_pin(0, sysfs=str(tmp_path))
        assert not pin["pinned"] and pin["numa_node"] == -1 and os.sched_getaffinity(0) == before
        assert bench.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and bench.format_cpulist({0, 1, 2, 3, 8, 10, 11}) == "0-3,8,10-11"
    finally:
        os.environ.pop("HIP_VISIBLE_DEVICES", None)
        os.sched_setaffinity(0, before)


def test_two_ranks_gather_the_per_rank_arrays_of_the_n_gt_1_line(tmp_path):
    """The N > 1 bench line's per-rank arrays (bench.rank_report) over two gloo ranks on the CPU: every field arrives in rank
    order, the skews are measured against the earliest start on the node's shared monotonic clock, and every rank pinned
    itself to its own GPU's NUMA node before anything else (here: a fake sysfs tree with two GPUs on two nodes)."""
    import json
    import bench
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 2:
        pytest.skip("needs two CPU cores")
    half = len(cpus) // 2
    _fake_sysfs(tmp_path, [bench.format_cpulist(cpus[:half]), bench.format_cpulist(cpus[half:])])
    script = tmp_path / "rank_report_worker.py"
    script.write_text(RANK_REPORT_WORKER)
    out = _torchrun(script, [ROOT, str(tmp_path)], 29523)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    pr, t = got["per_rank"], got["timing"]
    assert set(pr) == {"rank", "device", "value", "sustained", "region_mhz", "sustained_mhz", "host_dispatch_us", "host_wait_us", "numa_node", "cpus",
                       "pinned", "t0_us", "t1_us"}
    assert pr["rank"] == [0, 1] and pr["value"] == [1000.0, 500.0] and pr["region_mhz"] == [2300, 2301] and pr["host_dispatch_us"] == [10.5, 11.5]
    assert pr["numa_node"] == [0, 1] and pr["pinned"] == [True, True]
    assert pr["cpus"] == [bench.format_cpulist(cpus[:half]), bench.format_cpulist(cpus[half:])] and got["affinity_now"] == cpus[:half]
    assert min(pr["t0_us"]) == 0.0 and 0 <= t["start_skew_us"] == max(pr["t0_us"]) < 20000          # both left the barrier within milliseconds
    assert 40000 < pr["t1_us"][0] < pr["t1_us"][1] and 30000 < t["finish_skew_us"] < 90000             # rank 1 slept 50 ms longer


GPU_WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import vgen_amd as vg
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, device=0, frames=3)
cfg = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, start=1, end=0x2FFFF, shard=rank, n_shards=world)
res = vg.scan_gpu_with_runner("^1[A-C]", cfg, r)          # the product's own striping: scan_shard(shard, n_shards)
r.close()
gathered = [None] * world
dist.all_gather_object(gathered, ([m.hex for m in res.matches], res.operations, res.complete))
if rank == 0:
    print(json.dumps({"keys": sorted(k for m, _, _ in gathered for k in m), "ops": [o for _, o, _ in gathered],
                      "complete": [c for _, _, c in gathered]}))
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_ranks_striping_one_scan_through_vgen_scan_shards(tmp_path):
    """The N>1 shape of the product itself: two processes (ranks), each driving its own context through
    vgen_scan with shard = rank, n_shards = 2 (here both on GPU 0), merged on the host over gloo."""
    import json
    from oracle import pyoracle as vo
    script = tmp_path / "gpu_worker.py"
    script.write_text(GPU_WORKER)
    out = _torchrun(script, [ROOT], 29519)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    ref = vo.scan_range(0, "^1[A-C]", 1, 0x2FFFF, count=10**9)
    assert got["keys"] == [m["hex"] for m in ref["matches"]]
    assert got["ops"] == [12 * 8192, 12 * 8192] and got["complete"] == [True, True]   # 24 batches of 8192 cover 1..0x2FFFF


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_prints_one_contract_line():
    """bench.py's N>1 path (torchrun, gloo barrier / max over ranks, batch striping) run for real: two ranks
    sharing the one GPU of the test box (VGEN_BENCH_REHEARSE=1), rank 0 prints the single JSON line."""
    import json
    out = _torchrun(os.path.join(ROOT, "bench.py"), ["--gpus", "2", "--steps", "64", "--warmup", "8", "--sustained-seconds", "0.3", "--multi-leg-seconds", "1"],
                    29521, env_extra={"VGEN_BENCH_REHEARSE": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, out.stdout[:500]      # ONE line on stdout: gloo's own chatter must not reach it
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 64 and d["scaling"] == "weak" and d["value"] > 1000
    assert d["config"]["parallelism"].startswith("range-striped x2") and d["sustained"]["value"] > 1000
    assert "cpu_baseline" not in d and "other_configs" not in d
    t = d["timing"]   # whole-job wall time: latest finish - earliest start over the ranks, never less than the slowest rank's own time
    assert t["elapsed_ms"] >= t["slowest_rank_ms"] > 0 and abs(t["elapsed_ms"] / 64 - d["ms_per_step"]) < 1e-3
    # per-rank arrays: a sub-linear point must name its cause from one run
    pr = d["per_rank"]
    assert pr["rank"] == [0, 1] and all(v > 500 for v in pr["value"]) and all(v > 500 for v in pr["sustained"])
    assert all(1000 < m < 2600 for m in pr["region_mhz"]) and all(0 < h < 200 for h in pr["host_dispatch_us"]) and all(h >= 0 for h in pr["host_wait_us"])
    assert min(pr["t0_us"]) == 0.0 and t["start_skew_us"] == max(pr["t0_us"]) and t["finish_skew_us"] >= 0
    assert abs((max(pr["t1_us"]) - 0.0) / 1e3 - t["elapsed_ms"]) < 0.01                       # latest finish - earliest start
    assert "cpu_affinity" in d["config"] and "cpus" in d["config"]["cpu_affinity"]
    # the in-process multi-device path ran once over every visible device (one here), in a child, after the ranks' contexts closed
    m = d["in_process_multi"]
    assert m.get("error") is None and m["n_devices"] >= 1 and m["failed_shards"] == 0 and m["value"] > 1000 and m["matches"] == 0


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        if not (env_extra and k in env_extra):
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_gpus_flag_and_launcher_world_size_must_agree():
    """`--gpus` is the number of ranks: a launcher that started another number is an error, decided before torch or HIP
    are touched (so it is testable without a device)."""
    out = _bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=60)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and out.stdout == ""
    out = _bench(["--gpus", "0"], timeout=60)
    assert out.returncode != 0


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """Plain `python bench.py --gpus 2` (no launcher) must start two ranks as a child process.  Without a device every rank
    refuses loudly (no CPU fallback) and the launcher passes the failure on; on a GPU box the twin test below gets the line."""
    import vgen_amd as vg
    if vg.device_count() > 0:
        pytest.skip("a GPU is present: covered by test_bench_plain_gpus_2_prints_n_gpus_2")
    out = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1"], timeout=300)
    assert out.returncode != 0 and out.stdout.strip() == ""
    # both ranks were started; the first to refuse says why (the launcher may stop the other before it gets to say so too)
    assert out.stderr.count("needs an MI355X") >= 1, out.stderr[-1500:]
    assert "local_rank: 0" in out.stderr and "local_rank: 1" in out.stderr, out.stderr[-1500:]


@pytest.mark.gpu
def test_bench_plain_gpus_2_prints_n_gpus_2():
    """The driver's shape of command without a launcher: `python bench.py --gpus 2` starts its two ranks itself (here
    sharing the test box's one GPU, VGEN_BENCH_REHEARSE=1) and relays ONE line with n_gpus = 2; without the rehearsal
    switch a box with fewer GPUs than ranks is refused instead of silently measuring something else."""
    import json
    import vgen_amd as vg
    args = ["--gpus", "2", "--steps", "64", "--warmup", "8", "--sustained-seconds", "0.3", "--multi-leg-seconds", "0"]
    out = _bench(args, {"VGEN_BENCH_REHEARSE": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, out.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 64 and d["value"] > 1000 and d["config"]["parallelism"].startswith("range-striped x2")
    if vg.device_count() < 2:
        out = _bench(args)
        # (a box that exports *_VISIBLE_DEVICES itself is taken for a pinning launcher and refused one step later)
        assert out.returncode != 0 and out.stdout.strip() == "" and ("needs 2 devices" in out.stderr or "the ranks sit on 1 device(s)" in out.stderr)
        # a launcher that pins one device per rank shows every rank a single device 0: accepted, but the ranks must then
        # sit on different devices — here both were given the same one
        out = _bench(args, {"HIP_VISIBLE_DEVICES": "0"})
        assert out.returncode != 0 and out.stdout.strip() == "" and "the ranks sit on 1 device(s)" in out.stderr, out.stderr[-1500:]


def test_cli_warns_about_impossible_patterns_before_touching_the_device():
    """lib.rs:684-706.  Without a device the command then fails loudly (no CPU scan path in this build)."""
    import subprocess
    import vgen_amd as vg
    exe = os.path.join(os.path.dirname(vg.library_path()), "vgen-hip")
    out = subprocess.run([exe, "generate", "-p", "^bc1qB", "-f", "p2wpkh"], capture_output=True, text=True, timeout=60)
    assert "Warning: Pattern contains characters not valid in Bech32 addresses: 'b1B'" in out.stderr
    assert "Base58 excludes" not in out.stderr
    out = subprocess.run([exe, "generate", "-p", "^1Cat"], capture_output=True, text=True, timeout=60)
    assert "Warning" not in out.stderr


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """The boundary is a C ABI: include/vgen_hip.h must compile as C99 and a C program must link against
    libvgen_hip.so (no C++ types or torch in the signatures).  Only calls that need no device are made."""
    import vgen_amd as vg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "vgen_hip.h"\n'
                   'int main(void) {\n'
                   '  unsigned char k[32] = {0}, out[32]; char addr[128], wif[128], bad[8]; size_t n = 0; unsigned long long d = 0;\n'
                   '  k[31] = 1;\n'
                   '  if (vgen_abi_version() != VGEN_ABI_VERSION) return 1;\n'
                   '  if (vgen_derive(VGEN_FMT_P2PKH, k, addr, sizeof addr, wif, sizeof wif) != VGEN_OK) return 2;\n'
                   '  if (strcmp(addr, "1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH")) return 3;\n'
                   '  if (vgen_key_add(k, 41, out) != VGEN_OK || out[31] != 42) return 4;\n'
                   '  if (vgen_pattern_invalid_chars("^1O0", 0, VGEN_FMT_P2PKH, bad, sizeof bad, &n) != VGEN_OK || n != 2) return 5;\n'
                   '  if (vgen_pattern_difficulty("^1Ab", 0, VGEN_FMT_P2PKH, (uint64_t *)&d) != VGEN_OK || d != 3364) return 6;\n'
                   '  printf("%s %s\\n", addr, wif);\n  return 0;\n}\n')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(vg.library_path())
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src),
                         "-o", str(exe), "-L", libdir, "-lvgen_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0, (run.returncode, run.stderr)
    assert run.stdout.split()[0] == "1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH"
