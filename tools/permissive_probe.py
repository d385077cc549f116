"""Where a permissive-pattern scan spends its host time (profiles/r04_permissive.txt).
usage: python tools/permissive_probe.py [pattern] [max_batches]     (with vgen_amd/libvgen_hip.so.prof copied over the library: per-phase times on stderr)"""
import sys
import time
sys.path.insert(0, ".")
import vgen_amd as vg

pat = sys.argv[1] if len(sys.argv) > 1 else "^1C"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
fmt = vg.AddressFormat.P2pkh
r = vg.GpuRunner(batch_size=1 << 20, fmt=fmt, frames=4, timing=False)
for rep in range(2):
    t0 = time.perf_counter()
    res = vg.scan_gpu_with_runner(pat, vg.ScanConfig(format=fmt, count=None, seed=42, max_batches=nb), r)
    wall = time.perf_counter() - t0
    print(f"{pat}: {res.operations} keys, {len(res.matches)} matches, vgen_scan {res.elapsed_secs:.3f} s = {res.operations / res.elapsed_secs / 1e6:.1f} Mkeys/s "
          f"(python wall {wall:.2f} s)", flush=True)
# count-limited: the first 100 matches of the same pattern
t0 = time.perf_counter()
res = vg.scan_gpu_with_runner(pat, vg.ScanConfig(format=fmt, count=100, seed=42), r)
print(f"{pat} -c 100: {res.operations} keys, {len(res.matches)} matches, {res.elapsed_secs * 1e3:.2f} ms")
r.close()
