"""Contexts that come and go while their stream helper thread is still at work: short scans on fresh contexts."""
import sys, time
sys.path.insert(0, ".")
import vgen_amd as v
fmt = v.AddressFormat.P2pkh
t0 = time.perf_counter()
for i in range(12):
    r = v.GpuRunner(batch_size=1 << 18, fmt=fmt, frames=12, timing=False)
    pat = ("^1Cat", "^1Cats", "^1CatsX")[i % 3]
    res = v.scan_gpu_with_runner(pat, v.ScanConfig(format=fmt, count=1, seed=300 + i, max_batches=(40, 400, 1200)[i % 3]), r)
    t = time.perf_counter(); r.close(); tc = time.perf_counter() - t
    print("%-8s %d matches, %d keys, scan %.1f ms, close %.1f ms" % (pat, len(res.matches), res.operations, res.elapsed_secs * 1e3, tc * 1e3))
print("total %.2f s" % (time.perf_counter() - t0))
