"""Time to first match, cold (new context + scan) and warm (same context again), a few patterns."""
import sys, time
sys.path.insert(0, ".")
import vgen_amd as v
fmt = v.AddressFormat.P2pkh
v.GpuRunner(batch_size=1 << 20, fmt=fmt, frames=12, timing=False).close()      # HIP itself is up
for pat in ("^1Cat", "^1Cats", "^1CatsX"):
    for rep in range(3):
        t = time.perf_counter()
        r = v.GpuRunner(batch_size=1 << 20, fmt=fmt, frames=12, timing=False)
        created = time.perf_counter() - t
        res = v.scan_gpu_with_runner(pat, v.ScanConfig(format=fmt, count=1, seed=100 + rep), r)
        cold = time.perf_counter() - t
        t = time.perf_counter()
        res2 = v.scan_gpu_with_runner(pat, v.ScanConfig(format=fmt, count=1, seed=200 + rep), r)
        warm = time.perf_counter() - t
        r.close()
        print("%-8s create %.2f ms, cold %.2f ms (%d keys)   warm %.2f ms (%d keys)" % (pat, created * 1e3, cold * 1e3, res.operations, warm * 1e3, res2.operations))
