import sys, time
sys.path.insert(0, ".")
import vgen_amd as vg
t0 = time.perf_counter()
r = vg.GpuRunner(batch_size=1 << 20, fmt=vg.AddressFormat.P2tr, frames=12)
t1 = time.perf_counter()
res = vg.scan_gpu_with_runner("^bc1pqq", vg.ScanConfig(format=vg.AddressFormat.P2tr, count=1, seed=5), r)
t2 = time.perf_counter()
print("create %.1f ms, first match %.1f ms after create (%d keys)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, res.operations))
r.close()
