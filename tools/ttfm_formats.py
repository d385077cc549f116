import sys, time
sys.path.insert(0, ".")
import vgen_amd as v
for fmt, pat in ((v.AddressFormat.P2tr, "^bc1pqqq"), (v.AddressFormat.P2wpkh, "^bc1qqqq"), (v.AddressFormat.Ethereum, "^0xdead")):
    v.GpuRunner(batch_size=1 << 20, fmt=fmt, frames=12, timing=False).close()
    for rep in range(3):
        t = time.perf_counter()
        r = v.GpuRunner(batch_size=1 << 20, fmt=fmt, frames=12, timing=False)
        created = time.perf_counter() - t
        res = v.scan_gpu_with_runner(pat, v.ScanConfig(format=fmt, count=1, seed=100 + rep, case_insensitive=True), r)
        cold = time.perf_counter() - t
        t = time.perf_counter()
        res2 = v.scan_gpu_with_runner(pat, v.ScanConfig(format=fmt, count=1, seed=200 + rep, case_insensitive=True), r)
        warm = time.perf_counter() - t
        r.close()
        print("%-10s %-9s create %.2f ms, cold %.2f ms (%d keys)   warm %.2f ms (%d keys)" % (fmt.name, pat, created * 1e3, cold * 1e3, res.operations, warm * 1e3, res2.operations))
