import time, sys
sys.path.insert(0, '.')
import vgen_amd as vg
for endo in (False, True):
    t0 = time.perf_counter()
    r = vg.GpuRunner(batch_size=1 << 20, fmt=vg.AddressFormat.P2pkh, endo=endo)
    t1 = time.perf_counter()
    r.set_filter(None)
    t2 = time.perf_counter()
    cfg = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=1) if endo else vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=1, start=1 << 65, end=(1 << 66) - 1)
    res = vg.scan_gpu_with_runner(".", cfg, r)
    t3 = time.perf_counter()
    res = vg.scan_gpu_with_runner(".", cfg, r)
    t4 = time.perf_counter()
    print(f"endo={endo} create {1e3*(t1-t0):.1f} ms, set_filter(None) {1e3*(t2-t1):.1f} ms, scan {1e3*(t3-t2):.1f} ms, scan again {1e3*(t4-t3):.1f} ms, elapsed field {res.elapsed_secs*1e3:.1f}")
    r.close()
