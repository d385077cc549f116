"""One-time costs of a context: creation, the first dispatch of every frame (its stream is created then), first scan."""
import sys, time
sys.path.insert(0, ".")
import vgen_amd as v
N = 1 << 20
t = time.perf_counter()
r = v.GpuRunner(batch_size=N, fmt=v.AddressFormat.P2pkh, frames=12, timing=False)
print("vgen_create: %.1f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); r.set_filter(v.Pattern("^1CatCatCat", False, v.AddressFormat.P2pkh)); print("filter compile + set: %.1f ms" % ((time.perf_counter() - t) * 1e3))
key = 0x3a8ae174e51b7b1117ab406c6570970f453c4376b6d381977db7c02fb5a993e0
for f in range(12):
    t = time.perf_counter(); r.dispatch(key, f); key += N
    print("first dispatch on frame %d: %.2f ms" % (f, (time.perf_counter() - t) * 1e3))
for f in range(12):
    r.wait(f)
t = time.perf_counter()
for f in range(12):
    r.dispatch(key, f); key += N
print("second round, 12 dispatches: %.2f ms" % ((time.perf_counter() - t) * 1e3))
for f in range(12):
    r.wait(f)
for n in (256, 256, 2048):
    t = time.perf_counter()
    res = v.scan_gpu_with_runner("^1CatCatCat", v.ScanConfig(count=None, seed=9, max_batches=n), r)
    dt = time.perf_counter() - t
    print("vgen_scan %d batches: %.1f ms wall (%.1f ms ideal at 12.4 G), engine-reported %.1f ms" % (n, dt * 1e3, n * N / 12.4e9 * 1e3, res.elapsed_secs * 1e3))
