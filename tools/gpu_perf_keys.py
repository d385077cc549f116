"""KEYS-mode throughput probe (arbitrary scalars, full k*G per key)."""
import sys, time, os, random
sys.path.insert(0, ".")
import vgen_amd as v
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
batch = 1 << 20
rng = random.Random(1)
blob = bytes(rng.getrandbits(8) for _ in range(32 * 4096)) * (batch // 4096)
keys = [blob[32*i:32*i+32] for i in range(batch)]
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
r = v.GpuRunner(batch_size=batch, fmt=v.AddressFormat.P2pkh, frames=F)
p = v.Pattern("^1Cat", False, v.AddressFormat.P2pkh)
r.set_filter(p)
import ctypes
from vgen_amd import api
def disp(f):
    api._check(api._L.vgen_dispatch_keys(r._h, f, blob, batch), r._h)
for f in range(F): disp(f)      # warm-up on every frame (streams and scratch are created on first use)
for f in range(F): r.wait(f)
t0 = time.perf_counter()
steps = 8 * F
for f in range(F): disp(f)
for s in range(steps):
    f = s % F
    r.wait(f)
    if s + F < steps: disp(f)
dt = time.perf_counter() - t0
print("KEYS mode, %d frames: %.1f Mkeys/s, kernel %.3f ms per 2^20 keys" % (F, steps * batch / dt / 1e6, r.kernel_ms(0)))
