"""Randomised dump-mode parity soak: many random start keys x formats x batch sizes against the oracle."""
import random, sys, time
sys.path.insert(0, ".")
import vgen_amd as vg
from oracle import pyoracle as vo

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t0 = time.time()
runs = keys = 0
for fmt in (0, 2, 3, 4, 5):
    for batch in (8192, 65536):
        r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=2)
        r.set_filter(None)
        t1 = time.time()
        while time.time() - t1 < budget / 10:
            start = rng.choice([rng.randrange(1, N - batch - 100), rng.randrange(1, 2**64), N - batch - rng.randrange(20, 5000),
                                rng.randrange(1, N) >> rng.randrange(0, 250) or 1])
            start = min(start, N - batch - 20)
            r.dispatch(start, 0)
            blob, _, _ = r.await_result(0)
            ref = vo.payload_seq(fmt, start, batch)
            assert blob == ref, (fmt, batch, hex(start))
            runs += 1; keys += batch
        r.close()
print("soak ok: %d dispatches, %d keys, %.0f s" % (runs, keys, time.time() - t0))

# second phase: twelve frames in flight at once (each on its own stream / hardware queue), all checked — no frame may
# see another's scratch, tables or results whatever the overlap
t1 = time.time()
runs2 = keys2 = 0
for fmt, batch in ((0, 16384), (5, 16384), (2, 32768), (3, 8192)):
    F = 12
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=F)
    r.set_filter(None)
    t2 = time.time()
    while time.time() - t2 < budget / 8:
        starts = [min(rng.randrange(1, N) >> rng.randrange(0, 200) or 1, N - batch - 20) for _ in range(F)]
        for f in range(F):
            r.dispatch(starts[f], f)
        for f in reversed(range(F)):          # consumed out of dispatch order on purpose
            blob, _, _ = r.await_result(f)
            assert blob == vo.payload_seq(fmt, starts[f], batch), (fmt, batch, f, hex(starts[f]))
            runs2 += 1; keys2 += batch
    r.close()
print("concurrent soak ok: %d dispatches (12 in flight), %d keys, %.0f s" % (runs2, keys2, time.time() - t1))

# third phase: endomorphism contexts — image 0 in full, the other five sampled, random bases, per format
t1 = time.time()
LAM = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72
runs3 = keys3 = 0
for fmt in (0, 1, 2, 4, 5):
    batch = 16384
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=2, endo=True)
    r.set_filter(None)
    t2 = time.time()
    while time.time() - t2 < budget / 10:
        start = min(rng.randrange(1, N) >> rng.randrange(0, 200) or 1, N - batch - 20)
        r.dispatch(start, 0)
        blob, _, tested = r.await_result(0)
        assert tested == 6 * batch
        assert blob[:20 * batch] == vo.payload_seq(fmt, start, batch), (fmt, hex(start))
        for _ in range(300):
            v, i = rng.randrange(1, 6), rng.randrange(batch)
            k = pow(LAM, v % 3, N) * (start + i) % N
            k = N - k if v >= 3 else k
            assert blob[20 * (v * batch + i):20 * (v * batch + i) + 20] == vo.payload(fmt, k), (fmt, hex(start), v, i)
        runs3 += 1; keys3 += 6 * batch
    r.close()
print("endomorphism soak ok: %d dispatches, %d keys (image 0 in full, 300 sampled images per dispatch), %.0f s" % (runs3, keys3, time.time() - t1))

# fourth phase: the random-key stream (vgen_dispatch_random), one key per draw and six per draw, random seeds / streams / indices
t1 = time.time()
runs4 = keys4 = 0
for fmt, endo in ((0, False), (0, True), (5, True), (3, False), (2, False), (4, True)):
    batch = 8192
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=2, endo=endo)
    r.set_filter(None)
    pl = 32 if fmt == 3 else 20
    t2 = time.time()
    while time.time() - t2 < budget / 12:
        seed, stream = rng.getrandbits(64), rng.getrandbits(32)
        first = rng.choice([0, rng.getrandbits(64) % (2**64 - batch), 2**64 - batch, rng.getrandbits(20)])
        r.dispatch_random(seed, stream, first, 0)
        blob, _, tested = r.await_result(0)
        assert tested == (6 if endo and fmt != 3 else 1) * batch
        for _ in range(400):
            i = rng.randrange(batch)
            k = vo.random_key(seed, stream, first + i)
            v = rng.randrange(6) if endo and fmt != 3 else 0
            kv = pow(LAM, v % 3, N) * k % N
            kv = N - kv if v >= 3 else kv
            want = vo.payload(fmt, kv) if 0 < k < N else bytes(pl)
            assert blob[pl * (v * batch + i):pl * (v * batch + i) + pl] == want, (fmt, endo, seed, stream, first, v, i)
        runs4 += 1; keys4 += tested
    r.close()
print("random-key soak ok: %d dispatches, %d keys (400 sampled per dispatch), %.0f s" % (runs4, keys4, time.time() - t1))
