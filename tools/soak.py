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
