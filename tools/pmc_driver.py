"""Driver for the counter passes of the scalar-multiplication paths (tools/pmc_keys.sh): a few dispatches on ONE frame, no torch.
usage: python3 tools/pmc_driver.py keys|random|random_endo|p2tr|seq [dispatches]      (VGEN_GTAB_BITS selects the table width)"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vgen_amd as vg  # noqa: E402

N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
batch = 1 << 20
fmt = vg.AddressFormat.P2tr if mode == "p2tr" else vg.AddressFormat.P2pkh
r = vg.GpuRunner(batch_size=batch, fmt=fmt, frames=1, endo=mode == "random_endo")
r.set_filter(vg.Pattern("^bc1pqqq" if mode == "p2tr" else "^1Cat", False, fmt))
if mode == "keys":
    # 2^20 DISTINCT random scalars (a block of keys repeated would turn the table gathers into cache hits; 32 random bytes are
    # a valid scalar except with probability 2^-128)
    blob = random.Random(42).randbytes(32 * batch)
k0 = int.from_bytes(__import__("hashlib").sha256(b"pmc").digest(), "big") % N_ORDER
for i in range(steps):
    if mode == "keys":
        r.dispatch_keys(blob, 0)
    elif mode in ("random", "random_endo"):
        r.dispatch_random(42, 0, i * batch, 0)
    else:
        r.dispatch(k0 + i * batch, 0)
    r.wait(0)
r.close()
print("done", mode, steps)
