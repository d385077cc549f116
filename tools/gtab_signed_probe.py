"""Rates of the scalar-multiplication paths by generator-table width, signed windows included (profiles/r04_gtab_signed.txt).
usage: python tools/gtab_signed_probe.py [bits ...]      default: 24 26 25 27 29"""
import os
import sys
import time
sys.path.insert(0, ".")
import bench
import vgen_amd as vg

widths = [int(x) for x in sys.argv[1:]] or [24, 26, 25, 27, 29]
batch = 1 << 20
for bits in widths:
    os.environ["VGEN_GTAB_BITS"] = str(bits)
    # first use: allocation + build (a dispatch of the random stream on a fresh context)
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=2, timing=False)
    r.set_filter(vg.Pattern("^1Cat", False, vg.AddressFormat.P2pkh))
    t0 = time.perf_counter()
    r.dispatch_random(42, 0, 0, 0)
    r.wait(0)
    first = time.perf_counter() - t0
    res = r.resources()
    r.close()
    out = {"bits": bits, "table_bits_in_use": res["table_bits"], "first_dispatch_s": round(first, 3), "note": res["note"]}
    for name, kw in (("random", dict(random_stream=True)), ("random_endo", dict(random_stream=True, endo=True)), ("uploaded", dict())):
        e = bench.keys_mode_config(vg, batch, 8, 0, 2.0, **kw)
        out[name + "_Mkeys"] = e["value"]
    e = bench.timed_config(vg, "p2tr", "^bc1pqqq", False, batch, 12, 0, 2.0, "p2tr")
    out["p2tr_Mkeys"] = e["value"]
    print(out, flush=True)
