"""Random-key mode (vgen_dispatch_random) rate by frames in flight.   usage: python tools/rnd_frames.py [frames ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vgen_amd as vg
batch = 1 << 20
for F in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 12]:
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=F, timing=False)
    r.set_filter(vg.Pattern("^1Cat", False, vg.AddressFormat.P2pkh))
    ctr = 0
    for f in range(F):
        r.dispatch_random(42, 0, ctr * batch, f); ctr += 1
    for f in range(F):
        r.wait(f)
    t0 = time.perf_counter()
    issued = done = fw = 0
    for f in range(F):
        r.dispatch_random(42, 0, ctr * batch, f); ctr += 1; issued += 1
    while done < issued:
        r.wait(fw); done += 1
        if time.perf_counter() - t0 < 1.5:
            r.dispatch_random(42, 0, ctr * batch, fw); ctr += 1; issued += 1
        fw = (fw + 1) % F
    dt = time.perf_counter() - t0
    r.close()
    print("frames %2d: %.1f Mkeys/s" % (F, issued * batch / dt / 1e6), flush=True)
