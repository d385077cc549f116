"""How sensitive is the driver-shaped 20-step window to the HOST's speed?  The same loop as bench.py's timed region with an
artificial busy-wait before every dispatch (a slower host thread); prints Mkeys/s by added delay.
usage: python tools/host_delay_probe.py [delays_us ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, vgen_amd as vg
N, F, K = 1 << 20, 12, 20
fmt = vg.AddressFormat.P2pkh
r = vg.GpuRunner(batch_size=N, fmt=fmt, frames=F, timing=False)
r.set_filter(vg.Pattern("^1Cat", False, fmt))
k0 = bench.seed_key(42, 0)


def spin(us):
    if us <= 0:
        return
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass


def run_steps(first, n_steps, delay):
    issued = done = fi = fw = 0
    while issued < min(F, n_steps):
        spin(delay); r.dispatch(k0 + (first + issued) * N, fi); issued += 1; fi = (fi + 1) % F
    while done < n_steps:
        r.wait(fw); done += 1
        if issued < n_steps:
            spin(delay); r.dispatch(k0 + (first + issued) * N, fw); issued += 1
        fw = (fw + 1) % F


step = 0
run_steps(step, 64, 0); step += 64
for delay in [float(x) for x in sys.argv[1:]] or [0, 10, 20, 40, 80]:
    vals = []
    for rep in range(5):
        run_steps(step, 600, 0); step += 600          # steady state first, as bench.py's sustained leg does
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(step, K, delay); step += K
        torch.cuda.synchronize()
        vals.append(K * N / (time.perf_counter() - t0) / 1e6)
    print("added host delay %5.0f us per dispatch: %s Mkeys/s (median %.0f)" % (delay, " ".join("%.0f" % v for v in vals), sorted(vals)[2]), flush=True)
r.close()
