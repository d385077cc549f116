"""Timeline of a short run (the driver's --steps 20): when is each dispatch issued, when does each wait return?
usage: ramp_trace.py [frames] [steps] [idle seconds before the window] [dispatches of steady load before it]"""
import sys, time
sys.path.insert(0, ".")
import vgen_amd as v
F = int(sys.argv[1]) if len(sys.argv) > 1 else 12
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = 1 << 20
r = v.GpuRunner(batch_size=N, fmt=v.AddressFormat.P2pkh, frames=F, timing=False)
r.set_filter(v.Pattern("^1Cat", False, v.AddressFormat.P2pkh))
key = 0x3a8ae174e51b7b1117ab406c6570970f453c4376b6d381977db7c02fb5a993e0
HEAT = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # dispatches of steady load before every window
for rep in range(3):
    for f in range(F):
        r.dispatch(key, f); key += N
    fw = 0
    for i in range(HEAT):
        r.wait(fw); r.dispatch(key, fw); key += N
        fw = (fw + 1) % F
    for f in range(F):
        r.wait((fw + f) % F)
    time.sleep(float(sys.argv[3]) if len(sys.argv) > 3 else 0.01)
    ev = []
    mhz = []
    t0 = time.perf_counter()
    issued = done = 0
    for f in range(min(F, K)):
        r.dispatch(key, f); key += N; issued += 1
        ev.append(("d", issued, (time.perf_counter() - t0) * 1e6))
    fw = 0
    while done < K:
        r.wait(fw); done += 1
        ev.append(("w", done, (time.perf_counter() - t0) * 1e6))
        cyc, tck = r.frame_clock(fw)
        mhz.append(round(cyc / tck * 100) if tck else 0)
        if issued < K:
            r.dispatch(key, fw); key += N; issued += 1
            ev.append(("d", issued, (time.perf_counter() - t0) * 1e6))
        fw = (fw + 1) % F
    total = (time.perf_counter() - t0) * 1e6
    print("F=%d K=%d total %.0f us = %.0f Mkeys/s" % (F, K, total, K * N / total))
print(" ".join("%s%d@%.0f" % e for e in ev))
print("shader MHz seen by block 0 of each seq_bwd launch, in wait order:", mhz)
