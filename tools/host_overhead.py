"""Host-side cost of one step of the scan loop, call by call (run on the GPU box)."""
import sys, time
sys.path.insert(0, ".")
import vgen_amd as v

F = 12
TIMING = len(sys.argv) > 1 and sys.argv[1] == "timing"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20      # keys per dispatch
r = v.GpuRunner(batch_size=B, fmt=v.AddressFormat.P2pkh, frames=F, timing=TIMING)
r.set_filter(v.Pattern("^1Cat", False, v.AddressFormat.P2pkh))
key = 0x3a8ae174e51b7b1117ab406c6570970f453c4376b6d381977db7c02fb5a993e0
for f in range(F):
    r.dispatch(key, f); key += B
for f in range(F):
    r.wait(f)
# (1) dispatch cost with an idle device
t = time.perf_counter()
for f in range(F):
    r.dispatch(key, f); key += B
td = (time.perf_counter() - t) / F
time.sleep(0.05)   # everything has finished by now
# (2) wait cost when the frame is already complete
t = time.perf_counter()
for f in range(F):
    r.wait(f)
tw = (time.perf_counter() - t) / F
t = time.perf_counter()
for f in range(F):
    if TIMING: r.kernel_ms(f)
tk = (time.perf_counter() - t) / F
print("dispatch %.1f us, wait(completed) %.1f us, kernel_ms %.1f us per call" % (td * 1e6, tw * 1e6, tk * 1e6))
# (3) steady state: how long does wait() block, how long does dispatch take while the device is busy
N = 2048
tb = tdd = 0.0
for f in range(F):
    r.dispatch(key, f); key += B
t0 = time.perf_counter()
for s in range(N):
    f = s % F
    a = time.perf_counter(); r.wait(f); b = time.perf_counter(); r.dispatch(key, f); c = time.perf_counter()
    key += B
    tb += b - a; tdd += c - b
dt = time.perf_counter() - t0
for f in range(F):
    r.wait(f)
print("steady state: %.1f us/step = wait %.1f us + dispatch %.1f us  (%.0f Mkeys/s)" % (dt / N * 1e6, tb / N * 1e6, tdd / N * 1e6, N * B / dt / 1e6))
