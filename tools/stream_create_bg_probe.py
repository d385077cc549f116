"""Does creating streams (hardware queues, ~8 ms each) on another thread stall dispatches of the main thread?"""
import ctypes, sys, threading, time
sys.path.insert(0, ".")
import vgen_amd as v
hip = ctypes.CDLL("libamdhip64.so")
N = 1 << 20
r = v.GpuRunner(batch_size=N, fmt=v.AddressFormat.P2pkh, frames=12, timing=False)
r.set_filter(v.Pattern("^1CatCatCat", False, v.AddressFormat.P2pkh))
key = 0x3a8ae174e51b7b1117ab406c6570970f453c4376b6d381977db7c02fb5a993e0
for _ in range(20):
    r.dispatch(key, 0); key += N; r.wait(0)
made = []

def maker():
    for i in range(11):
        s = ctypes.c_void_p()
        prio = (0, -1, 1)[((i + 1) // 4) % 3]
        t = time.perf_counter()
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, prio)
        made.append((rc, (time.perf_counter() - t) * 1e3))

for label, bg in (("no background thread", False), ("11 streams being created on another thread", True)):
    th = threading.Thread(target=maker) if bg else None
    if th:
        th.start()
    its = []
    t_end = time.perf_counter() + 0.12
    while time.perf_counter() < t_end:
        t = time.perf_counter(); r.dispatch(key, 0); key += N; r.wait(0); its.append((time.perf_counter() - t) * 1e3)
    if th:
        th.join()
    its.sort()
    print("%-45s %d dispatch+wait rounds: median %.3f ms, p99 %.3f ms, max %.3f ms" % (label, len(its), its[len(its) // 2], its[int(len(its) * 0.99)], its[-1]))
print("stream creations (rc, ms):", [(rc, round(ms, 1)) for rc, ms in made])
