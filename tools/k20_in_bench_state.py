"""The 20-step region measured through bench.Pipeline and through an inline loop, alternately, right after a heat
period — to separate what the region costs from what bench.py's surroundings add."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
import vgen_amd as vg
N = 1 << 20
F = 12
K = 20
fmt = vg.AddressFormat.P2pkh
r = vg.GpuRunner(batch_size=N, fmt=fmt, frames=F, timing=False)
r.set_filter(vg.Pattern("^1Cat", False, fmt))
p = bench.Pipeline(r, bench.seed_key(42, 0))
p.run_steps(F)


sync_us = []


def inline(sync):
    key = bench.batch_key(p.k0, p.next_step, 1, 0, N)
    t0 = time.perf_counter()
    issued = done = fw = 0
    for f in range(min(F, K)):
        r.dispatch(key, f); key += N; issued += 1
    while done < K:
        r.wait(fw); done += 1
        if issued < K:
            r.dispatch(key, fw); key += N; issued += 1
        fw = (fw + 1) % F
    t1 = time.perf_counter()
    if sync:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sync_us.append((time.perf_counter() - t1) * 1e6)
    p.next_step += K
    return K * N / dt / 1e6


def through_pipeline(sync):
    t0 = time.perf_counter()
    p.run_steps(K)
    if sync:
        torch.cuda.synchronize()
    return K * N / (time.perf_counter() - t0) / 1e6


for label, fn, sync, pre in (("inline, no torch sync", inline, False, False), ("inline + torch sync after", inline, True, False),
                             ("inline, torch sync before+after", inline, True, True),
                             ("Pipeline.run_steps, no sync", through_pipeline, False, False),
                             ("Pipeline.run_steps, sync before+after", through_pipeline, True, True)):
    out = []
    for rep in range(5):
        p.run_seconds(0.5)
        if pre:
            torch.cuda.synchronize()
        out.append(fn(sync))
    print("%-42s" % label, " ".join("%.0f" % x for x in out), "Mkeys/s", "| closing sync us:", " ".join("%.0f" % x for x in sync_us[-5:]) if fn is inline else "")
