"""Timeline of the last 20-step window in a rocprofv3 kernel trace of tools/ramp_trace.py (tools/ramp_prof.sh): per dispatch the start and end
of its kernels relative to the window's first kernel, and how many seq_bwd / seq_hash launches are running over time.
usage: python tools/k20_timeline.py gpurun_out/ramp_prof/..._kernel_trace.csv [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("seq_fwd", "seq_inv", "seq_bwd", "seq_hash"):
        if k in n:
            return k
    return n[:24]
ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows]
per = len({k for k, _, _, _ in ks if k.startswith("seq_")})
# the window: the last K dispatches = the last K * per seq_* kernels
seq = [k for k in ks if k[0].startswith("seq_")][-K * per:]
t0 = min(s for _, s, _, _ in seq)
print("kernels per dispatch: %d; window %.1f us" % (per, (max(e for _, _, e, _ in seq) - t0) / 1e3))
byq = {}
for k, s, e, q in seq:
    byq.setdefault(q, []).append((k, (s - t0) / 1e3, (e - t0) / 1e3))
for q in sorted(byq, key=lambda q: byq[q][0][1]):
    print("queue %s: " % q + "  ".join("%s %.0f-%.0f" % x for x in byq[q]))
# utilisation: number of heavy launches (seq_bwd / seq_hash) running, in 25 us bins
end = max(e for _, _, e, _ in seq)
bins = int((end - t0) / 25e3) + 1
for name in ("seq_fwd", "seq_inv", "seq_bwd", "seq_hash"):
    occ = [0.0] * bins
    for k, s, e, _ in seq:
        if k != name:
            continue
        for b in range(bins):
            lo, hi = t0 + b * 25e3, t0 + (b + 1) * 25e3
            occ[b] += max(0.0, min(e, hi) - max(s, lo)) / 25e3
    if any(occ):
        print("%-8s running per 25 us bin: " % name + " ".join("%.1f" % o for o in occ))
for name in ("seq_fwd", "seq_inv", "seq_bwd", "seq_hash"):
    d = [(e - s) / 1e3 for k, s, e, _ in seq if k == name]
    if d:
        print("%-8s n=%d mean %.1f us min %.1f max %.1f" % (name, len(d), sum(d) / len(d), min(d), max(d)))
