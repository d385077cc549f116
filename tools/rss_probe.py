"""Resident host memory of a process by stage: import, context, first dispatch, scan.  usage: python tools/rss_probe.py"""
import sys, time
sys.path.insert(0, ".")
def rss():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS"): return int(ln.split()[1]) // 1024
def smaps_top(n=6):
    rows = []; cur = None
    for ln in open("/proc/self/smaps"):
        p = ln.split()
        if len(p) >= 5 and "-" in p[0] and ":" not in p[0]: cur = " ".join(p[5:]) or "[anon]"
        elif ln.startswith("Rss:"): rows.append((int(p[1]) // 1024, cur))
    agg = {}
    for r, name in rows: agg[name] = agg.get(name, 0) + r
    return sorted(((v, k) for k, v in agg.items()), reverse=True)[:n]
print("start", rss())
import vgen_amd as vg
print("import vgen_amd", rss())
n = vg.device_count(); print("device_count", rss())
r = vg.GpuRunner(batch_size=1 << 20, fmt=vg.AddressFormat.P2pkh, frames=12, timing=False)
print("context (12 frames)", rss())
r.set_filter(vg.Pattern("^1CatCatCat", False, vg.AddressFormat.P2pkh)); print("set_filter", rss())
r.dispatch(12345, 0); r.wait(0); print("first dispatch", rss())
for f in range(12): r.dispatch(12345 + (f << 20), f)
for f in range(12): r.wait(f)
print("all frames used", rss())
res = vg.scan_gpu_with_runner("^1CatCatCat", vg.ScanConfig(count=1, seed=9, max_batches=2000), r); print("scan", rss())
print(smaps_top())
r.close(); print("closed", rss())
