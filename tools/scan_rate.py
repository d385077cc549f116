"""Throughput of the C++ scan loop (vgen_scan) itself, as the CLI and a host application drive it."""
import sys
sys.path.insert(0, ".")
import vgen_amd as v

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
r = v.GpuRunner(batch_size=1 << 20, fmt=v.AddressFormat.P2pkh, frames=frames, timing=False)   # as the CLI creates it
for pat, n in (("^1CatCatCat", 2048), ("^1CatCatCat", 8192), ("^1Cat", 8192)):
    res = v.scan_gpu_with_runner(pat, v.ScanConfig(count=None, seed=9, max_batches=n), r)
    print("vgen_scan %-12s %5d batches: %.0f Mkeys/s, %d matches" % (pat, n, res.rate() / 1e6, len(res.matches)))
