"""A host that exits without vgen_destroy while the stream helper thread is still at work must exit cleanly."""
import sys
sys.path.insert(0, ".")
import vgen_amd as v
fmt = v.AddressFormat.P2pkh
r = v.GpuRunner(batch_size=1 << 18, fmt=fmt, frames=12, timing=False)
res = v.scan_gpu_with_runner("^1CatCatCat", v.ScanConfig(format=fmt, count=1, seed=5, max_batches=6), r)
print("scanned", res.operations, "keys; exiting without close()")
r._h = None          # (keep the Python wrapper from destroying the context: the handle is leaked on purpose)
sys.stdout.flush()
