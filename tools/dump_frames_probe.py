import sys, time
sys.path.insert(0, ".")
import vgen_amd as vg
import bench
for frames in (2, 4, 2, 4, 8):
    r = vg.GpuRunner(batch_size=1 << 20, fmt=vg.AddressFormat.P2pkh, device=0, frames=frames, timing=False)
    r.set_filter(None)
    p = bench.Pipeline(r, bench.seed_key(42, 0))
    p.run_steps(2 * frames)
    n, dt = p.run_seconds(1.0)
    print(f"frames={frames}: {n * (1 << 20) / dt / 1e6:.1f} Mkeys/s = {n * (1 << 20) * 20 / dt / 1e9:.1f} GB/s")
    r.close()
