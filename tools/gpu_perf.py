"""Throughput probe: K dispatches of `batch` keys with `frames` in flight, filter mode (bring-up tool)."""
import sys, time, os
sys.path.insert(0, ".")
import vgen_amd as v

def run(batch, frames, steps, fmt=0, pattern="^1Cat", ci=False):
    r = v.GpuRunner(batch_size=batch, fmt=v.AddressFormat(fmt), frames=frames)
    p = v.Pattern(pattern, ci, v.AddressFormat(fmt))
    r.set_filter(p)
    key = 0x3a8ae174e51b7b1117ab406c6570970f453c4376b6d381977db7c02fb5a993e0
    # warm-up
    for f in range(frames):
        r.dispatch(key, f); key += batch
    for f in range(frames):
        r.wait(f)
    t0 = time.perf_counter()
    nf = 0
    td = 0.0
    for i in range(frames):
        r.dispatch(key, i); key += batch
    done = 0; f = 0; kms = []
    while done < steps:
        n, _ = r.wait(f); nf += n; kms.append(r.kernel_ms(f)); done += 1
        if done + frames - 1 < steps:
            t1 = time.perf_counter(); r.dispatch(key, f); td += time.perf_counter() - t1; key += batch
        f = (f + 1) % frames
    # note: the last frames-1 waits above already consumed
    dt = time.perf_counter() - t0
    kms.sort()
    print("S=%s WG=%s PREG=%s batch=2^%d frames=%d steps=%d: %.1f Mkeys/s  (%.3f ms/step wall, kernel median %.3f ms) cand=%d host-dispatch %.3f ms/step" % (
        os.environ.get("VGEN_SEQ_S", "4"), "256", "-", batch.bit_length() - 1, frames, steps, steps * batch / dt / 1e6, dt / steps * 1e3,
        kms[len(kms) // 2], nf, td / max(1, steps - frames) * 1e3), flush=True)
    r.close()

if __name__ == "__main__":
    fmt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    sweep = [(1 << 20, 1, 64), (1 << 20, 2, 128), (1 << 20, 4, 128), (1 << 20, 8, 256), (1 << 24, 2, 16)]
    if len(sys.argv) > 2:
        b = int(os.environ.get("VGEN_PERF_BATCH", str(1 << 20)))
        sweep = [(b, int(f), max(32, int(os.environ.get("VGEN_PERF_STEPS", "256")) * (1 << 20) // b)) for f in sys.argv[2].split(",")]
    for batch, frames, steps in sweep:
        run(batch, frames, steps, fmt, os.environ.get("VGEN_PERF_PATTERN", "^0xdead" if fmt == 5 else "^1Cat"))
