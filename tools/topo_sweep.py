"""Stage-stream topology sweep: throughput of the sequential scan for VGEN_STREAMS x VGEN_SEQ_S x frames
(x VGEN_FUSED_INV), in one process (GPU_MAX_HW_QUEUES is whatever the caller exported before starting it).

    python tools/topo_sweep.py [--topos frame,0:2,1:2,2:2] [--s 4,8] [--frames 2,4,8,16] [--fused 1,0] [--steps 1024]

One line per configuration: Mkeys/s over `steps` dispatches of 2^20 keys (P2PKH '^1Cat'), median seq_bwd time.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vgen_amd as v  # noqa: E402


def run(batch, frames, steps, fmt, pattern, ci):
    r = v.GpuRunner(batch_size=batch, fmt=v.AddressFormat(fmt), frames=frames)
    r.set_filter(v.Pattern(pattern, ci, v.AddressFormat(fmt)))
    key = 0x3a8ae174e51b7b1117ab406c6570970f453c4376b6d381977db7c02fb5a993e0
    for _ in range(2):   # warm-up: streams, events, clocks
        for f in range(frames):
            r.dispatch(key, f)
            key += batch
        for f in range(frames):
            r.wait(f)
    t0 = time.perf_counter()
    issued = 0
    for f in range(min(frames, steps)):
        r.dispatch(key, f)
        key += batch
        issued += 1
    done, f, cand = 0, 0, 0
    kms = []
    while done < steps:
        n, _ = r.wait(f)
        cand += n
        kms.append(r.kernel_ms(f))
        done += 1
        if issued < steps:
            r.dispatch(key, f)
            key += batch
            issued += 1
        f = (f + 1) % frames
    dt = time.perf_counter() - t0
    topo = r.topology()
    r.close()
    kms.sort()
    return steps * batch / dt / 1e6, kms[len(kms) // 2], cand, topo


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--topos", default="frame,0:2,1:2,2:2,1:1,2:1,1:3,0:4")
    ap.add_argument("--s", default="8")
    ap.add_argument("--frames", default="2,4,8,16")
    ap.add_argument("--fused", default="1")
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--format", type=int, default=0)
    ap.add_argument("--pattern", default="^1Cat")
    ap.add_argument("--ci", action="store_true")
    a = ap.parse_args()
    import faulthandler
    q = os.environ.get("GPU_MAX_HW_QUEUES", "default(4)")
    for s in a.s.split(","):
        for fused in a.fused.split(","):
            for topo in a.topos.split(","):
                for fr in a.frames.split(","):
                    os.environ["VGEN_SEQ_S"] = s
                    os.environ["VGEN_FUSED_INV"] = fused
                    os.environ["VGEN_STREAMS"] = topo.replace(":", ",")
                    faulthandler.dump_traceback_later(40, exit=True)   # a hung configuration ends the sweep with a stack
                    try:
                        rate, med, cand, tp = run(a.batch, int(fr), a.steps * (1 << 20) // a.batch, a.format, a.pattern, a.ci)
                        print(f"queues={q} S={s} fused_inv={fused} streams={topo} frames={fr}: {rate:9.1f} Mkeys/s  "
                              f"bwd median {med:.3f} ms  cand={cand} oversub={int(tp['oversubscribed'])}", flush=True)
                        faulthandler.cancel_dump_traceback_later()
                    except Exception as e:   # noqa: BLE001
                        print(f"queues={q} S={s} fused_inv={fused} streams={topo} frames={fr}: FAILED {e}", flush=True)


if __name__ == "__main__":
    main()
