(timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "endomorphism" 2>&1 | tail -5)
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
import vgen_amd as v
N, F = 1 << 20, 12
for fmt, pat, ci in ((0, "^1Cat", False), (2, "^3Cat", False), (4, "^1Cat", False), (5, "^0xdead", True), (1, "dead$", False)):
    for endo in (False, True):
        r = v.GpuRunner(batch_size=N, fmt=v.AddressFormat(fmt), frames=F, timing=False, endo=endo)
        r.set_filter(v.Pattern(pat, ci, v.AddressFormat(fmt)))
        key = 0x3a8ae174e51b7b1117ab406c6570970f453c4376b6d381977db7c02fb5a993e0
        steps = 300 if endo else 1000
        for rep in range(2):
            t0 = time.perf_counter(); tested = issued = done = 0
            for f in range(F):
                r.dispatch(key, f); key += N; issued += 1
            fw = 0
            while done < steps:
                n, t = r.wait(fw); tested += t; done += 1
                if issued < steps:
                    r.dispatch(key, fw); key += N; issued += 1
                fw = (fw + 1) % F
            dt = time.perf_counter() - t0
        r.close()
        print("fmt %d %-8s endo=%d: %8.1f Mkeys/s" % (fmt, pat, endo, tested / dt / 1e6), flush=True)
PY
