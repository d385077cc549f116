"""Frames in flight against rate AND resident host memory: for each frame count, in a process of its own,
P2PKH '^1Cat' at 2^20 keys per dispatch: the 20-dispatch region of bench.py's contract (median and best of 9, each after a
sustained leg so that the clocks are up), the sustained rate over SECONDS, and the process's peak resident set (VmHWM) — the HIP
runtime keeps ~175 MiB of host memory per hardware queue from its first use.
usage: python tools/frames_rss.py [frames,frames,...] [seconds]          (child: --one FRAMES SECONDS)"""
import json, statistics, subprocess, sys, time
sys.path.insert(0, ".")


def hwm_mib():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmHWM"):
            return int(ln.split()[1]) // 1024


def one(frames, seconds):
    import vgen_amd as v
    N = 1 << 20
    r = v.GpuRunner(batch_size=N, fmt=v.AddressFormat.P2pkh, frames=frames)
    r.set_filter(v.Pattern("^1Cat", False, v.AddressFormat.P2pkh))
    key = [0x3a8ae174e51b7b1117ab406c6570970f453c4376b6d381977db7c02fb5a993e0]

    def steps(n):
        issued = done = 0
        f = 0
        while issued < min(frames, n):
            r.dispatch(key[0], issued % frames); key[0] += N; issued += 1
        while done < n:
            r.wait(f); done += 1
            if issued < n:
                r.dispatch(key[0], f); key[0] += N; issued += 1
            f = (f + 1) % frames

    def sustained(sec):
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < sec:
            steps(4 * frames); n += 4 * frames
        return n * N / (time.perf_counter() - t0) / 1e6
    steps(2 * frames)
    sus = sustained(seconds)
    regions = []
    for _ in range(9):
        sustained(0.2)
        t0 = time.perf_counter(); steps(20); regions.append(20 * N / (time.perf_counter() - t0) / 1e6)
    r.close()
    print(json.dumps({"frames": frames, "sustained_mkeys": round(sus, 1), "region20_median": round(statistics.median(regions), 1),
                      "region20_best": round(max(regions), 1), "region20_worst": round(min(regions), 1), "peak_rss_mib": hwm_mib()}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(int(sys.argv[2]), float(sys.argv[3]))
    else:
        fl = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,4,6,8,10,12,16").split(",")]
        sec = sys.argv[2] if len(sys.argv) > 2 else "3"
        for rnd in range(2):
            for f in fl:
                subprocess.run([sys.executable, __file__, "--one", str(f), sec], check=False)
