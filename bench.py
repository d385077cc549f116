#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X scan engine (driver contract: see README/DESIGN.md).

Metric (BASELINE.json): Mkeys/s = keys tested per second, whole job over all GPUs.
Workload at N=1 (BASELINE configs[1]): P2PKH, pattern "^1Cat", 2^20 keys per dispatch, compressed
public keys, sequential scalars k0(seed=42) + i, inputs resident on the device (the only per-dispatch
upload is the 1-2 KB of base points).  A "step" is one dispatch of the hot path over 2^20 keys;
`frames` dispatches are kept in flight the way the reference's scan loop keeps 2 (src/gpu.rs:399); the
default here is 6, one HIP stream / hardware queue each, so that one dispatch's serial root inversion
overlaps the others' full-chip stages.
For N>1 (python -m torch.distributed.run ... bench.py --gpus N) every rank drives its own GPU over
batch-striped disjoint scalar ranges — no data-path collective; torch.distributed only provides the
barriers and the max-over-ranks of the elapsed time.

Extra objects on the JSON line: "roofline" (integer-VALU bound: SURVEY.md §8(d), re-based on the
measured issue rates in profiles/r01_ubench_valu.jsonl) and, on rank 0 at N=1, "cpu_baseline" (the
CPU oracle — a port of the reference's rayon path — timed on a bounded sample on this box's cores).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One hardware queue per frame stream (the HIP default of 4 makes frames share queues and serialise);
# must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
FORMATS = {"p2pkh": 0, "p2wpkh": 1, "p2sh-p2wpkh": 2, "p2tr": 3, "p2pkh-uncompressed": 4, "ethereum": 5}

# Algorithmic work per key (lane-op equivalents), SURVEY.md §8(d) / BASELINE.md §5, with the 32x32
# multiply weight re-based from the estimate r_mul = 4 to the measured issue ratio r_mul = 2
# (v_mad_u64_u32 / v_mul_*_u32 issue at half the v_add_u32 rate on gfx950, profiles/r01_ubench_valu.jsonl).
# P2TR adds, per key, the BIP-341 tweak: one fixed-base multiplication (64 mixed additions of 8M+3S), one
# more field inversion (255S+15M) and one SHA-256 compression: ~990 field multiplications of 57 imul + ~60 iop.
W_IMUL = {"p2pkh": 518, "p2wpkh": 518, "p2sh-p2wpkh": 518, "p2pkh-uncompressed": 518, "ethereum": 518,
          "p2tr": 518 + 990 * 57}
W_IOP = {"p2pkh": 550 + 1450 + 1130 + 20, "p2wpkh": 550 + 1450 + 1130 + 20 + 500,
         "p2sh-p2wpkh": 550 + 2 * (1450 + 1130) + 20, "p2pkh-uncompressed": 550 + 2 * 1450 + 1130 + 20,
         "ethereum": 550 + 6400 + 20, "p2tr": 550 + 1450 + 20 + 500 + 990 * 60}
R_MUL = 2
# peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz full-rate integer lane-ops (MI355X_MICROARCH.md: SIMD-32,
# 2400 MHz; equals the 157.3 TFLOP/s fp32 vector peak / 2).  Measured sustained: 64.7 T/s.
PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12


def seed_key(seed, shard=0):
    d = hashlib.sha256(b"vgen-mi355x" + seed.to_bytes(8, "little") + shard.to_bytes(4, "little")).digest()
    k = int.from_bytes(d, "big") % N_ORDER
    assert k != 0
    return k


def usable_cores():
    """CPU threads this process may really use: affinity mask, capped by a cgroup-v2 quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(fmt_name, pattern, ci, seconds_target=12.0):
    """The oracle's restatement of scan_range_cpu (reference src/scanner.rs:211-330) on this host's cores."""
    from oracle import pyoracle as vo
    cores = usable_cores()
    fmt = FORMATS[fmt_name]
    start = seed_key(42, 0)
    probe = vo.scan_range(fmt, pattern, start, start + 20000 * cores - 1, count=10**9, ci=ci, threads=cores)
    rate = probe["operations"] / max(probe["elapsed_secs"], 1e-9)
    n = int(max(20000 * cores, min(rate * seconds_target, 5e7)))
    res = vo.scan_range(fmt, pattern, start, start + n - 1, count=10**9, ci=ci, threads=cores)
    # and one thread (SURVEY 8(d) asks for both), on a sample of ~3 s
    n1 = int(max(20000, min(rate / cores * 3.0, 5e6)))
    one = vo.scan_range(fmt, pattern, start, start + n1 - 1, count=10**9, ci=ci, threads=1)
    return {"value": res["operations"] / res["elapsed_secs"] / 1e6, "unit": "Mkeys/sec", "cores": cores, "kind": "port",
            "single_thread_value": one["operations"] / one["elapsed_secs"] / 1e6,
            "sample": f"oracle scan_range (full scalar mult + hash + encode + regex per key) over {res['operations']} "
                      f"consecutive keys from k0(seed=42), {cores} threads, {res['elapsed_secs']:.1f} s "
                      f"(single thread: {one['operations']} keys, {one['elapsed_secs']:.1f} s); "
                      "CPU restatement, not the reference binary: the reference publishes 0.05-0.2 Mkeys/s for its "
                      "rayon path (README.md:175)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1 << 20, help="keys per dispatch (BASELINE config: 2^20)")
    ap.add_argument("--frames", type=int, default=int(os.environ.get("VGEN_BENCH_FRAMES", "0")),
                    help="dispatches in flight per GPU (0 = 16, or 14 beside an RCCL communicator)")
    ap.add_argument("--format", default="p2pkh", choices=sorted(FORMATS))
    ap.add_argument("--pattern", default="^1Cat")
    ap.add_argument("--ci", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.frames <= 0:
        # A device serves ~22 busy streams before its hardware queues are oversubscribed and throughput collapses;
        # torch.distributed's RCCL communicator brings streams of its own, so leave it room when it is there.
        args.frames = 16 if world == 1 else 14

    import torch   # first: the process then shares torch's HIP runtime with libvgen_hip.so
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no HIP device visible; there is no CPU fallback)")
    # VGEN_BENCH_REHEARSE=1: rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share
    # devices, barrier / max over gloo).  Never the measured configuration: one rank per GPU over RCCL is.
    rehearse = world > 1 and os.environ.get("VGEN_BENCH_REHEARSE") == "1"
    if rehearse:
        per_dev = -(-world // torch.cuda.device_count())   # ranks sharing one GPU: split the frames between them,
        args.frames = max(2, args.frames // per_dev)       # more than ~20 streams per device oversubscribe its queues
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import vgen_amd as vg
    fmt = vg.AddressFormat(FORMATS[args.format])
    runner = vg.GpuRunner(batch_size=args.batch, fmt=fmt, device=local_rank, frames=args.frames)
    pat = vg.Pattern(args.pattern, args.ci, fmt)
    runner.set_filter(pat if pat.device_kind != 0 else None)
    N, F = runner.batch_size, runner.frames
    k0 = seed_key(42, 0)

    def key_of(step):   # batch striping: global batch b = step * world + rank
        return k0 + (step * world + rank) * N

    def run(first_step, n_steps, collect):
        cand = 0
        kms = []
        issued = done = 0
        frame_issue = frame_wait = 0
        while issued < min(F, n_steps):
            runner.dispatch(key_of(first_step + issued), frame_issue)
            issued += 1
            frame_issue = (frame_issue + 1) % F
        while done < n_steps:
            n, _ = runner.wait(frame_wait)
            if collect:
                kms.append(runner.kernel_ms(frame_wait))
            cand += n
            done += 1
            if issued < n_steps:
                runner.dispatch(key_of(first_step + issued), frame_wait)
                issued += 1
            frame_wait = (frame_wait + 1) % F
        return cand, kms

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: the first pass over the frames also creates their streams (milliseconds each), so the step time
    # that sizes the clock probe is taken from the rest of the warm-up only
    if args.warmup < F:   # set-up, not a step: every frame gets its stream before anything is timed
        run(0, F, False)
    w_first = min(args.warmup, F)
    run(0, w_first, False)
    tw = time.perf_counter()
    run(w_first, args.warmup - w_first, False)
    barrier()
    per_step_ms = (time.perf_counter() - tw) / max(1, args.warmup - w_first) * 1e3 if args.warmup > w_first else 0.0
    # shader-clock probe beside the timed region (one sleeping wave on its own stream), sized to end well
    # before the region does so that the closing synchronize never waits for it
    probe_ms = int(min(2000.0, 0.4 * per_step_ms * args.steps))
    if os.environ.get("VGEN_BENCH_PROBE") == "0":
        probe_ms = 0
    if probe_ms >= 2:
        runner.clock_probe_start(probe_ms)
    t0 = time.perf_counter()
    cand, kms = run(args.warmup, args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t0
    shader_mhz = runner.clock_probe_read() if probe_ms >= 2 else None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    keys = world * args.steps * N
    value = keys / elapsed / 1e6
    w_key = W_IMUL[args.format] * R_MUL + W_IOP[args.format]
    # the dominant kernel (seq_bwd_kernel) does everything except the per-lane prefix products of
    # seq_fwd_kernel (half an F_p multiplication per key: 74 imul + 60 iop per multiplication)
    w_bwd = w_key - (74 * R_MUL + 60) // 2
    avg_ms = sum(kms) / len(kms)
    # Launches of different frames execute concurrently (one launch is only 1 wave per SIMD), each taking
    # correspondingly longer, so the kernel's rate is its per-launch rate times the mean number of its launches
    # in flight (sum of launch durations / wall time, Little's law) - both factors are reported.
    per_launch = N * w_bwd / (avg_ms * 1e-3) / 1e12
    concurrency = sum(kms) * 1e-3 / elapsed
    achieved = per_launch * concurrency
    traffic = None
    try:   # HBM bytes per launch from the committed PMC passes (profiles/, FETCH_SIZE x2 per the guide)
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        traffic = pm.get(f"{args.format}:{N}", {}).get("seq_bwd_kernel_bytes_per_launch")
    except (OSError, ValueError):
        pass
    out = {
        "metric": "Mkeys/sec (keys tried per second)", "value": round(value, 2), "unit": "Mkeys/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic (sequential scalars from k0 = SHA-256('vgen-mi355x'||seed=42||shard=0) mod n)",
        "config": {"workload": f"{args.format} pattern {args.pattern!r}{' -i' if args.ci else ''}, "
                               f"{N} keys/dispatch, compressed pubkey, sequential-range mode",
                   "keys_per_dispatch": N, "frames_in_flight": F,
                   "parallelism": f"range-striped x{world}" + (" (REHEARSAL: ranks share a GPU)" if rehearse else ""),
                   "device_filter_kind": pat.device_kind, "candidates_reported": cand},
        "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": round(PEAK_TLANEOPS, 1),
                     "unit": "Tlaneop/s", "frac": round(achieved / PEAK_TLANEOPS, 4), "traffic": traffic,
                     "hbm_gb_per_s": round(traffic * args.steps / elapsed / 1e9, 1) if traffic else None,
                     "hbm_frac_of_8TBps": round(traffic * args.steps / elapsed / 8e12, 4) if traffic else None,
                     "kernel": "seq_bwd_kernel", "avg_launch_ms": round(avg_ms, 4), "frames_in_flight": F,
                     "mean_launches_in_flight": round(concurrency, 2), "achieved_per_launch": round(per_launch, 3),
                     "work_per_key": w_bwd, "work_per_key_whole_path": w_key,
                     "chip_achieved": round(value * 1e6 / world * w_key / 1e12, 3),
                     "chip_frac": round(value * 1e6 / world * w_key / 1e12 / PEAK_TLANEOPS, 4),
                     "shader_clock_mhz": round(shader_mhz) if shader_mhz else None,
                     "chip_frac_at_shader_clock": round(value * 1e6 / world * w_key / 1e12 /
                                                        (PEAK_TLANEOPS * shader_mhz / 2400.0), 4) if shader_mhz else None,
                     "note": "integer-VALU bound path (no MFMA; HBM traffic is a few % of peak); achieved = "
                             "algorithmic lane-op-equivalents of one seq_bwd_kernel launch / its average HIP-event "
                             "duration (achieved_per_launch) x the mean number of its launches in flight (sum of launch "
                             "durations / wall time): one launch is 1 wave per SIMD and the frames overlap on the device; "
                             "chip_* is the same ratio for the whole path from wall time; shader_clock_mhz = "
                             "s_memtime / s_memrealtime sampled by a probe wave during the timed region (the peak "
                             "assumes the nominal 2400 MHz, which power management does not sustain under this load)"},
    }
    if rank == 0 and world == 1:
        # time-to-first-match (the second half of BASELINE.json's metric): a `generate -c 1` style scan
        # through the scanner (vgen_scan), warm = existing context, cold = including context creation
        # (offset-table build + allocations; the HIP runtime itself is already initialised here)
        t1 = time.perf_counter()
        res = vg.scan_gpu_with_runner(args.pattern, vg.ScanConfig(format=fmt, count=1, seed=43,
                                                                  case_insensitive=args.ci), runner)
        warm = time.perf_counter() - t1
        t1 = time.perf_counter()
        r2 = vg.GpuRunner(batch_size=args.batch, fmt=fmt, device=local_rank, frames=args.frames)
        res2 = vg.scan_gpu_with_runner(args.pattern, vg.ScanConfig(format=fmt, count=1, seed=44,
                                                                   case_insensitive=args.ci), r2)
        cold = time.perf_counter() - t1
        r2.close()
        out["time_to_first_match"] = {"warm_s": round(warm, 5), "cold_s": round(cold, 5),
                                      "keys_scanned_warm": res.operations, "keys_scanned_cold": res2.operations,
                                      "found": bool(res.matches) and bool(res2.matches)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.format, args.pattern, args.ci)
    runner.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
