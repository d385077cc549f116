#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X scan engine (driver contract: see README/DESIGN.md).

Metric (BASELINE.json): Mkeys/s = keys tested per second, whole job over all GPUs.
Workload at N=1 (BASELINE configs[1]): P2PKH, pattern "^1Cat", 2^20 keys per dispatch, compressed public keys,
sequential scalars k0(seed=42) + i, inputs resident on the device (the only per-dispatch upload is the 1-2 KB of
base points in the kernel arguments).  A "step" is one dispatch of the hot path over 2^20 keys; `frames`
dispatches are kept in flight the way the reference's scan loop keeps 2 (src/gpu.rs:399) — 12 here, every frame on
a stream with a hardware queue of its own (three priority pools of four queues: no GPU_MAX_HW_QUEUES involved),
because one launch of the per-key kernel is only one wave per SIMD and a dispatch is a chain of dependent launches.
For N>1 every rank drives its own GPU over batch-striped disjoint scalar ranges — no data-path collective;
torch.distributed only provides the barriers and the max-over-ranks of the elapsed time.  `--gpus N` IS the number of
ranks: under a launcher (python -m torch.distributed.run ... bench.py --gpus N) WORLD_SIZE must agree with it; without
one, `python bench.py --gpus N` starts its N ranks itself, as a child process, before anything here touches HIP.

The JSON line: `value` is measured over EXACTLY --steps dispatches between two barrier+synchronize brackets (the
contract).  Because a short region is mostly pipeline fill and drain (20 steps = 2 ms), the same loop is also run
for >= 3 s of wall time: `sustained` — before the contract's region, so that the region is measured on a device at its
steady clocks (the metric is a steady-state rate; out of idle the shader clock needs milliseconds to come up).  Further objects: `roofline` (integer-VALU bound; SURVEY.md §8(d) yardstick
re-based on measured issue rates; chip-level fraction from wall time, the kernel's lone-launch rate from a
frames=1 pass timed with HIP events, and the issue-side counters VALU-busy / instructions per key from the
committed PMC pass under profiles/), `other_configs` (the other single-GPU BASELINE configurations and modes,
>= 1 s each), `time_to_first_match`, and `cpu_baseline` (the CPU oracle — a port of the reference's rayon path —
on this box's cores; rank 0, N=1 only).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
# 2400 MHz; equals the 157.3 TFLOP/s fp32 vector peak / 2).
PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
N_SIMD = 256 * 4


W_EC_IMUL, W_EC_IOP = 518, 550       # the batched affine addition's share of W (per curve point)
W_FMUL_IMUL, W_FMUL_IOP = 57, 60     # one more field multiplication


# A pattern evaluated as its whole DFA on the device needs the address string: for the Base58Check formats the 4-byte
# checksum (two more SHA-256 compressions) and the base conversion, for Bech32 the checksum polymod; the DFA walk itself.
W_FULL_IOP = {"p2pkh": 2 * 1450 + 400, "p2sh-p2wpkh": 2 * 1450 + 400, "p2pkh-uncompressed": 2 * 1450 + 400, "p2wpkh": 500 + 150,
              "ethereum": 150}


def work_per_key(fmt_name, endo=False, full=False):
    """W of one key.  With VGEN_FLAG_ENDO a curve point serves six keys: its addition is shared by six, two field
    multiplications (beta x, beta^2 x) are added per point, and every key still pays its own hashes.  None for P2TR:
    no frozen yardstick describes its per-key scalar multiplication over the wide-window table; its entry carries the
    issue-bound roofline from measured instructions per key instead (issue_roofline)."""
    if fmt_name == "p2tr":
        return None
    w = W_IMUL[fmt_name] * R_MUL + W_IOP[fmt_name] + (W_FULL_IOP.get(fmt_name, 0) if full else 0)
    if endo:
        ec = W_EC_IMUL * R_MUL + W_EC_IOP
        w = w - ec + (ec + 2 * (W_FMUL_IMUL * R_MUL + W_FMUL_IOP)) / 6.0
    return w


def chip_frac(rate, fmt_name, endo=False, full=False):
    w = work_per_key(fmt_name, endo, full)
    return None if w is None else round(rate * w / 1e12 / PEAK_TLANEOPS, 4)


def seed_key(seed, shard=0):
    d = hashlib.sha256(b"vgen-mi355x" + seed.to_bytes(8, "little") + shard.to_bytes(4, "little")).digest()
    k = int.from_bytes(d, "big") % N_ORDER
    assert k != 0
    return k


def batch_key(k0, step, world, rank, n):
    """Batch striping: global batch b = step * world + rank covers [k0 + b*n, k0 + (b+1)*n)."""
    return k0 + (step * world + rank) * n


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


def cpu_model():
    """Model name of the host CPU (core count AND model beside the CPU baseline)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def cpu_baseline(fmt_name, pattern, ci, seconds_target=12.0):
    """The oracle's restatements of scan_range_cpu (reference src/scanner.rs:211-330; `value`) and of the default random-key
    loop scan_with_progress (src/scanner.rs:118-169; `random_loop`) on this host's cores, at nproc and on one thread."""
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
    # The reference's DEFAULT CPU loop (scan_with_progress, src/scanner.rs:118-169: an independent random key per candidate,
    # 10 000 per batch) beside the range loop, at nproc and on one thread (SURVEY.md 8(d) asks for both loops).  `value`
    # stays the range loop so that the record is comparable across rounds.
    random_loop = None
    try:
        nr = int(max(10000 * cores, min(rate * 6.0, 3e7)))
        rr = vo.scan_random(fmt, pattern, 42, count=10**9, max_keys=nr, ci=ci, threads=cores)
        nr1 = int(max(10000, min(rate / cores * 3.0, 5e6)))
        r1 = vo.scan_random(fmt, pattern, 42, count=10**9, max_keys=nr1, ci=ci, threads=1)
        random_loop = {"value": rr["operations"] / rr["elapsed_secs"] / 1e6, "single_thread_value": r1["operations"] / r1["elapsed_secs"] / 1e6,
                       "unit": "Mkeys/sec", "cores": cores,
                       "sample": f"oracle scan_random (src/scanner.rs:118-169 restated: a fresh random key per candidate — here the seeded counter-based "
                                 f"stream —, full scalar mult + hash + encode + regex each, 10 000 per batch and thread): {rr['operations']} keys on {cores} "
                                 f"threads in {rr['elapsed_secs']:.1f} s, {r1['operations']} keys on one thread in {r1['elapsed_secs']:.1f} s"}
    except Exception as e:   # noqa: BLE001
        random_loop = {"value": None, "error": f"{type(e).__name__}: {e}"}
    return {"value": res["operations"] / res["elapsed_secs"] / 1e6, "unit": "Mkeys/sec", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "single_thread_value": one["operations"] / one["elapsed_secs"] / 1e6, "random_loop": random_loop,
            "sample": f"oracle scan_range (full scalar mult + hash + encode + regex per key) over {res['operations']} "
                      f"consecutive keys from k0(seed=42), {cores} threads, {res['elapsed_secs']:.1f} s "
                      f"(single thread: {one['operations']} keys, {one['elapsed_secs']:.1f} s); "
                      "CPU restatement, not the reference binary: the reference publishes 0.05-0.2 Mkeys/s for its "
                      "rayon path (README.md:175)"}


class Pipeline:
    """The benchmark's dispatch loop over one runner: keep `frames` dispatches in flight, consume in order."""

    def __init__(self, runner, k0, world=1, rank=0):
        self.r, self.k0, self.world, self.rank = runner, k0, world, rank
        self.n, self.f = runner.batch_size, runner.frames
        self.next_step = 0
        # host time of the last run_steps(host_times=True): seconds inside dispatch calls (base points + three launches + a
        # copy through ctypes) and inside wait calls (blocked on the frame's stream + the return)
        self.host_dispatch_s = self.host_wait_s = 0.0

    def run_steps(self, n_steps, collect=False, host_times=False):
        """Exactly n_steps dispatches, all completed on return.  -> (candidates, [seq_bwd ms]).  host_times: also sum the
        wall time spent inside the dispatch and the wait calls (two clock reads per call, ~0.1 us each)."""
        r, F = self.r, self.f
        cand, kms, issued, done, fi, fw = 0, [], 0, 0, 0, 0
        first = self.next_step
        clock = time.perf_counter
        td = tw = 0.0
        stamps, t_last_issue = [], None    # host_times: when each step was seen complete; when the last dispatch call returned
        while issued < min(F, n_steps):
            if host_times:
                t = clock()
            r.dispatch(batch_key(self.k0, first + issued, self.world, self.rank, self.n), fi)
            if host_times:
                t_last_issue = clock()
                td += t_last_issue - t
            issued += 1
            fi = (fi + 1) % F
        while done < n_steps:
            if host_times:
                t = clock()
            n, _ = r.wait(fw)
            if host_times:
                t_done = clock()
                tw += t_done - t
                stamps.append(t_done)
            if collect:
                kms.append(r.kernel_ms(fw))
            cand += n
            done += 1
            if issued < n_steps:
                if host_times:
                    t = clock()
                r.dispatch(batch_key(self.k0, first + issued, self.world, self.rank, self.n), fw)
                if host_times:
                    t_last_issue = clock()
                    td += t_last_issue - t
                issued += 1
            fw = (fw + 1) % F
        self.next_step = first + n_steps
        self.host_dispatch_s, self.host_wait_s = td, tw
        self.completion_stamps, self.last_issue_stamp = stamps, t_last_issue
        return cand, kms

    def run_seconds(self, seconds, min_steps=0, clock=False):
        """Dispatches until `seconds` of wall time have passed (checked every `frames` completions), then drains.
        clock: also sum the shader-clock samples of the dispatches (vgen_frame_clock) into self.clk_cycles / clk_ticks.
        -> (dispatches, elapsed seconds incl. fill and drain)"""
        r, F = self.r, self.f
        first = self.next_step
        t0 = time.perf_counter()
        issued = done = 0
        for f in range(F):
            r.dispatch(batch_key(self.k0, first + issued, self.world, self.rank, self.n), f)
            issued += 1
        fw = 0
        stop = False
        self.clk_cycles = self.clk_ticks = 0
        while done < issued:
            r.wait(fw)
            if clock:
                c, t = r.frame_clock(fw)
                if t and 500.0 < c / t * 100.0 < 3000.0:   # (a sample torn by a counter wrap is dropped)
                    self.clk_cycles += c
                    self.clk_ticks += t
            done += 1
            if not stop and fw == F - 1:
                stop = time.perf_counter() - t0 >= seconds and issued >= min_steps
            if not stop:
                r.dispatch(batch_key(self.k0, first + issued, self.world, self.rank, self.n), fw)
                issued += 1
            fw = (fw + 1) % F
        self.next_step = first + issued
        return issued, time.perf_counter() - t0


def timed_config(vg, fmt_name, pattern, ci, batch, frames, device, seconds, label, note=None, endo=False, k0=None, table_bits=0):
    """One `other_configs` entry: sustained rate of the dispatch loop for another format / pattern (keys from k0, default
    the seeded base key).  table_bits: vgen_params.table_bits of the context (0 = the default 24-bit generator table)."""
    fmt = vg.AddressFormat(FORMATS[fmt_name])
    r = vg.GpuRunner(batch_size=batch, fmt=fmt, device=device, frames=frames, timing=False, endo=endo, table_bits=table_bits)
    pat = vg.Pattern(pattern, ci, fmt)
    r.set_filter(pat if pat.device_kind != 0 else None)
    p = Pipeline(r, seed_key(42, 0) if k0 is None else k0)
    p.run_steps(2 * frames)
    n, dt = p.run_seconds(seconds)
    r.close()
    rate = n * batch * (6 if endo else 1) / dt
    out = {"config": label, "format": fmt_name, "pattern": pattern + (" -i" if ci else ""), "value": round(rate / 1e6, 1),
           "unit": "Mkeys/sec", "seconds": round(dt, 2), "dispatches": n, "device_filter_kind": pat.device_kind,
           "chip_frac": chip_frac(rate, fmt_name, endo, pat.device_kind == 4), "work_per_key": work_per_key(fmt_name, endo, pat.device_kind == 4)}
    if fmt_name == "p2tr":
        # no frozen yardstick describes the taproot path (a scalar multiplication per key over the wide-window table): its
        # roofline is the issue bound the counters name, from the measured instructions per key
        roof = issue_roofline(rate, ("seq_fwd_kernel", "seq_inv_kernel", "seq_bwd_kernel", "p2tr_finish_kernel"), "p2tr29" if table_bits == 29 else "p2tr", "seq_bwd_kernel")
        if roof:
            out["chip_frac"], out["roofline"] = roof["frac"], roof
    if note:
        out["note"] = note
    return out


def issue_roofline(rate_keys, kernels, pmc_mode, dominant):
    """Roofline of a path whose counters (profiles/pmc_keys.json, tools/pmc_keys.sh) name VALU ISSUE as its bound: achieved =
    keys per second x VALU instructions per key (summed over the path's kernels, from SQ_INSTS_VALU) against one wave64
    VALU instruction per SIMD every 4 cycles at the nominal 2.4 GHz (the issue rate of the half-rate classes that make up
    most of the multiplication code: v_mad_u64_u32, v_add3, v_alignbit).  traffic: memory-side bytes per launch of the
    dominant kernel (FETCH_SIZE x 2 + WRITE_SIZE; for its 64-byte gathers the x2 over-corrects — see the note)."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_keys.json")))[pmc_mode]
    except (OSError, ValueError, KeyError):
        return None
    ipk = sum(v["valu_instr_per_key"] for k, v in pmc.items() if any(k.startswith(n) for n in kernels))
    dom = next((v for k, v in pmc.items() if k.startswith(dominant)), {})
    peak = N_SIMD * 2.4e9 / 4.0 * 64 / 1e12
    ach = rate_keys * ipk / 1e12
    return {"bound": "valu-issue", "achieved": round(ach, 2), "peak": round(peak, 2), "unit": "T lane-instr/s", "frac": round(ach / peak, 4),
            "traffic": dom.get("fetch_bytes_x2", 0) + dom.get("write_bytes", 0) or None,
            "valu_instr_per_key": round(ipk, 1), "kernel": dominant,
            "kernel_valu_busy_alone": dom.get("valu_busy"), "kernel_simd_cycles_per_valu_instr": dom.get("simd_cycles_per_valu_instr"),
            "kernel_lone_launch_us": dom.get("lone_launch_us_under_pmc"), "kernel_l2_hit_rate": dom.get("l2_hit_rate"),
            "kernel_hbm_side_gb_per_s": dom.get("hbm_side_gb_per_s"),
            "note": "bound named by the counters: the dominant kernel alone on the chip has VALU-busy 1.00 (keys_fwd_kernel) / 0.76 (seq_bwd_kernel<P2TR>, "
                    "two waves per SIMD; ~0.88 with twelve frames overlapped) at 3.97-4.0 SIMD cycles per instruction, while its table gathers (one "
                    "64-byte sector per window from a multi-GB table, L2 hit ~10 %, 2^20 DISTINCT random scalars per launch) stay at 1.2 TB/s of "
                    "useful bytes; a table that fits the 256 MB Infinity Cache (16 bits) is slower in proportion to its extra additions "
                    "(profiles/pmc_keys.json: keys16 / keys20 / keys24) — the gather does not bind, the instruction count of the mixed additions "
                    "does.  frac is priced at the NOMINAL 2.4 GHz: under this multiplier-dense code (60 % v_mad_u64_u32) the chip holds ~2.07 GHz "
                    "(GRBM_GUI_ACTIVE / launch duration), so ~0.86 is what VALU-busy 1.0 delivers.  FETCH_SIZE x 2 is the guide's correction for "
                    "wide streaming reads; for these gathers the uncorrected figure (half) matches the algorithmic 64 B per window."}


def keys_mode_config(vg, batch, frames, device, seconds, random_stream=False, endo=False, table_bits=0):
    """Arbitrary-scalar (KEYS) mode: the 'random 256-bit scalar' reading of north_star — a full fixed-base
    multiplication per key.  random_stream: the scalars are drawn on the device from the counter-based stream
    (vgen_dispatch_random: nothing uploaded); otherwise 32 B/key are uploaded by every dispatch (vgen_dispatch_keys)."""
    import random
    fmt = vg.AddressFormat.P2pkh
    r = vg.GpuRunner(batch_size=batch, fmt=fmt, device=device, frames=frames, timing=False, endo=endo, table_bits=table_bits)
    r.set_filter(vg.Pattern("^1Cat", False, fmt))
    if random_stream:
        ctr = [0]

        def go(f):
            r.dispatch_random(42, 0, ctr[0] * batch, f)
            ctr[0] += 1
    else:
        # `batch` DISTINCT random scalars: a short block repeated would make the table gathers cache hits and flatter the rate
        # (round 2's 4 096 keys x 256: 1.44 Gkeys/s where distinct scalars give ~1.3); 32 random bytes are a valid scalar
        # except with probability 2^-128
        blob = random.Random(42).randbytes(32 * batch)

        def go(f):
            r.dispatch_keys(blob, f)
    for f in range(frames):
        go(f)
    for f in range(frames):
        r.wait(f)
    t0 = time.perf_counter()
    issued = done = fw = 0
    for f in range(frames):
        go(f)
        issued += 1
    while done < issued:
        r.wait(fw)
        done += 1
        if time.perf_counter() - t0 < seconds:
            go(fw)
            issued += 1
        fw = (fw + 1) % frames
    dt = time.perf_counter() - t0
    r.close()
    rate = issued * batch * (6 if endo else 1) / dt
    kernels = ("rnd_fill_kernel", "keys_fwd_kernel", "keys_bwd_kernel", "seq_inv_kernel") if random_stream else ("keys_fwd_kernel", "keys_bwd_kernel", "seq_inv_kernel")
    # (the counter passes are of the one-key-per-draw kernels: per multiplication, i.e. per draw, also for the six-image form)
    # (counters are per launch of 2^20 draws: the six-image form is priced per DRAW — keys per second / 6 x instructions per draw)
    roof = issue_roofline(rate / 6, kernels, "random_endo", "keys_fwd_kernel") if endo else \
        issue_roofline(rate, kernels, "keys29" if table_bits == 29 else "random" if random_stream else "keys", "keys_fwd_kernel")
    if endo:
        return {"config": "independent random draws on an endomorphism context (VGEN_FLAG_ENDO): six keys per draw — k, lambda k, lambda^2 k and their negations",
                "format": "p2pkh", "pattern": "^1Cat", "value": round(rate / 1e6, 1), "unit": "Mkeys/sec", "seconds": round(dt, 2), "dispatches": issued,
                "draws_per_sec_M": round(rate / 6e6, 1), "chip_frac": roof["frac"] if roof else None, "roofline": roof,
                "note": "vgen_dispatch_random on an endomorphism context (`vgen-hip generate --random-keys`): every scalar multiplication serves six keys "
                        "(keys_bwd_kernel<FMT, FULL, ENDO> hashes the six images); keys per second = 6 x draws per second; the draws themselves run at the "
                        "issue-bound rate of the entry above minus the five extra hash pairs per draw"}
    return {"config": ("independent random keys drawn on the device (vgen_dispatch_random), P2PKH '^1Cat'" if random_stream
                       else "arbitrary-scalar (KEYS) mode, uploaded scalars (vgen_dispatch_keys), P2PKH '^1Cat'"),
            "format": "p2pkh", "pattern": "^1Cat",
            "value": round(rate / 1e6, 1), "unit": "Mkeys/sec", "seconds": round(dt, 2), "dispatches": issued,
            "chip_frac": roof["frac"] if roof else None, "roofline": roof,
            "note": ("the reference CPU loop's shape (rng.fill per candidate, src/scanner.rs:144-152): 2^20 independent scalars per dispatch "
                     "from the counter-based stream SHA-256('vgen-mi355x-rand' || seed || stream || index), a full k*G each over the 24-bit "
                     "window table (10 mixed additions); nothing is uploaded" if random_stream else
                     "2^20 independent scalars per dispatch, a full k*G each over the 24-bit window table (10 mixed additions); includes the "
                     "32 MB host-to-device upload of every dispatch")}


def wide_table_configs(vg, batch, frames, device, seconds):
    """The scalar-multiplication paths over the 29-bit SIGNED-window generator table (8 additions per multiplication instead of the
    24-bit table's 10; 138 of the device's 288 GB; 0.7 - 2.3 s to allocate and build): what vgen_scan moves to, in the background, for
    scans it expects to run half a minute or more (scanner.cpp) — asked for here by name (vgen_params.table_bits = 29), so the context
    builds it before its first dispatch.  Rooflines from the counter passes of the same table (profiles/pmc_keys.json: keys29 / p2tr29)."""
    out = []
    t0 = time.perf_counter()
    e = keys_mode_config(vg, batch, frames, device, seconds, random_stream=True, table_bits=29)
    e["config"] += " — 29-bit signed-window table (8 additions)"
    e["seconds_incl_table_build"] = round(time.perf_counter() - t0, 2)
    e["note"] = ("as the entry over the default table, on a context created with table_bits = 29: windows of 29 bits with signed digits, a table of magnitudes "
                 "(m * 2^(29 w) * G, m <= 2^28; negative digits take (x, p - y)), 9 windows = 8 mixed additions; profiles/r04_gtab_signed.txt")
    out.append(e)
    out.append(timed_config(vg, "p2tr", "^bc1pqqq", False, batch, frames, device, seconds,
                            "P2TR (taproot tweak on the device) — 29-bit signed-window table (8 additions)", table_bits=29))
    return out


def dump_mode_configs(vg, batch, device, seconds):
    """The reference's own mode (every payload back to the host, src/gpu.rs:602-658,1030-1093): (a) the device +
    PCIe side alone, (b) a whole scan whose pattern is too permissive for the match ring, filtered on the host."""
    fmt = vg.AddressFormat.P2pkh
    frames = 4
    r = vg.GpuRunner(batch_size=batch, fmt=fmt, device=device, frames=frames, timing=False)
    r.set_filter(None)
    p = Pipeline(r, seed_key(42, 0))
    p.run_steps(2 * frames)
    n, dt = p.run_seconds(seconds)
    out = [{"config": "dump mode, device + PCIe only (20 B/key copied to pinned host memory by every dispatch)",
            "format": "p2pkh", "pattern": None, "value": round(n * batch / dt / 1e6, 1), "unit": "Mkeys/sec",
            "seconds": round(dt, 2), "dispatches": n,
            "pcie_gb_per_s": round(n * batch * 20 / dt / 1e9, 1),
            "chip_frac": round(n * batch * 20 / dt / 64e9, 4),
            "roofline": {"bound": "pcie", "achieved": round(n * batch * 20 / dt / 1e9, 1), "peak": 64.0, "unit": "GB/s",
                         "frac": round(n * batch * 20 / dt / 64e9, 4), "traffic": batch * 20,
                         "note": "PCIe 5.0 x16 device-to-host, 64 GB/s per direction nominal; the kernels of this dispatch take 0.13 ms, the copy 0.38 ms"},
            "note": "bound by the 20 B/key device-to-host copy, not by the kernels"}]
    # a pattern too permissive for the default match ring (1 key in ~23 matches): the scan grows the ring and stays on
    # the device filter; the host only encodes and confirms the candidates
    t0 = time.perf_counter()
    res = vg.scan_gpu_with_runner("^1C", vg.ScanConfig(format=fmt, count=None, seed=42, max_batches=64), r)
    dt = res.elapsed_secs   # vgen_scan's own clock (the ctypes view then spends seconds turning 3 M matches into Python objects)
    out.append({"config": "permissive prefix (1 key in ~23 matches): match ring grown by the scan, candidates confirmed on the host",
                "format": "p2pkh", "pattern": "^1C", "value": round(res.operations / dt / 1e6, 1), "unit": "Mkeys/sec",
                "seconds": round(dt, 2), "dispatches": res.operations // batch, "matches": len(res.matches),
                "host_threads": usable_cores(),
                "chip_frac": None, "roofline": host_roofline(len(res.matches) / dt, usable_cores()),
                "note": "host-bound: every candidate is encoded and confirmed on the worker pool as it arrives (address only), the matches travel as "
                        "(key, address) blocks and are rendered — WIF, hex — once, in parallel, when vgen_scan hands them over; 75.6 Mkeys/s in round 3 "
                        "(profiles/r04_permissive.txt)"})
    # a pattern nearly every address matches: full dumps, every key encoded and matched on the host (the reference's mode)
    t0 = time.perf_counter()
    res = vg.scan_gpu_with_runner("^1[1-9A-Za-z]", vg.ScanConfig(format=fmt, count=200000, seed=42, max_batches=8), r)
    dt = res.elapsed_secs
    r.close()
    out.append({"config": "host-filter scan (nearly every key matches: full dumps, every key encoded and matched on the host)",
                "format": "p2pkh", "pattern": "^1[1-9A-Za-z]", "value": round(res.operations / dt / 1e6, 2), "unit": "Mkeys/sec",
                "seconds": round(dt, 2), "dispatches": res.operations // batch, "matches": len(res.matches),
                "host_threads": usable_cores(),
                "chip_frac": None, "roofline": host_roofline(res.operations / dt, usable_cores()),
                "note": "the reference's only mode; bound by Base58Check encoding + regex on the host cores"})
    return out


def host_roofline(strings_per_s, cores):
    """Roofline of the two legs the HOST bounds (every candidate / every key encoded as Base58Check and walked through the DFA on the worker
    pool): address strings per second against cores / 0.73 us — two SHA-256 compressions of the checksum at 263 ns each on one EPYC 9575F core
    plus ~0.2 us of base conversion and DFA walk (profiles/r04_permissive.txt).  The device is idle most of the time in these legs."""
    peak = cores / 0.73e-6
    return {"bound": "host-cpu", "achieved": round(strings_per_s / 1e6, 2), "peak": round(peak / 1e6, 2), "unit": "M address strings/s",
            "frac": round(strings_per_s / peak, 4), "traffic": None, "cores": cores,
            "note": "Base58Check + DFA per string on the host's worker pool: 2 x 263 ns SHA-256 + ~0.2 us per core (profiles/r04_permissive.txt)"}


def parse_cpulist(text):
    """'0-15,64-79' -> {0..15, 64..79} (sysfs cpulist format)."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def format_cpulist(cpus):
    out, run = [], []
    for c in sorted(cpus) + [None]:
        if run and (c is None or c != run[-1] + 1):
            out.append(str(run[0]) if len(run) == 1 else f"{run[0]}-{run[-1]}")
            run = []
        if c is not None:
            run.append(c)
    return ",".join(out)


def gpu_pci_addresses(sysfs="/sys"):
    """PCI addresses of the GPUs in HIP device order, from the KFD topology — no HIP call, so it can run before torch or
    libvgen_hip.so touch the runtime.  (ROCr enumerates its GPU agents in KFD node order and HIP keeps that order;
    *_VISIBLE_DEVICES index lists are applied by the caller.)"""
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    out = []
    for node in sorted(os.listdir(base), key=int):
        props = {}
        for line in open(os.path.join(base, node, "properties")):
            k, _, v = line.strip().partition(" ")
            props[k] = v
        if int(props.get("simd_count", "0")) == 0:
            continue   # a CPU node
        loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        out.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    return out


def numa_pin(local_rank, sysfs="/sys"):
    """Pins this process to the CPU cores of the NUMA node its GPU hangs off (VERDICT r03 #1b), BEFORE torch or HIP are
    touched: the dispatch thread, the HIP runtime's own threads and the pinned host buffers they first touch then stay on
    the socket the device's PCIe root belongs to — on an 8-GPU node four ranks per socket, instead of eight dispatch loops
    wherever the scheduler puts them.  Never fatal: anything unreadable (no KFD topology, UUID-style *_VISIBLE_DEVICES, a
    single-node box) leaves the affinity alone and says why.  -> dict for the result line."""
    info = {"pinned": False, "numa_node": None, "cpus": format_cpulist(os.sched_getaffinity(0))}
    try:
        addrs = gpu_pci_addresses(sysfs)
        vis = next((os.environ[v] for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if os.environ.get(v)), None)
        if vis:
            addrs = [addrs[int(x)] for x in vis.split(",")]     # (UUIDs: ValueError -> left alone)
        if not addrs:
            raise RuntimeError("no GPU in the KFD topology")
        bdf = addrs[local_rank % len(addrs)]                    # (rehearsals put several ranks on one device)
        dev = os.path.join(sysfs, "bus/pci/devices", bdf)
        node = int(open(os.path.join(dev, "numa_node")).read())
        info.update({"numa_node": node, "pci": bdf})
        cur = os.sched_getaffinity(0)
        want = parse_cpulist(open(os.path.join(dev, "local_cpulist")).read()) & cur
        if node < 0 or not want:
            info["why_not"] = "the device reports no NUMA node" if node < 0 else "none of the node's cores is in this process's affinity mask"
        elif want == cur:
            info["why_not"] = "already confined to the node's cores"
        else:
            os.sched_setaffinity(0, want)
            info.update({"pinned": True, "cpus": format_cpulist(want)})
    except Exception as e:   # noqa: BLE001  (pinning is an optimisation, never a reason to fail)
        info["why_not"] = f"{type(e).__name__}: {e}"
    return info


def rank_report(per_rank):
    """The N > 1 line's per-rank arrays and what they say together (VERDICT r03 #1a), from the dicts the ranks gathered:
    a sub-linear point of the 1 -> 2 -> 4 -> 8 curve must name its cause from ONE run.
      value / sustained        a slow RANK (its device, its socket) shows as one low entry; a slow NODE as all low
      region_mhz / sustained_mhz   the shader clock each device held: a power / thermal cap shows here, not in the host times
      host_dispatch_us / host_wait_us  per step: the dispatch call growing with N = the host is the bottleneck (cores, memory,
                               runtime locks); wait shrinking to ~0 with a low rate = the device starves
      t0_us / t1_us            start and finish offsets from the earliest start: start_skew_us large against the region =
                               the opening barrier released the ranks unevenly (the job's elapsed time counts it)"""
    per_rank = sorted(per_rank, key=lambda r: r["rank"])
    t0 = min(r["t0"] for r in per_rank)
    cols = {}
    for r in per_rank:
        r = dict(r, t0_us=round((r["t0"] - t0) * 1e6, 1), t1_us=round((r["t1"] - t0) * 1e6, 1))
        for k, v in r.items():
            if k not in ("t0", "t1"):
                cols.setdefault(k, []).append(v)
    timing = {"start_skew_us": round((max(r["t0"] for r in per_rank) - t0) * 1e6, 1),
              "finish_skew_us": round((max(r["t1"] for r in per_rank) - min(r["t1"] for r in per_rank)) * 1e6, 1)}
    return cols, timing


def multi_leg_child(seconds, batch, frames):
    """Runs in a CHILD process of rank 0 (bench.py --multi-leg-child): the product's own multi-device path — vgen_scan_multi,
    what `vgen-hip --devices all` calls: one process, one context and one host thread per device, 12 streams each — over ALL
    visible devices for `seconds` of wall time, on a pattern nothing matches.  Prints one JSON object."""
    import ctypes
    import threading
    import vgen_amd as vg
    n = vg.device_count()
    fmt = vg.AddressFormat.P2pkh
    t0 = time.perf_counter()
    runners = [vg.GpuRunner(batch_size=batch, fmt=fmt, device=i, frames=frames, timing=False) for i in range(n)]
    t_create = time.perf_counter() - t0
    stop = ctypes.c_int32(0)
    timer = threading.Timer(seconds, lambda: setattr(stop, "value", 1))
    timer.start()
    res = vg.scan_gpu_with_runner("^1ZZZZZZZZZZZZ", vg.ScanConfig(format=fmt, count=None, seed=42), runners, stop=stop, force_multi=True)
    timer.cancel()
    for r in runners:
        r.close()
    rss_kb = 0
    try:
        rss_kb = int(next(l for l in open("/proc/self/status") if l.startswith("VmHWM")).split()[1])
    except Exception:   # noqa: BLE001
        pass
    print(json.dumps({"value": round(res.operations / res.elapsed_secs / 1e6, 1), "unit": "Mkeys/sec", "n_devices": n,
                      "failed_shards": res.failed_shards, "seconds": round(res.elapsed_secs, 3), "operations": res.operations,
                      "matches": len(res.matches), "frames_per_device": frames, "create_s": round(t_create, 3),
                      "peak_rss_gib": round(rss_kb / 2**20, 2),
                      "what": "vgen_scan_multi over all visible devices in ONE process (a context and a host thread per device; pattern "
                              "'^1ZZZZZZZZZZZZ', seed 42, stopped by the host's stop flag); includes the ramp-up of the frames' streams"}), flush=True)


def in_process_multi(seconds, batch, frames, affinity=None):
    """Rank 0, after every rank has closed its context: the leg above in a child process with a deadline — a failure, a hang
    or a crash of the in-process path costs this entry, never the headline line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                            "MASTER_ADDR", "MASTER_PORT", "VGEN_BENCH_REHEARSE") and not k.startswith("TORCHELASTIC")}
    cmd = [sys.executable, os.path.abspath(__file__), "--multi-leg-child", "--multi-leg-seconds", str(seconds), "--batch", str(batch), "--frames", str(frames)]

    # the child drives every device of the node: it gets the whole box's cores back (it widens its own mask at start — no
    # preexec_fn: running Python between fork and exec of a process full of runtime threads can deadlock)
    if affinity:
        cmd += ["--multi-leg-cpus", format_cpulist(affinity)]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=seconds + 120)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"value": None, "error": f"child exited with {p.returncode}: {p.stderr.strip()[-300:]}"}
        return json.loads(lines[-1])
    except Exception as e:   # noqa: BLE001  (TimeoutExpired included: subprocess.run has killed the child)
        return {"value": None, "error": f"{type(e).__name__}: {e}"[:400]}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run as a child process
    (rendezvous on 127.0.0.1, a free port), pass the same arguments on, relay the one JSON line of rank 0.
    -> exit code.  The child's stderr goes straight through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    for l in p.stdout.splitlines():     # anything else a library printed there: to stderr, the contract is ONE line on stdout
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if p.returncode == 0 and not lines:
        print("bench.py: the ranks exited without printing the result line", file=sys.stderr)
        return 1
    return p.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1 << 20, help="keys per dispatch (BASELINE config: 2^20)")
    ap.add_argument("--frames", type=int, default=int(os.environ.get("VGEN_BENCH_FRAMES", "12")),
                    help="dispatches in flight per GPU")
    ap.add_argument("--format", default="p2pkh", choices=sorted(FORMATS))
    ap.add_argument("--pattern", default="^1Cat")
    ap.add_argument("--ci", action="store_true")
    ap.add_argument("--endo", action="store_true", help="profiling aid: run the legs on a VGEN_FLAG_ENDO context (six keys per curve "
                                                        "point; NOT the BASELINE configuration, and said so in config.workload)")
    ap.add_argument("--sustained-seconds", type=float, default=3.0, help="wall time of the `sustained` leg (0 = skip)")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-numa-pin", action="store_true", help="leave the CPU affinity alone (default: every rank pins itself to the cores of its GPU's NUMA node)")
    ap.add_argument("--multi-leg-seconds", type=float, default=3.0, help="wall time of the in-process vgen_scan_multi leg over all visible devices "
                                                                         "(N > 1, or N = 1 on a multi-GPU box; 0 = skip)")
    ap.add_argument("--multi-leg-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--multi-leg-cpus", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.multi_leg_child:
        if args.multi_leg_cpus:
            try:
                os.sched_setaffinity(0, parse_cpulist(args.multi_leg_cpus))
            except (OSError, ValueError):
                pass
        multi_leg_child(args.multi_leg_seconds, args.batch, args.frames)
        return

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    # --gpus N is the number of ranks, one per GPU.  Under a launcher (torch.distributed.run sets WORLD_SIZE) the two must
    # agree; without one, N > 1 makes this process the launcher: it starts the N ranks as a CHILD (nothing here has
    # touched HIP or imported torch yet — never an exec of a process that has), relays rank 0's single JSON line and
    # exits with the child's code.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks (one rank per GPU: they must agree)")
    # before torch or HIP are touched: the cores of this rank's GPU's NUMA node (reported in config.cpu_affinity / per_rank)
    affinity0 = os.sched_getaffinity(0)
    pin = {"pinned": False, "why_not": "--no-numa-pin", "cpus": format_cpulist(affinity0), "numa_node": None} if args.no_numa_pin else numa_pin(local_rank)

    import torch   # first: the process then shares torch's HIP runtime with libvgen_hip.so
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no HIP device visible; there is no CPU fallback)")
    # VGEN_BENCH_REHEARSE=1: rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share
    # devices).  Never the measured configuration: one rank per GPU is.
    frames_asked = args.frames      # (a rehearsal splits the frames between the ranks of a device; the in-process leg runs alone)
    rehearse = world > 1 and os.environ.get("VGEN_BENCH_REHEARSE") == "1"
    # A launcher may also pin ONE device per rank through HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES: every rank then sees a single
    # device 0.  Accepted — and checked below, like every N > 1 run: the ranks must sit on N DIFFERENT devices.
    pinned = torch.cuda.device_count() == 1 and world > 1 and any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    if torch.cuda.device_count() < world and not rehearse and not pinned:
        sys.exit(f"bench.py: --gpus {world} needs {world} devices, {torch.cuda.device_count()} visible "
                 "(VGEN_BENCH_REHEARSE=1 lets ranks share a device for a rehearsal; its numbers are not a measurement)")
    if pinned and not rehearse:
        local_rank = 0
    if rehearse:
        per_dev = -(-world // torch.cuda.device_count())   # ranks sharing one GPU split the frames between them:
        args.frames = max(2, args.frames // per_dev)       # more than ~20 busy queues per device collapse the throughput
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    real_stdout = None
    if world > 1:
        # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."): the contract is ONE JSON line
        # there, so everything this process and its libraries print goes to stderr and the line is written to the
        # original descriptor at the end
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        # The data path has no collective (disjoint scalar ranges): the process group only carries the barriers and
        # the max of the elapsed time, so it runs over gloo — no RCCL communicator, no extra device queues beside
        # the frames' own.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (a rank that dies must fail the job within minutes, not leave the others at a barrier for gloo's default half hour;
        #  the longest legitimate wait is rank 0's in-process leg: its seconds + a 120 s deadline)
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        if not rehearse:
            # one rank per GPU, verified: N ranks on fewer than N devices would print an aggregate that is not N GPUs' worth
            pr = torch.cuda.get_device_properties(local_rank)
            ident = (os.uname().nodename, pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id, str(pr.uuid))
            idents = [None] * world
            dist.all_gather_object(idents, ident)
            if len(set(idents)) < world:
                sys.exit(f"bench.py: --gpus {world} but the ranks sit on {len(set(idents))} device(s) — one rank per GPU is the measured configuration "
                         "(VGEN_BENCH_REHEARSE=1 for a rehearsal on fewer devices)")

    # ---- the product's own multi-device path, once, over ALL visible devices in ONE process (vgen_scan_multi: what
    # `vgen-hip --devices all` runs; bench.py's ranks are N processes).  FIRST, before any rank has created a stream: a
    # device's hardware queues belong to the processes that made them until those exit, and a rank's twelve idle queues
    # beside the leg's twelve busy ones are the oversubscription that collapses throughput (measured in the rehearsal:
    # 0.54 instead of 11.9 Gkeys/s when the leg ran after the ranks had used their frames).  Rank 0 starts it as a child
    # process with a deadline, the other ranks wait at a (gloo-only) barrier; when the child has exited its queues are
    # gone and the measured run starts on a clean device.  Never part of `value`.
    multi_leg = None
    if args.multi_leg_seconds > 0 and (world > 1 or torch.cuda.device_count() > 1):
        if rank == 0:
            multi_leg = in_process_multi(args.multi_leg_seconds, args.batch, frames_asked, affinity0)
        if world > 1:
            dist.barrier()

    import vgen_amd as vg
    fmt = vg.AddressFormat(FORMATS[args.format])
    # (no per-dispatch HIP events on the measured runner: they cost the host ~5 us per dispatch, which a short timed
    # region feels; launch durations are sampled afterwards on a runner of their own)
    runner = vg.GpuRunner(batch_size=args.batch, fmt=fmt, device=local_rank, frames=args.frames, timing=False, endo=args.endo)
    pat = vg.Pattern(args.pattern, args.ci, fmt)
    runner.set_filter(pat if pat.device_kind != 0 else None)
    N, F = runner.batch_size, runner.frames
    K6 = 6 if args.endo else 1          # keys per dispatch = K6 x N
    pipe = Pipeline(runner, seed_key(42, 0), world, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # set-up, not a step: every frame gets its stream (milliseconds each, created at first use) before anything is timed
    pipe.run_steps(F)
    barrier()

    # ---- the contract's --warmup untimed steps ----
    pipe.run_steps(args.warmup)
    barrier()

    # ---- the same loop for >= 3 s of wall time.  It runs between the warm-up and the contract's region (nothing but
    # the barrier separates it from the region): BASELINE.json's metric is a
    # steady-state rate, and a device coming out of idle spends its first milliseconds below its steady shader clock
    # (tools/ramp_trace.py: ~2.18 GHz over a 2 ms burst after idle against 2.38 GHz sustained), which is all a
    # 20-step region would see ----
    sustained = None
    shader_mhz = None
    n_s = own_dt_s = 0
    if args.sustained_seconds > 0:
        barrier()
        ts = time.perf_counter()
        n_s, own_dt_s = pipe.run_seconds(args.sustained_seconds, clock=True)
        barrier()
        dt_s = max_over_ranks(time.perf_counter() - ts)
        if world > 1:   # every rank stops on its own clock: sum the dispatches
            t = torch.tensor([float(n_s)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            n_total = int(t.item())
        else:
            n_total = n_s
        if pipe.clk_ticks:   # MHz the CUs ran at, sampled by the seq_bwd launches of this leg themselves
            shader_mhz = pipe.clk_cycles / pipe.clk_ticks * 100.0
        sustained = {"value": round(n_total * N * K6 / dt_s / 1e6, 2), "unit": "Mkeys/sec", "seconds": round(dt_s, 3),
                     "dispatches": n_total, "frames_in_flight": F}

    # ---- the contract's timed region: exactly --steps dispatches between two barrier + synchronize brackets ----
    # Opening bracket: barrier + synchronize, then every rank reads the clock.  Closing bracket: the rank's own
    # synchronize, clock, then the barrier.  All ranks are processes of one node and time.perf_counter() is the
    # system-wide CLOCK_MONOTONIC, so the job's elapsed time is (latest finish over ranks) - (earliest start over ranks):
    # the wall time of the whole job, a little MORE than the maximum of the per-rank times, without the latency of the
    # gloo barrier itself (hundreds of microseconds against a 2 ms region).  At N = 1 this is the plain bracket.
    barrier()
    if world > 1:
        # The gloo barrier releases its ranks tens to hundreds of microseconds apart — against a 2 ms region that alone would read as
        # 2-10 % of "scaling loss" in elapsed = latest finish - earliest start.  So the ranks agree on a start INSTANT on the node's
        # shared monotonic clock (rank 0's "now + 3 ms", broadcast) and each spins until it: the brackets stay barrier + synchronize on
        # both sides, the start is simply tight.  A rank that learns the instant too late starts at once; per_rank.t0_us shows it.
        go = torch.tensor([time.perf_counter() + 0.003], dtype=torch.float64)
        dist.broadcast(go, src=0)
        t_go = float(go.item())
        while time.perf_counter() < t_go:
            pass
    t0 = time.perf_counter()
    cand, _ = pipe.run_steps(args.steps, host_times=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    t_barrier = time.perf_counter()
    if world > 1:
        tt = torch.tensor([-t0, t1, t1 - t0], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[1].item()) + float(tt[0].item())     # max(t1) - min(t0)
        slowest_rank_s = float(tt[2].item())
    else:
        elapsed = slowest_rank_s = t1 - t0
    keys = world * args.steps * N * K6
    value = keys / elapsed / 1e6
    # What the region is made of, from this rank's own clock reads (no extra call in the timed loop: the stamps are the ones
    # host_times already takes): fill = start -> the first step seen complete (the chain seq_fwd -> seq_inv -> seq_bwd of the
    # first dispatches, with nothing to overlap it); steady = first completion -> the last dispatch call returning (every
    # completion is answered by a new dispatch: the device stays full); drain = last issue -> the closing synchronize (the last
    # `frames` dispatches run out, the final launches below four waves per SIMD).
    region = None
    if pipe.completion_stamps and pipe.last_issue_stamp:
        st, li = pipe.completion_stamps, max(pipe.last_issue_stamp, pipe.completion_stamps[0])
        in_steady = sum(1 for x in st if st[0] < x <= li)
        region = {"fill_us": round((st[0] - t0) * 1e6, 1), "steady_us": round((li - st[0]) * 1e6, 1), "drain_us": round((t1 - li) * 1e6, 1),
                  "steps_completed_in_steady": in_steady,
                  "steady_mkeys": round(in_steady * N * K6 / (li - st[0]) / 1e6, 1) if li > st[0] and in_steady else None,
                  "drain_steps": len(st) - 1 - in_steady, "closing_sync_us": round((t1 - st[-1]) * 1e6, 1),
                  "completion_us": [round((x - t0) * 1e6, 1) for x in st[:64]],
                  "how": "host clock of this rank: fill = region start -> first step seen complete; steady = -> return of the last dispatch call; "
                         "drain = -> after the closing synchronize.  completion_us: when each step's wait returned (in order, first 64)"}
    # shader clock during the region, from the samples its last min(F, steps) seq_bwd launches left in their frames
    # (read after the region: nothing is added to the timed loop)
    region_mhz = None
    try:
        cs = [runner.frame_clock(f) for f in range(min(F, args.steps))]
        if sum(t for _, t in cs):
            region_mhz = round(sum(c for c, _ in cs) / sum(t for _, t in cs) * 100.0)
    except Exception:   # noqa: BLE001  (dump-mode runs carry no clock sample)
        pass

    # What this rank saw, gathered from all ranks for the N > 1 line (rank_report): its own rate over the region and over the
    # sustained leg, the clocks its device held, the host's time inside the dispatch / wait calls, when it started and finished.
    mine = {"rank": rank, "device": local_rank, "value": round(args.steps * N * K6 / (t1 - t0) / 1e6, 1),
            "sustained": round(n_s * N * K6 / own_dt_s / 1e6, 1) if own_dt_s else None,
            "region_mhz": region_mhz, "sustained_mhz": round(shader_mhz) if shader_mhz else None,
            "host_dispatch_us": round(pipe.host_dispatch_s / args.steps * 1e6, 2), "host_wait_us": round(pipe.host_wait_s / args.steps * 1e6, 2),
            "numa_node": pin.get("numa_node"), "cpus": pin.get("cpus"), "pinned": pin.get("pinned"), "t0": t0, "t1": t1}
    per_rank = None
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    w_key = work_per_key(args.format, args.endo)
    # the dominant kernel (seq_bwd_kernel) does everything except the per-lane prefix products of
    # seq_fwd_kernel (half an F_p multiplication per key: 74 imul + 60 iop per multiplication)
    w_bwd = None if w_key is None else w_key - (74 * R_MUL + 60) // 2 / K6
    # HIP-event durations of seq_bwd launches while the frames overlap (informational), from a short run of their own
    avg_ms = in_flight = None
    try:
        rt = vg.GpuRunner(batch_size=args.batch, fmt=fmt, device=local_rank, frames=args.frames, timing=True, endo=args.endo)
        rt.set_filter(pat if pat.device_kind != 0 else None)
        pt = Pipeline(rt, seed_key(42, 0), world, rank)
        pt.run_steps(2 * F)
        t_ov = time.perf_counter()
        _, kms = pt.run_steps(max(64, 16 * F), collect=True)
        elapsed_ov = time.perf_counter() - t_ov
        rt.close()
        avg_ms = round(sum(kms) / len(kms), 4)
        in_flight = round(sum(kms) * 1e-3 / elapsed_ov, 2)
    except Exception:   # noqa: BLE001  (informational only)
        pass
    chip = None if w_key is None else value * 1e6 / world * w_key / 1e12
    traffic = None
    pmc = {}
    try:   # issue-side counters and HBM bytes per launch from the committed PMC passes (profiles/)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json"))).get(f"{args.format}:{N}", {})
        traffic = pmc.get("seq_bwd_kernel_hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    roofline = {
        "bound": "valu", "achieved": None if chip is None else round(chip, 3), "peak": round(PEAK_TLANEOPS, 1), "unit": "Tlaneop/s",
        "frac": None if chip is None else round(chip / PEAK_TLANEOPS, 4), "traffic": traffic,
        "work_per_key": w_key, "kernel": "seq_bwd_kernel", "kernel_work_per_key": w_bwd,
        "hbm_gb_per_s": round(traffic * args.steps / elapsed / 1e9, 1) if traffic else None,
        "hbm_frac_of_8TBps": round(traffic * args.steps / elapsed / 8e12, 4) if traffic else None,
        "avg_launch_ms_overlapped": avg_ms,
        "mean_launches_in_flight": in_flight,
        "note": "integer-VALU bound path (no MFMA; HBM traffic is a few % of peak).  achieved/frac: keys per second of "
                "the timed region x the frozen algorithmic work per key W (SURVEY.md 8(d), r_mul re-based to the "
                "measured 2) against 256 CU x 4 SIMD x 32 lanes x 2.4 GHz — a chip-level figure from wall time.  "
                "Kernel-level evidence: lone_launch (frames = 1, HIP events around every seq_bwd launch: one wave per "
                "SIMD, nothing else on the chip) and issue (VALU-busy and instructions per key from the committed "
                "rocprofv3 PMC pass).  avg_launch_ms_overlapped is the HIP-event duration of a launch while ~"
                "mean_launches_in_flight of them share the chip; it is reported, not multiplied back.",
    }
    if sustained and w_key is not None:
        s_chip = sustained["value"] * 1e6 / world * w_key / 1e12
        roofline["achieved_sustained"] = round(s_chip, 3)
        roofline["frac_sustained"] = round(s_chip / PEAK_TLANEOPS, 4)
        if region_mhz:
            roofline["shader_clock_mhz_timed_region"] = region_mhz
        if shader_mhz:
            roofline["shader_clock_mhz"] = round(shader_mhz)
            roofline["frac_sustained_at_shader_clock"] = round(s_chip / (PEAK_TLANEOPS * shader_mhz / 2400.0), 4)
    if pmc:
        roofline["issue"] = {k: pmc[k] for k in ("valu_busy", "valu_instr_per_key", "valu_instr_per_key_all_kernels", "waves_per_simd",
                                                 "source", "how") if k in pmc}
        # The same from wall time, against the issue stage as the probes of round 5 measured it (tools/ubench_phase*.hip, tools/issue_model.py): a SIMD
        # fills one 4-cycle slot with the next instruction of its highest-priority ready wave and, behind it, one FULL-RATE instruction of another wave;
        # multiply-adds fill a slot alone.  A key therefore needs at least X + max(C, (C + S) / 2) slots (profiles/r05_issue_classes.json: the static
        # census of the kernel by issue class) — the kernel-level ceiling `frac_of_slot_bound` is measured against.  Until round 4 every instruction
        # of the mixed stream cost a slot of its own (3.8 cycles per instruction); with the hash blocks as runs by issue class and the priority
        # changes between them (device/hashgen.py) the steady state retires ~1.3 instructions per slot.
        ipk = pmc.get("valu_instr_per_key_all_kernels")
        rate_keys = (sustained["value"] if sustained else value) * 1e6 / world
        if ipk and not args.endo:
            mhz = shader_mhz or region_mhz or 2400.0
            slots_per_s = N_SIMD * mhz * 1e6 / 4.0
            slots_per_key = slots_per_s / (rate_keys / 64.0)
            roofline["issue"].update({"valu_lane_instr_per_s_T": round(rate_keys * ipk / 1e12, 2),
                                      "simd_cycles_per_valu_instr_steady_state": round(4.0 * slots_per_key / ipk, 3),
                                      "issue_slots_per_key_steady_state": round(slots_per_key, 1),
                                      "valu_per_issue_slot_steady_state": round(ipk / slots_per_key, 3)})
            try:
                cls = json.load(open(os.path.join(ROOT, "profiles", "r05_issue_classes.json")))
                floor = cls["issue_slots_per_key_at_least"] * ipk / cls["per_key"]["valu"]     # (+ the chain kernels' share, priced like the rest)
                roofline["issue"].update({"classes_per_key": cls["per_key"], "issue_slots_per_key_at_least": round(floor, 1),
                                          "frac_of_slot_bound": round(floor / slots_per_key, 4),
                                          "slot_model": "slots >= X + max(C, (C + S) / 2): X exclusive (v_mad_u64_u32), C half-rate (first place of a slot only), "
                                                        "S full-rate (either place); profiles/r05_issue_classes.json, tools/issue_model.py"})
            except (OSError, ValueError, KeyError):
                pass

    out = {
        "metric": "Mkeys/sec (keys tried per second)", "value": round(value, 2), "unit": "Mkeys/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic (sequential scalars from k0 = SHA-256('vgen-mi355x'||seed=42||shard=0) mod n)",
        "config": {"workload": f"{args.format} pattern {args.pattern!r}{' -i' if args.ci else ''}, "
                               + (f"{N} curve points/dispatch x 6 endomorphism / negation images (NOT the BASELINE configuration)" if args.endo
                                  else f"{N} keys/dispatch, compressed pubkey, sequential-range mode"),
                   "keys_per_dispatch": N, "frames_in_flight": F, "topology": runner.topology(),
                   "parallelism": f"range-striped x{world}" + (" (REHEARSAL: ranks share a GPU)" if rehearse else ""),
                   "device_filter_kind": pat.device_kind, "candidates_reported": cand,
                   "cpu_affinity": {k: pin.get(k) for k in ("pinned", "numa_node", "cpus", "pci", "why_not") if pin.get(k) is not None},
                   "host_us_per_step": {"dispatch": mine["host_dispatch_us"], "wait": mine["host_wait_us"]}},
        "sustained": sustained,
        "timing_region": region,
        "roofline": roofline,
    }
    if world > 1:
        cols, skew = rank_report(per_rank)
        out["per_rank"] = cols
        out["timing"] = {"elapsed_ms": round(elapsed * 1e3, 4), "slowest_rank_ms": round(slowest_rank_s * 1e3, 4),
                         "closing_barrier_ms_rank0": round((t_barrier - t1) * 1e3, 4), **skew,
                         "how": "elapsed = latest finish - earliest start over the ranks (one node, system-wide monotonic clock); "
                                "start: after the opening barrier + synchronize, at an instant the ranks agreed on (rank 0's now + 3 ms, broadcast; "
                                "each rank spins until it, so the barrier's uneven release does not count as work time); finish: after the rank's "
                                "closing synchronize; the closing barrier follows the clock"}
    if rank == 0 and world == 1:
        # the kernel alone: frames = 1, so every launch has the chip to itself (one wave per SIMD), HIP events
        # recorded on the launch's own stream immediately before and after seq_bwd_kernel
        try:
            r1 = vg.GpuRunner(batch_size=args.batch, fmt=fmt, device=local_rank, frames=1, endo=args.endo)
            r1.set_filter(pat if pat.device_kind != 0 else None)
            p1 = Pipeline(r1, seed_key(42, 0))
            p1.run_steps(8)
            _, k1 = p1.run_steps(64, collect=True)
            r1.close()
            lone = sum(k1) / len(k1)
            roofline["lone_launch"] = {"avg_launch_ms": round(lone, 4), "launches": len(k1),
                                       "achieved": None if w_bwd is None else round(N * K6 * w_bwd / (lone * 1e-3) / 1e12, 3),
                                       "frac": None if w_bwd is None else round(N * K6 * w_bwd / (lone * 1e-3) / 1e12 / PEAK_TLANEOPS, 4),
                                       "keys_per_s_equivalent": round(N * K6 / (lone * 1e-3) / 1e6, 1),
                                       "variant": "a dispatch issued while at most one other frame of its context is in flight launches seq_bwd_kernel<.., LONE> (P2PKH / P2WPKH prefilter only): hipcc's "
                                                  "schedule of the hash pair, because a wave that has its SIMD to itself pays 4 cycles for every priority change of "
                                                  "the steady-state kernel's hash blocks (that kernel alone at 2^20 keys: 0.132 ms, profiles/r05_prio_ab.txt)"}
        except Exception as e:   # noqa: BLE001  (an auxiliary leg never costs the headline line)
            roofline["lone_launch"] = {"error": f"{type(e).__name__}: {e}"}
        # time-to-first-match (the second half of BASELINE.json's metric): a `generate -c 1` style scan
        # through the scanner (vgen_scan), warm = existing context, cold = including context creation
        # (offset-table build + allocations; the HIP runtime itself is already initialised here)
        try:
            t1 = time.perf_counter()
            res = vg.scan_gpu_with_runner(args.pattern, vg.ScanConfig(format=fmt, count=1, seed=0 if args.endo else 43,
                                                                      case_insensitive=args.ci), runner)
            warm = time.perf_counter() - t1
            t1 = time.perf_counter()
            r2 = vg.GpuRunner(batch_size=args.batch, fmt=fmt, device=local_rank, frames=args.frames, endo=args.endo)
            res2 = vg.scan_gpu_with_runner(args.pattern, vg.ScanConfig(format=fmt, count=1, seed=0 if args.endo else 44,
                                                                       case_insensitive=args.ci), r2)
            cold = time.perf_counter() - t1
            r2.close()
            out["time_to_first_match"] = {"warm_s": round(warm, 5), "cold_s": round(cold, 5),
                                          "keys_scanned_warm": res.operations, "keys_scanned_cold": res2.operations,
                                          "found": bool(res.matches) and bool(res2.matches)}
        except Exception as e:   # noqa: BLE001
            out["time_to_first_match"] = {"error": f"{type(e).__name__}: {e}"}
    runner.close()
    if multi_leg is not None:
        out["in_process_multi"] = multi_leg
    if rank == 0 and world == 1 and not args.no_other_configs:
        sec = 1.0
        # every auxiliary leg stands alone: a failure there is recorded in its entry and never costs the headline line
        def leg(fn, *a, **kw):
            try:
                r_ = fn(*a, **kw)
                return r_ if isinstance(r_, list) else [r_]
            except Exception as e:   # noqa: BLE001
                return [{"config": kw.get("label") or (a[8] if len(a) > 8 else fn.__name__), "error": f"{type(e).__name__}: {e}"}]

        oc = []
        oc += leg(timed_config, vg, "p2wpkh", "dead$", False, args.batch, F, local_rank, sec, "BASELINE config 3: P2WPKH bech32 suffix")
        oc += leg(timed_config, vg, "ethereum", "^0xdead", True, args.batch, F, local_rank, sec, "BASELINE config 5 (one GPU): Ethereum, case-insensitive")
        oc += leg(timed_config, vg, "p2pkh", "^13zb1hQbWVsc2S7ZTZnP2G4undNNpdh5so$", False, args.batch, F, local_rank, sec,
                  "BASELINE config 4 (one GPU): puzzle-66 exact address", note="sequential scalars from 2^65, the start of the puzzle-66 range [2^65, 2^66 - 1]", k0=1 << 65)
        oc += leg(timed_config, vg, "p2sh-p2wpkh", "^3Cat", False, args.batch, F, local_rank, sec, "P2SH-P2WPKH prefix")
        oc += leg(timed_config, vg, "p2pkh-uncompressed", "^1Cat", False, args.batch, F, local_rank, sec, "P2PKH, uncompressed public key")
        oc += leg(timed_config, vg, "p2tr", "^bc1pqqq", False, args.batch, F, local_rank, sec, "P2TR (taproot tweak on the device)")
        oc += leg(timed_config, vg, "p2pkh", "1[Oo]ri", False, args.batch, F, local_rank, sec, "unanchored pattern: full Base58Check + DFA match on the device")
        oc += leg(timed_config, vg, "p2pkh", "^1Cat", False, args.batch, F, local_rank, sec,
                  "vanity search proper (VGEN_FLAG_ENDO): six keys per curve point — k, lambda k, lambda^2 k and their negations",
                  note="what `vgen-hip generate` runs for unseeded searches (any format but P2TR); every dispatch tests 6 x 2^20 keys for one "
                       "batch of point arithmetic; not a contiguous range, hence not the headline configuration", endo=True)
        oc += leg(timed_config, vg, "p2wpkh", "dead$", False, args.batch, F, local_rank, sec, "BASELINE config 3 as a vanity search (VGEN_FLAG_ENDO)", endo=True)
        oc += leg(timed_config, vg, "ethereum", "^0xdead", True, args.batch, F, local_rank, sec, "BASELINE config 5 (one GPU) as a vanity search (VGEN_FLAG_ENDO)", endo=True)
        oc += leg(timed_config, vg, "p2pkh", "1[Oo]ri", False, args.batch, F, local_rank, sec,
                  "unanchored pattern as a vanity search (VGEN_FLAG_ENDO): on-device Base58Check + DFA on six images per point", endo=True)
        oc += leg(keys_mode_config, vg, args.batch, min(F, 8), local_rank, sec)
        oc += leg(keys_mode_config, vg, args.batch, min(F, 8), local_rank, sec, random_stream=True)
        oc += leg(keys_mode_config, vg, args.batch, min(F, 8), local_rank, sec, random_stream=True, endo=True)
        oc += leg(wide_table_configs, vg, args.batch, min(F, 8), local_rank, sec)
        oc += leg(dump_mode_configs, vg, args.batch, local_rank, sec)
        out["other_configs"] = oc
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            os.sched_setaffinity(0, affinity0)      # the CPU baseline is the whole box's cores, not this rank's socket
            out["cpu_baseline"] = cpu_baseline(args.format, args.pattern, args.ci)
        except Exception as e:   # noqa: BLE001  (the oracle library missing or failing must not cost the line)
            out["cpu_baseline"] = {"value": None, "unit": "Mkeys/sec", "cores": usable_cores(), "cpu_model": cpu_model(), "kind": "port", "sample": None,
                                   "error": f"{type(e).__name__}: {e}"}
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        if real_stdout is not None:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
