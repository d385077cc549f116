"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
libvgen_hip.so, against the CPU oracle on identical private keys — bit-exact, as BASELINE.json asks.

Sizes: the oracle does ~50 k keys/s per core, so exhaustive comparisons use 16 Ki..128 Ki keys per
case and ONE full 2^20-key dispatch; full-size runs beyond that are checked through
size-independent properties (overlapping dispatches agree, filter-mode output equals the DFA applied
to the dump, every reported match re-derives on the oracle).
"""
import hashlib
import os
import re

import pytest

pytestmark = pytest.mark.gpu

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


@pytest.fixture(scope="module")
def vg():
    import vgen_amd
    assert vgen_amd.device_count() >= 1, "no HIP device: the gpu-marked tests need an MI355X"
    return vgen_amd


@pytest.fixture(scope="module")
def vo():
    from oracle import pyoracle
    return pyoracle


def dump(runner, start, frame=0):
    runner.set_filter(None)
    runner.dispatch(start, frame)
    blob, _, tested = runner.await_result(frame)
    assert tested == runner.batch_size
    return blob


FORMATS = [0, 1, 2, 4, 5]   # P2PKH, P2WPKH, P2SH-P2WPKH, P2PKH-uncompressed, Ethereum


@pytest.mark.parametrize("fmt", FORMATS)
def test_dump_matches_oracle_small_batches(vg, vo, fmt):
    batch = 16384
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt))
    starts = [vo.seed_key(42, 0), 1, 2, 2**65, 2**128 - 5, N - 3 * batch, 0xFF, 2**255 + 12345]
    for start in starts:
        got = dump(r, start)
        ref = vo.payload_seq(fmt, start, batch)
        assert got == ref, f"format {fmt} start {start:#x}: first mismatch at key index " \
                           f"{next(i for i in range(batch) if got[20*i:20*i+20] != ref[20*i:20*i+20])}"
    r.close()


@pytest.mark.parametrize("batch", [8192, 3 * 8192, 5 * 8192, 1 << 18])
def test_offset_table_built_on_the_device_for_any_lane_count(vg, vo, batch):
    """vgen_create builds the offset table R_u = (u S + S/2) G on the device (rtab_build_kernel: bits of u select the
    doublings of S G).  Every key of a dispatch goes through one entry, so a dump equal to the oracle's checks all of
    them — here also for lane counts that are not powers of two (the top bit of u is then not set for every prefix)."""
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh)
    for start in (vo.seed_key(5, batch & 0xFFFF), 1):
        assert dump(r, start) == vo.payload_seq(0, start, batch)
    r.close()


def test_the_largest_dispatch(vg, vo):
    """VGEN_MAX_BATCH = 2^24 keys per dispatch (sixteen times BASELINE's): the dump against the oracle at both ends and on a
    sample in between, every candidate of a filtered dispatch against that dump and against sixteen 2^20-key dispatches
    over the same keys, and one key more per dispatch refused."""
    B = 1 << 24
    with pytest.raises(vg.VgenError):
        vg.GpuRunner(batch_size=B + 8192, fmt=vg.AddressFormat.P2pkh, frames=2)
    r = vg.GpuRunner(batch_size=B, fmt=vg.AddressFormat.P2pkh, frames=2, match_cap=8192)
    start = vo.seed_key(24, 24)
    blob = bytes(dump(r, start))
    assert len(blob) == 20 * B
    assert blob[:20 * 8192] == vo.payload_seq(0, start, 8192)
    assert blob[20 * (B - 8192):] == vo.payload_seq(0, start + B - 8192, 8192)
    import random
    rng = random.Random(24)
    for i in [B // 2 - 1, B // 2, B // 2 + 1] + [rng.randrange(B) for _ in range(3000)]:
        assert blob[20 * i:20 * i + 20] == vo.payload(0, start + i), i
    pat = vg.Pattern("^1Cat", False, vg.AddressFormat.P2pkh)
    r.set_filter(pat)
    r.dispatch(start, 1)
    big, n_big, tested = r.await_result(1)
    assert tested == B and n_big == len(big) > 40 and all(blob[20 * i:20 * i + 20] == pl for i, pl in big)
    small = vg.GpuRunner(batch_size=1 << 20, fmt=vg.AddressFormat.P2pkh, frames=2)
    small.set_filter(pat)
    want = []
    for j in range(16):
        small.dispatch(start + (j << 20), 0)
        want += [((j << 20) + i, pl) for i, pl in small.await_result(0)[0]]
    assert big == want
    small.close()
    r.close()
    if os.environ.get("VGEN_TEST_FULL") != "1":
        return      # (the six-image half below: 6 x 2^24 keys and 16 more full dispatches — part of the full suite only)
    # six images per point at that size: candidate indices v * 2^24 + i stay below 2^32
    r = vg.GpuRunner(batch_size=B, fmt=vg.AddressFormat.P2pkh, frames=2, match_cap=8192, endo=True)
    small = vg.GpuRunner(batch_size=1 << 20, fmt=vg.AddressFormat.P2pkh, frames=2, endo=True)
    r.set_filter(pat)
    small.set_filter(pat)
    r.dispatch(start, 0)
    big, n_big, tested = r.await_result(0)
    assert tested == 6 * B and n_big == len(big) > 300
    want = []
    for j in range(16):
        small.dispatch(start + (j << 20), 0)
        want += [((i >> 20) * B + (j << 20) + (i & 0xFFFFF), pl) for i, pl in small.await_result(0)[0]]
    assert big == sorted(want)
    for i, pl in big[::7]:
        assert vo.payload(0, variant_key(start + (i % B), i // B)) == pl
    small.close()
    r.close()


@pytest.mark.parametrize("S", [2, 4, 16])
def test_every_keys_per_lane_setting_gives_the_same_keys(vg, vo, S, monkeypatch):
    """VGEN_SEQ_S: a lane of the sequential kernels tests 2S keys (S uniform points Q_j, each with +R_u and -R_u); the default
    is 8.  The other settings (measured in profiles/r03_s_sweep.txt) must produce the same payloads: dumps against the oracle
    for the hash160, Keccak and taproot forms, the six images of an endomorphism context, and a filtered scan."""
    monkeypatch.setenv("VGEN_SEQ_S", str(S))
    batch = 16384
    for fmt in (0, 5, 3):
        r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=2)
        for start in (vo.seed_key(S, fmt), 1, N - 3 * batch):
            assert dump(r, start) == vo.payload_seq(fmt, start, batch), (S, fmt, hex(start))
        r.close()
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=2, endo=True)
    r.set_filter(None)
    start = vo.seed_key(S, 9)
    r.dispatch(start, 0)
    blob, _, tested = r.await_result(0)
    assert tested == 6 * batch and blob[:20 * batch] == vo.payload_seq(0, start, batch)
    for v in range(1, 6):
        for i in range(0, batch, 257):
            assert blob[20 * (v * batch + i):20 * (v * batch + i) + 20] == vo.payload(0, variant_key(start + i, v)), (S, v, i)
    r.close()
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=3)
    res = vg.scan_gpu_with_runner("^1[A-C]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, start=7, end=7 + 5 * batch - 1), r)
    ref = vo.scan_range(0, "^1[A-C]", 7, 7 + 5 * batch - 1, count=10**9)
    assert [(m.address, m.wif) for m in res.matches] == [(x["address"], x["wif"]) for x in ref["matches"]]
    r.close()


def test_dump_full_size_dispatch_p2pkh(vg, vo):
    # BASELINE config 2, dispatch 0: all 2^20 hash160 byte-equal to the oracle
    batch = 1 << 20
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh)
    start = vo.seed_key(42, 0)
    got = dump(r, start)
    ref = vo.payload_seq(0, start, batch)
    assert hashlib.sha256(got).digest() == hashlib.sha256(ref).digest()
    # dispatch 255 of the same run, checked against dispatch 0's layout by an overlapping window:
    # a dispatch started half a batch later must agree on the shared half
    got2 = dump(r, start + batch // 2, frame=1)
    assert got2[: 20 * (batch // 2)] == got[20 * (batch // 2):]
    last = start + 255 * batch
    got3 = dump(r, last)
    sample = [0, 1, batch // 2 - 1, batch // 2, batch - 1] + list(range(1000, batch, 65521))
    for i in sample:
        assert got3[20 * i:20 * i + 20] == vo.payload(0, last + i)
    r.close()


@pytest.mark.parametrize("fmt", [2, 3, 4, 5])
def test_dump_full_size_dispatch_other_formats(vg, vo, fmt):
    """BASELINE's dispatch size for the formats whose payload is not the plain hash160 of the compressed key (P2WPKH shares
    P2PKH's): all 2^20 payloads of one dispatch — P2SH-P2WPKH, the taproot output key, P2PKH-uncompressed, Ethereum — byte-equal
    to the oracle's."""
    batch = 1 << 20
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=2)
    start = vo.seed_key(42, fmt)
    got = dump(r, start)
    ref = vo.payload_seq(fmt, start, batch)
    assert len(got) == len(ref) == batch * (32 if fmt == 3 else 20)
    assert hashlib.sha256(got).digest() == hashlib.sha256(ref).digest()
    r.close()


def test_both_frames_and_repeat_are_consistent(vg, vo):
    batch = 32768
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=4)
    r.set_filter(None)
    start = vo.seed_key(7, 1)
    for f in range(4):
        r.dispatch(start + f * batch, f)
    blobs = [r.await_result(f)[0] for f in range(4)]
    ref = vo.payload_seq(0, start, 4 * batch)
    assert b"".join(blobs) == ref
    r.close()


@pytest.mark.parametrize("fmt,pattern", [(0, "^1Cat"), (1, "^bc1qaa")])
def test_one_frame_contexts_run_the_variant_without_yields_and_find_the_same_keys(vg, vo, fmt, pattern, monkeypatch):
    """A dispatch issued while at most one other frame of its context is in flight launches seq_bwd_kernel<.., LONE> (hipcc's schedule
    of core/hash.h instead of the scheduled hash block, kernels.hip: payload_from_point; runtime.cpp: rt_dispatch) — whatever the number
    of frames the context was created with: dump and filter mode of both variants against the oracle."""
    batch = 1 << 16
    start = vo.seed_key(11, fmt)
    ref = vo.payload_seq(fmt, start, batch)
    p = vg.Pattern(pattern, False, vg.AddressFormat(fmt))
    assert p.device_kind != 0
    found = []
    # one frame; twelve frames driven one dispatch at a time (the twin); the same with the twin switched off (the steady-state kernel)
    for frames, twin in ((1, "1"), (12, "1"), (3, "0"), (1, "0")):
        monkeypatch.setenv("VGEN_LONE_VARIANT", twin)
        r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=frames, match_cap=65536)
        r.set_filter(None)
        r.dispatch(start, frames - 1)
        assert r.await_result(frames - 1)[0] == ref
        r.set_filter(p)
        r.dispatch(start, 0)
        recs, _, _ = r.await_result(0)
        for i, payload in recs:
            assert payload == ref[20 * i:20 * i + 20]
        found.append([i for i, payload in recs if p.matches(vg.address_from_payload(fmt, payload))])
        r.close()
    assert found[0] == found[1] == found[2] == found[3]
    assert found[0] == [i for i in range(batch) if p.matches(vg.address_from_payload(fmt, ref[20 * i:20 * i + 20]))]
    # a burst on a six-frame context: the first two dispatches find at most one other in flight (the twin), the rest run the
    # steady-state kernel beside them — every frame's dump is the oracle's
    monkeypatch.setenv("VGEN_LONE_VARIANT", "1")
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=6, match_cap=65536)
    r.set_filter(None)
    for f in range(6):
        r.dispatch(start + f * batch, f)
    for f in range(6):
        assert r.await_result(f)[0] == (ref if f == 0 else vo.payload_seq(fmt, start + f * batch, batch)), f
    r.close()


@pytest.mark.parametrize("fmt,pattern", [(0, "^1Cat"), (1, "dead$"), (2, "^3Cat"), (0, "1[Oo]ri")])
def test_split_form_point_arithmetic_and_hash_kernels_give_the_fused_kernels_results(vg, vo, fmt, pattern, monkeypatch):
    """VGEN_SPLIT=1 (an A/B switch, profiles/r05_occupancy_ab.txt): seq_bwd_kernel<.., SPLIT> parks x and the prefix byte of every key,
    seq_hash_kernel hashes 1 / 4 / 16 keys per lane — dump, prefilter and on-device DFA against the oracle."""
    batch = 1 << 16
    start = vo.seed_key(12, fmt)
    ref = vo.payload_seq(fmt, start, batch)
    p = vg.Pattern(pattern, False, vg.AddressFormat(fmt))
    want = [i for i in range(batch) if p.matches(vg.address_from_payload(fmt, ref[20 * i:20 * i + 20]))]
    monkeypatch.setenv("VGEN_SPLIT", "1")
    for kpl in ("1", "4", "16"):
        monkeypatch.setenv("VGEN_HASH_KPL", kpl)
        r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=2, match_cap=65536)
        r.set_filter(None)
        r.dispatch(start, 0)
        assert r.await_result(0)[0] == ref
        r.set_filter(p)
        r.dispatch(start, 1)
        recs, _, _ = r.await_result(1)
        for i, payload in recs:
            assert payload == ref[20 * i:20 * i + 20]
        assert [i for i, payload in recs if p.matches(vg.address_from_payload(fmt, payload))] == want
        r.close()


CASES = [
    (0, "^1Cat", False), (0, "^1[Oo]ri", False), (0, "^1cat", True), (0, "^1(Ab|Zz)", False), (0, "^11", False),
    (1, "dead$", False), (1, "^bc1qaa", False), (1, "^bc1q.*dd$", False), (1, "^bc1qq[qp]", False),
    (2, "^3Cat", False), (2, "^3[5-7]A", False),
    (4, "^1Dog", False),
    (5, "^0xdead", True), (5, "beef$", True), (5, "^0x00.*00$", False), (5, "^0xAb", False),
]


@pytest.mark.parametrize("fmt,pattern,ci", CASES, ids=lambda x: str(x))
def test_filter_mode_equals_dfa_over_dump(vg, vo, fmt, pattern, ci):
    """(ii) of SURVEY §8(d): the confirmed match set of a dispatch equals the oracle's regex applied to
    every address of that dispatch; the device prefilter may over-report but never miss."""
    batch = 1 << 17
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), match_cap=65536)
    start = vo.seed_key(42, fmt)
    blob = dump(r, start)
    oracle_re = vo.Regex(pattern, ci)
    expect = [i for i in range(batch)
              if oracle_re.matches(vo.address_from_hash160(fmt, blob[20 * i:20 * i + 20]))]
    p = vg.Pattern(pattern, ci, vg.AddressFormat(fmt))
    if p.device_kind == 0:
        pytest.skip("pattern has no device prefilter (host filtering path)")
    r.set_filter(p)
    r.dispatch(start, 0)
    recs, n_found, _ = r.await_result(0)
    assert n_found <= r.match_cap
    idx = [i for i, _ in recs]
    assert idx == sorted(idx)
    for i, payload in recs:
        assert payload == blob[20 * i:20 * i + 20]
    confirmed = [i for i, payload in recs if p.matches(vg.address_from_payload(fmt, payload))]
    assert confirmed == expect
    # the prefilter should be tight, not merely correct
    assert len(recs) <= max(4 * len(expect) + 64, 64)
    r.close()


@pytest.mark.parametrize("fmt", [0, 1, 2, 3, 4, 5])
def test_generated_patterns_on_the_device(vg, vo, fmt):
    """Patterns grown from addresses of the dispatch itself (tests/test_host_pattern_filter.py: classes, dots, alternatives,
    optional characters, gaps, either case; anchored prefixes, suffixes, both, pieces of the middle) through whichever device
    test they compile to — ranges, masks, checksum masks, the on-device automaton —: every candidate's payload is the
    dump's, and the confirmed set equals the oracle's regex over all 2^16 addresses of the dispatch."""
    import random
    from test_host_pattern_filter import generalise
    rng = random.Random(4000 + fmt)
    batch = 1 << 16
    plen = 32 if fmt == 3 else 20
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), match_cap=65536, frames=2)
    start = vo.seed_key(43, fmt)
    blob = dump(r, start)
    assert blob == vo.payload_seq(fmt, start, batch)
    addr_of = (lambda pl: vo.segwit_addr("bc", 1, pl)) if fmt == 3 else (lambda pl: vo.address_from_hash160(fmt, pl))
    addrs = [addr_of(blob[plen * i:plen * (i + 1)]) for i in range(batch)]
    head = {0: 1, 4: 1, 2: 1, 1: 4, 3: 4, 5: 2}[fmt]
    alphabet = {1: "qpzry9x8gf2tvdw0s3jn54khce6mua7l", 3: "qpzry9x8gf2tvdw0s3jn54khce6mua7l", 5: "0123456789abcdefABCDEF"}.get(
        fmt, "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz")
    kinds = {}
    import os
    for _ in range(int(os.environ.get("VGEN_PATTERN_WALK", "60" if os.environ.get("VGEN_TEST_FULL") == "1" else "36"))):
        a = rng.choice(addrs)
        pattern, ci = generalise(rng, a, head, alphabet, fmt)
        p = vg.Pattern(pattern, ci, vg.AddressFormat(fmt))
        kinds[p.device_kind] = kinds.get(p.device_kind, 0) + 1
        if p.device_kind == 0:
            continue
        ore = vo.Regex(pattern, ci)
        expect = [i for i, x in enumerate(addrs) if ore.matches(x)]
        assert addrs.index(a) in expect
        r.set_filter(p)
        r.dispatch(start, 1)
        recs, n_found, _ = r.await_result(1)
        if n_found > r.match_cap:
            continue      # nearly every key is a candidate (a one-symbol class): the scan would filter dumps on the host
        for i, payload in recs:
            assert payload == blob[plen * i:plen * (i + 1)]
        confirmed = [i for i, payload in recs if ore.matches(addr_of(payload))]
        assert confirmed == expect, (pattern, ci, p.device_kind, len(confirmed), len(expect))
    assert sum(v for k, v in kinds.items() if k != 0) >= 12, kinds
    r.close()


def test_scan_finds_first_match_like_reference_cpu_path(vg, vo):
    # BASELINE config 1/2 shape: generate -p ^1Cat -f p2pkh -c 1 with a fixed seed
    r = vg.GpuRunner(batch_size=1 << 20, fmt=vg.AddressFormat.P2pkh)
    cfg = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=1, seed=42)
    res = vg.scan_gpu_with_runner("^1Cat", cfg, r)
    assert len(res.matches) == 1 and res.operations % (1 << 20) == 0 and res.operations > 0
    m = res.matches[0]
    assert m.address.startswith("1Cat")
    g = vo.generate(0, int(m.hex, 16))
    assert (g["address"], g["wif"]) == (m.address, m.wif)
    # it is the FIRST matching key of the walk from k0(seed=42): oracle range scan over the prefix
    k0 = vo.seed_key(42, 0)
    upto = int(m.hex, 16)
    assert k0 <= upto < k0 + res.operations
    if upto - k0 <= 400000:
        ref = vo.scan_range(0, "^1Cat", k0, upto, count=1)
        assert [x["key"] for x in ref["matches"]] == [upto]
    r.close()


def test_scan_range_mode_matches_oracle_scan_range(vg, vo):
    # `vgen range --range 1:FFFF -f p2pkh` with a selective pattern: same match list as scan_range_cpu
    r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh)
    cfg = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=10**9, start=1, end=0xFFFF)
    res = vg.scan_gpu_with_runner("^1[A-C]", cfg, r)
    ref = vo.scan_range(0, "^1[A-C]", 1, 0xFFFF, count=10**9)
    assert [(m.address, m.wif, m.hex) for m in res.matches] == [(x["address"], x["wif"], x["hex"]) for x in ref["matches"]]
    assert res.operations == 8 * 8192   # whole batches are counted, even past `end` (gpu.rs:1106)
    # reference lib.rs:1597-1605: range 1:FF with the match-all default pattern must simply work
    cfg = vg.ScanConfig(format=vg.AddressFormat.Ethereum, count=3, start=1, end=0xFF)
    r2 = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.Ethereum)
    res = vg.scan_gpu_with_runner(".", cfg, r2)
    assert [m.address for m in res.matches] == [vo.generate(5, k)["address"] for k in (1, 2, 3)]
    r.close()
    r2.close()


def test_puzzle_style_exact_address_scan(vg, vo):
    # `range --puzzle`-shaped run on a small window around a known key with an exact-address pattern
    target = 2**65 + 123457
    addr = vo.generate(0, target)["address"]
    r = vg.GpuRunner(batch_size=65536, fmt=vg.AddressFormat.P2pkh)
    cfg = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=1, start=2**65, end=2**66 - 1)
    res = vg.scan_gpu_with_runner("^" + addr + "$", cfg, r)
    assert [int(m.hex, 16) for m in res.matches] == [target]
    assert res.matches[0].wif == vo.wif(target)
    r.close()


def test_errors_are_loud(vg):
    with pytest.raises(vg.VgenError):
        vg.GpuRunner(batch_size=12345)            # not a multiple of 8192
    r = vg.GpuRunner(batch_size=8192)
    with pytest.raises(vg.VgenError):
        r.dispatch(0, 0)                           # invalid scalar (SecretKey::from_slice fails)
    with pytest.raises(vg.VgenError):
        r.dispatch(N, 0)
    with pytest.raises(vg.VgenError):
        r.await_result(1)                          # "No pending operation on frame" (gpu.rs:622-625)
    with pytest.raises(vg.VgenError):
        vg.Pattern("", False)
    with pytest.raises(vg.VgenError):
        vg.Pattern("[invalid", False)
    r.close()


@pytest.mark.parametrize("fmt", FORMATS)
def test_keys_mode_arbitrary_scalars(vg, vo, fmt):
    """vgen_dispatch_keys: independent scalars, full fixed-base multiplication per key (the shape of the
    reference CPU loop, scanner.rs:151-155); invalid scalars yield nothing (address.rs:93)."""
    import random
    rng = random.Random(1000 + fmt)
    batch = 8192
    keys = [1, 2, 3, N - 1, N - 2, 0, N, N + 1, 2**256 - 1, 15, 16, 2**252, 0x1000000000000000000000000000000]
    keys += [rng.randrange(1, N) for _ in range(3000 - len(keys))]
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt))
    r.set_filter(None)
    r.dispatch_keys(keys, 0)
    blob, _, tested = r.await_result(0)
    assert tested == len(keys)
    for i, k in enumerate(keys):
        want = vo.payload(fmt, k) if vo.key_valid(k) else bytes(20)
        assert blob[20 * i:20 * i + 20] == want, (i, hex(k))
    assert blob[20 * len(keys):] == bytes(20 * (batch - len(keys)))
    # filter mode on the same keys: candidates are exactly the keys whose address matches
    pat = {0: "^1[A-D]", 1: "^bc1q[qp]", 2: "^3[A-D]", 4: "^1[A-D]", 5: "^0x[0-3]"}[fmt]
    p = vg.Pattern(pat, False, vg.AddressFormat(fmt))
    oracle_re = vo.Regex(pat, False)
    r.set_filter(p)
    r.dispatch_keys(keys, 1)
    recs, n_found, _ = r.await_result(1)
    got = [i for i, pl in recs if p.matches(vg.address_from_payload(fmt, pl))]
    want = [i for i, k in enumerate(keys) if vo.key_valid(k) and oracle_re.matches(vo.generate(fmt, k)["address"])]
    assert got == want and len(want) > 50
    r.close()


def test_sequential_batch_that_reaches_the_group_order(vg, vo):
    """Keys >= n yield no result (increment_key -> None, gpu.rs:963): the batch goes through the complete
    per-key kernel and must still agree with the oracle key by key."""
    batch = 8192
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh)
    for start in (N - 5000, N - batch, N - 1, N - batch - 3):
        got = dump(r, start)
        assert got == vo.payload_seq(0, start, batch), hex(start)
    # and a range scan ending at n-1 reports the last valid keys
    cfg = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=10**9, start=N - 20000, end=N - 1)
    res = vg.scan_gpu_with_runner("^1[1-9A-F]", cfg, r)
    ref = vo.scan_range(0, "^1[1-9A-F]", N - 20000, N - 1, count=10**9)
    assert [m.hex for m in res.matches] == [x["hex"] for x in ref["matches"]]
    r.close()


def test_multi_context_striped_scan_equals_single_scan(vg, vo):
    """vgen_scan_multi: batch striping over several contexts (here three contexts on the one GPU of the
    test box) gives the single-context result; operations are whole batches summed over shards."""
    batch = 8192
    rs = [vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=2) for _ in range(3)]
    cfg = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=10**9, start=0x10000, end=0x10000 + 20 * batch - 1)
    multi = vg.scan_gpu_with_runner("^1[A-C]", cfg, rs)
    single = vg.scan_gpu_with_runner("^1[A-C]", cfg, rs[0])
    ref = vo.scan_range(0, "^1[A-C]", 0x10000, 0x10000 + 20 * batch - 1, count=10**9)
    want = [(x["address"], x["wif"]) for x in ref["matches"]]
    assert [(m.address, m.wif) for m in single.matches] == want
    assert [(m.address, m.wif) for m in multi.matches] == want
    assert multi.operations == single.operations == 20 * batch
    # count-limited: five of the range's true matches, in ascending key order.  NOT necessarily the first five of the
    # range: the contexts stop as soon as the shared counter reaches `count`, and which batches had completed by then
    # depends on the devices' relative speed (the reference's rayon path is unordered too, src/scanner.rs:305-308).
    cfg.count = 5
    multi = vg.scan_gpu_with_runner("^1[A-C]", cfg, rs)
    assert len(multi.matches) == 5
    got5 = [(m.address, m.wif) for m in multi.matches]
    assert set(got5) <= set(want) and len(set(got5)) == 5
    assert [int(m.hex, 16) for m in multi.matches] == sorted(int(m.hex, 16) for m in multi.matches)
    for r in rs:
        r.close()


def test_cli_generate_and_range(vg, vo, tmp_path):
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(vg.library_path()), "vgen-hip")
    out = subprocess.run([exe, "generate", "-p", "^1Cat", "--seed", "42", "-o", "json", "--no-tui"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout)
    assert list(d.keys()) == ["address", "wif", "private_key_hex", "format", "pattern", "operations", "elapsed_secs", "rate"]
    assert d["address"].startswith("1Cat") and d["format"] == "P2PKH" and d["pattern"] == "^1Cat"
    g = vo.generate(0, int(d["private_key_hex"], 16))
    assert (g["address"], g["wif"]) == (d["address"], d["wif"])
    # range --range 1:FFFF, count 0 = whole range (lib.rs:524), csv writer
    out = subprocess.run([exe, "range", "--range", "1:FFFF", "-p", "^1[A-C]", "-c", "0", "-o", "csv",
                          "--gpu-batch-size", "8192", "-q"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "address,wif,private_key_hex,format,pattern,operations,elapsed_secs,rate"
    ref = vo.scan_range(0, "^1[A-C]", 1, 0xFFFF, count=10**9)
    assert [l.split(",")[:3] for l in lines[1:]] == [[x["address"], x["wif"], x["hex"]] for x in ref["matches"]]
    # puzzle range parsing + exact pattern + minimal writer
    target = 2**9 + 77
    addr = vo.generate(0, target)["address"]
    out = subprocess.run([exe, "range", "--puzzle", "10", "-p", "^" + addr + "$", "-o", "minimal",
                          "--gpu-batch-size", "8192"], capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == vo.wif(target)
    out = subprocess.run([exe, "generate", "-p", "^1Cat", "--no-gpu"], capture_output=True, text=True)
    assert out.returncode != 0 and "no CPU" in out.stderr
    # --random-keys: an independent random key per candidate, drawn on the device (the reference CPU path's shape); with a
    # seed the first matches are the oracle's scan_random walk of stream 0
    out = subprocess.run([exe, "generate", "-p", "^1Ab", "--random-keys", "--no-endo", "--seed", "42", "-c", "3", "-o", "jsonl",
                          "--gpu-batch-size", "16384", "-q"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = [json.loads(l) for l in out.stdout.strip().splitlines()]
    ref = vo.scan_random(0, "^1Ab", 42, count=3, threads=1)["matches"]
    assert [(g["address"], g["wif"]) for g in got] == [(x["address"], x["wif"]) for x in ref]
    # ... and by default six keys per draw (the candidate and its endomorphism / negation images): any keys will do for a vanity search
    out = subprocess.run([exe, "generate", "-p", "^1Ab", "--random-keys", "-c", "4", "-o", "jsonl", "--gpu-batch-size", "16384", "-q"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = [json.loads(l) for l in out.stdout.strip().splitlines()]
    assert len(got) == 4 and all(g["operations"] % (6 * 16384) == 0 for g in got)
    for g in got:
        o = vo.generate(0, int(g["private_key_hex"], 16))
        assert g["address"].startswith("1Ab") and (o["address"], o["wif"]) == (g["address"], g["wif"])
    # provider pattern: address + key range from the static table / a table file (provider.rs, lib.rs:599-631)
    out = subprocess.run([exe, "range", "-p", "boha:b1000:1", "-o", "minimal", "--gpu-batch-size", "8192"],
                         capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == vo.wif(1) and "exact match" in out.stderr
    table = tmp_path / "puzzles.csv"
    table.write_text("b1000/12,%s,p2pkh\n" % vo.generate(0, 2683)["address"])
    out = subprocess.run([exe, "range", "-p", "boha:b1000:12", "-o", "minimal", "--gpu-batch-size", "8192",
                          "--provider-table", str(table), "--checkpoint", str(tmp_path / "p12.ckpt")],
                         capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == vo.wif(2683), out.stderr
    assert "\nmatch=%064x\n" % 2683 in open(tmp_path / "p12.ckpt").read()
    # estimate (lib.rs:345-375): difficulty heuristic over the measured device rate
    out = subprocess.run([exe, "estimate", "-p", "^1Cat"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Estimated difficulty: 1 in %d\n" % 58 ** 3 in out.stdout and "Format: P2PKH\n" in out.stdout
    rate = float([l for l in out.stdout.splitlines() if l.startswith("Benchmark rate:")][0].split()[2])
    assert rate > 1e8                       # >= 100 Mkeys/s on one MI355X (BASELINE.json north_star)
    # impossible pattern: the warning of lib.rs:684-706, then an (empty) bounded range scan
    out = subprocess.run([exe, "range", "--range", "1:FFF", "-p", "^1O0", "-c", "0", "--gpu-batch-size", "8192"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "not valid in Base58 addresses: 'O0'" in out.stderr and "NEVER match" in out.stderr
    assert "No match found after" in out.stderr and out.stdout == ""


DFA_CASES = [
    (0, "Cat", False), (0, "1[Oo]ri", False), (0, "abc$", False), (0, "(?i)dead", False), (0, "[0-9]{5}$", False),
    (2, "Cat", False), (2, "xyz$", False), (4, "AAA", False),
    (1, "dead", False), (1, "q{4}", False), (5, "dead", False), (5, "[0-9]{7}", True), (5, "Ab.*Cd", False),
]


@pytest.mark.parametrize("fmt,pattern,ci", DFA_CASES, ids=lambda x: str(x))
def test_full_device_match_equals_dfa_over_dump(vg, vo, fmt, pattern, ci):
    """Patterns without a cheap prefilter are matched in full on the device (address encoded and walked
    through the DFA, core/dfa_eval.h).  Same contract as the prefilter: never misses, exact for
    Base58/Bech32, case-folded superset for Ethereum; the confirmed set equals the oracle's."""
    batch = 1 << 17
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), match_cap=65536)
    start = vo.seed_key(4242, fmt)
    blob = dump(r, start)
    oracle_re = vo.Regex(pattern, ci)
    addrs = [vo.address_from_hash160(fmt, blob[20 * i:20 * i + 20]) for i in range(batch)]
    expect = [i for i in range(batch) if oracle_re.matches(addrs[i])]
    p = vg.Pattern(pattern, ci, vg.AddressFormat(fmt))
    assert p.device_kind == 4
    r.set_filter(p)
    r.dispatch(start, 0)
    recs, n_found, _ = r.await_result(0)
    assert n_found <= r.match_cap
    confirmed = [i for i, payload in recs if p.matches(vg.address_from_payload(fmt, payload))]
    assert confirmed == expect
    if fmt != 5:
        assert [i for i, _ in recs] == expect        # exact on the device for Base58 / Bech32
    r.close()


def test_scan_with_unanchored_pattern_and_ring_overflow_fallback(vg, vo):
    # the reference's own example pattern "1[Oo]ri" (pattern.rs:304) is unanchored
    r = vg.GpuRunner(batch_size=1 << 18, fmt=vg.AddressFormat.P2pkh)
    res = vg.scan_gpu_with_runner("1[Oo]ri", vg.ScanConfig(count=3, seed=7), r)
    assert len(res.matches) == 3
    for m in res.matches:
        assert re.search("1[Oo]ri", m.address) and vo.generate(0, int(m.hex, 16))["address"] == m.address
    # a permissive full-match pattern overflows the candidate ring: the scan falls back to host filtering
    # of full dumps without losing or duplicating matches
    r2 = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, match_cap=256, frames=3)
    cfg = vg.ScanConfig(count=10**9, start=0x5000, end=0x5000 + 6 * 8192 - 1)
    res = vg.scan_gpu_with_runner("[A-Z]{2}", cfg, r2)
    ref = vo.scan_range(0, "[A-Z]{2}", 0x5000, 0x5000 + 6 * 8192 - 1, count=10**9)
    assert [m.hex for m in res.matches] == [x["hex"] for x in ref["matches"]]
    assert res.operations == 6 * 8192
    r.close()
    r2.close()


def test_a_pattern_every_key_matches_returns_its_first_matches_without_encoding_the_batch(vg, vo):
    """The reference's default `range --puzzle N` (pattern ".", count 1, src/lib.rs:519): the first key of the range is the
    match.  The scan filters full dumps on the host there; it examines the dump in index order and stops at `count` — the
    oracle's first matches, in order, and in milliseconds rather than the time it takes to encode 2^20 addresses."""
    import time
    r = vg.GpuRunner(batch_size=1 << 20, fmt=vg.AddressFormat.P2pkh)
    lo = 1 << 65
    for pattern, count in ((".", 1), ("^1", 7), ("[a-z]", 1500)):
        cfg = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=count, start=lo, end=2 * lo - 1)
        t0 = time.perf_counter()
        res = vg.scan_gpu_with_runner(pattern, cfg, r)
        dt = time.perf_counter() - t0
        ref = sorted(vo.scan_range(0, pattern, lo, lo + 4095, count=10**9)["matches"], key=lambda x: int(x["hex"], 16))[:count]
        assert [m.hex for m in res.matches] == [x["hex"] for x in ref] and len(res.matches) == count
        assert [m.address for m in res.matches] == [x["address"] for x in ref]
        assert res.operations == 1 << 20 and not res.complete
        if pattern != ".":   # (the first scan also pays for the dump buffers)
            assert dt < 0.1, dt
    # six images per point: the first matches are images of the first points, in index order
    e = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, endo=True)
    res = vg.scan_gpu_with_runner(".", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=5), e)
    assert len(res.matches) == 5
    for m in res.matches:
        assert vo.generate(0, int(m.hex, 16))["address"] == m.address
    e.close()
    r.close()


def test_scan_with_a_pattern_whose_dfa_is_not_built(vg, vo):
    """"a.{20}$" and its like (regex_dfa.h: Dfa::lazy): no device test can be derived, the scan filters full dumps on the host
    by walking the NFA — the oracle's matches over the same range, in order."""
    r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, frames=3)
    for pattern, ci in (("a.{20}$", False), ("^1.*[ab].{16}Q", True)):
        lo, hi = 0x77000, 0x77000 + 3 * 8192 - 1
        res = vg.scan_gpu_with_runner(pattern, vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, start=lo, end=hi, case_insensitive=ci), r)
        ref = sorted(vo.scan_range(0, pattern, lo, hi, count=10**9, ci=ci)["matches"], key=lambda x: int(x["hex"], 16))
        assert len(ref) > 20 and [(m.address, m.hex) for m in res.matches] == [(x["address"], x["hex"]) for x in ref]
        assert res.operations == 3 * 8192 and res.complete
    r.close()


# ---- P2TR: 32-byte payload (x-only output key), BIP-341 tweak done on the device -------------------------------


def test_p2tr_dump_matches_oracle(vg, vo):
    batch = 8192
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2tr)
    for start in (vo.seed_key(42, 3), 1, 2**65, N - 2 * batch - 7, N - 100):
        r.set_filter(None)
        r.dispatch(start, 0)
        blob, _, _ = r.await_result(0)
        assert len(blob) == 32 * batch
        assert blob == vo.payload_seq(vo.FMT_P2TR, start, batch), hex(start)
    r.close()


def test_p2tr_keys_mode_and_filters(vg, vo):
    import random
    rng = random.Random(341)
    batch = 8192
    keys = [1, 2, N - 1, 0, N, 0x0C28FCA386C7A227600B2FE50B7CAE11EC86D3BF1FBE471BE89827E19D72AA1D] + \
           [rng.randrange(1, N) for _ in range(4000)]
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2tr, match_cap=8192)
    r.set_filter(None)
    r.dispatch_keys(keys, 0)
    blob, _, _ = r.await_result(0)
    addrs = []
    for i, k in enumerate(keys):
        want = vo.payload(vo.FMT_P2TR, k) if vo.key_valid(k) else bytes(32)
        assert blob[32 * i:32 * i + 32] == want, hex(k)
        addrs.append(vo.generate(vo.FMT_P2TR, k)["address"] if vo.key_valid(k) else None)
    assert addrs[0] == "bc1pmfr3p9j00pfxjh0zmgp99y8zftmd3s5pmedqhyptwy6lm87hf5sspknck9"
    for pat, kind in [("^bc1pq", 2), ("^bc1p[qp]z", 2), ("aa$", 2), ("^bc1pq.*q$", 2), ("qqq", 4), ("[0-9]{6}", 4)]:
        p = vg.Pattern(pat, False, vg.AddressFormat.P2tr)
        assert p.device_kind == kind, (pat, p.device_kind)
        oracle_re = vo.Regex(pat, False)
        r.set_filter(p)
        r.dispatch_keys(keys, 1)
        recs, n_found, _ = r.await_result(1)
        got = [i for i, pl in recs if p.matches(vg.address_from_payload(3, pl))]
        want = [i for i, a in enumerate(addrs) if a and oracle_re.matches(a)]
        assert got == want, pat
        for i, pl in recs:
            assert pl == blob[32 * i:32 * i + 32]
        if kind == 4:
            assert [i for i, _ in recs] == want
    r.close()


def test_p2tr_scan(vg, vo):
    r = vg.GpuRunner(batch_size=65536, fmt=vg.AddressFormat.P2tr)
    res = vg.scan_gpu_with_runner("^bc1pqq", vg.ScanConfig(format=vg.AddressFormat.P2tr, count=2, seed=11), r)
    assert len(res.matches) == 2
    for m in res.matches:
        g = vo.generate(vo.FMT_P2TR, int(m.hex, 16))
        assert m.address.startswith("bc1pqq") and (g["address"], g["wif"]) == (m.address, m.wif)
    d = vg.derive(3, 1)
    assert d.address == "bc1pmfr3p9j00pfxjh0zmgp99y8zftmd3s5pmedqhyptwy6lm87hf5sspknck9"
    r.close()


# ---- resumable scans (vgen_scan_config.checkpoint_path) -----------------------------------


def test_checkpointed_range_scan_resumes_where_it_stopped(vg, vo, tmp_path):
    batch, lo, hi, pat = 8192, 1, 0x2FFFF, "^1[A-C]"
    want = vo.scan_range(0, pat, lo, hi, count=10**9)["matches"]
    assert len(want) > 50
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh)
    ck = str(tmp_path / "scan.ckpt")
    cfg = lambda **kw: vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, start=lo, end=hi, checkpoint_path=ck, **kw)
    # interrupted after 5 batches
    a = vg.scan_gpu_with_runner(pat, cfg(max_batches=5), r)
    assert a.operations == 5 * batch and not a.complete and a.resumed_operations == 0
    assert [m.hex for m in a.matches] == [x["hex"] for x in want if int(x["hex"], 16) < lo + 5 * batch]
    text = open(ck).read()
    assert text.startswith("vgen-hip checkpoint v1\n") and "\ndone=5\n" in text and "\ncomplete=0\n" in text
    assert text.count("\nmatch=") == len(a.matches)
    # a different scan must not pick the file up
    with pytest.raises(vg.VgenError) as e:
        vg.scan_gpu_with_runner("^1[A-D]", cfg(), r)
    assert "does not belong to this scan (pattern)" in str(e.value)
    with pytest.raises(vg.VgenError):
        vg.scan_gpu_with_runner(pat, vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, start=lo + 1, end=hi,
                                                   checkpoint_path=ck), r)
    # resumed: 3 more batches, then to the end of the range
    b = vg.scan_gpu_with_runner(pat, cfg(max_batches=3), r)
    assert b.resumed_operations == 5 * batch and b.operations == 3 * batch and not b.complete
    c = vg.scan_gpu_with_runner(pat, cfg(), r)
    total_batches = (hi - lo + 1 + batch - 1) // batch
    assert c.complete and c.resumed_operations == 8 * batch and c.operations == (total_batches - 8) * batch
    assert [(m.address, m.wif, m.hex) for m in c.matches] == [(x["address"], x["wif"], x["hex"]) for x in want]
    # a finished scan returns its result without touching the device again
    d = vg.scan_gpu_with_runner(pat, cfg(), r)
    assert d.complete and d.operations == 0 and [m.hex for m in d.matches] == [x["hex"] for x in want]
    # count-limited: matches recorded earlier count towards `count`
    ck2 = str(tmp_path / "count.ckpt")
    cfg2 = lambda n, **kw: vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=n, start=lo, end=hi, checkpoint_path=ck2, **kw)
    first = vg.scan_gpu_with_runner(pat, cfg2(3), r)
    more = vg.scan_gpu_with_runner(pat, cfg2(3), r)
    assert [m.hex for m in first.matches] == [x["hex"] for x in want[:3]] and more.operations == 0
    assert [m.hex for m in more.matches] == [x["hex"] for x in want[:3]]
    r.close()


def test_checkpointed_striped_and_seeded_scans(vg, vo, tmp_path):
    batch, lo, hi, pat = 8192, 1, 0x2FFFF, "^1[A-C]"
    want = [x["hex"] for x in vo.scan_range(0, pat, lo, hi, count=10**9)["matches"]]
    rs = [vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh) for _ in range(2)]
    ck = str(tmp_path / "multi.ckpt")
    cfg = lambda **kw: vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, start=lo, end=hi, checkpoint_path=ck,
                                     checkpoint_interval_ms=1, **kw)
    a = vg.scan_gpu_with_runner(pat, cfg(max_batches=4), rs)            # 4 batches per shard
    assert a.operations == 8 * batch and not a.complete and "\ndone=4 4\n" in open(ck).read()
    b = vg.scan_gpu_with_runner(pat, cfg(), rs)
    assert b.complete and b.resumed_operations == 8 * batch and [m.hex for m in b.matches] == want
    with pytest.raises(vg.VgenError):                                    # one context cannot resume a two-shard file
        vg.scan_gpu_with_runner(pat, cfg(), rs[0])
    # unseeded random scan: the resumed run adopts the base key of the file
    ck3 = str(tmp_path / "random.ckpt")
    rc = lambda **kw: vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, checkpoint_path=ck3, **kw)
    vg.scan_gpu_with_runner("^1zzzzzzzz", rc(max_batches=2), rs[0])
    base = [l for l in open(ck3).read().splitlines() if l.startswith("base=")][0]
    x = vg.scan_gpu_with_runner("^1zzzzzzzz", rc(max_batches=2), rs[0])
    assert x.resumed_operations == 2 * batch and base in open(ck3).read() and "\ndone=4\n" in open(ck3).read()
    # seeded: pinned base; another seed is a different scan
    ck4 = str(tmp_path / "seed.ckpt")
    vg.scan_gpu_with_runner("^1zzzzzzzz", vg.ScanConfig(count=None, seed=5, max_batches=1, checkpoint_path=ck4), rs[0])
    with pytest.raises(vg.VgenError):
        vg.scan_gpu_with_runner("^1zzzzzzzz", vg.ScanConfig(count=None, seed=6, max_batches=1, checkpoint_path=ck4), rs[0])
    for r in rs:
        r.close()


# ---- BASELINE-size cross-checks for every format: two independent device algorithms + sampled oracle -------------


@pytest.mark.parametrize("fmt", [0, 2, 3, 4, 5])
def test_full_size_sequential_and_arbitrary_scalar_paths_agree(vg, vo, fmt):
    """2^20 keys: the sequential path (Q_j +/- R_u with the batched inverse over three kernels) and the
    arbitrary-scalar path (fixed-window k*G per key, workgroup-shared inverse) must produce the same payloads,
    and both must equal the oracle on a sample; a dispatch shifted by half a batch must agree on the overlap."""
    batch, pb = 1 << 20, 32 if fmt == 3 else 20
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt))
    start = vo.seed_key(7, fmt)
    seq = dump(r, start)
    assert len(seq) == pb * batch
    r.set_filter(None)
    keys = b"".join((start + i).to_bytes(32, "big") for i in range(batch))
    r.dispatch_keys(keys, 1)
    arb, _, _ = r.await_result(1)
    assert hashlib.sha256(seq).digest() == hashlib.sha256(arb).digest()
    for i in [0, 1, 7, batch // 2 - 1, batch // 2, batch - 2, batch - 1] + list(range(777, batch, 99991)):
        assert seq[pb * i:pb * i + pb] == vo.payload(fmt, start + i), i
    shifted = dump(r, start + batch // 2)
    assert shifted[: pb * (batch // 2)] == seq[pb * (batch // 2):]
    r.close()


def test_first_dispatch_right_after_create_is_not_raced_by_setup(vg, vo):
    """vgen_create clears device memory on the null stream; the frames' streams do not wait for that stream, so
    the clears must be finished when vgen_create returns (a slab-wide clear once zeroed a running dispatch's
    scratch).  Fresh contexts, dispatched at once, on several frames."""
    batch = 1 << 18
    start = vo.seed_key(99, 0)
    want = hashlib.sha256(vo.payload_seq(0, start, batch)).digest()
    for i in range(6):
        r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=4)
        r.set_filter(None)
        for f in range(4):
            r.dispatch(start, f)
        for f in range(4):
            blob, _, _ = r.await_result(f)
            assert hashlib.sha256(blob).digest() == want, (i, f)
        r.close()


# ---- stop flag, progress callback, count-limited checkpoints (reference src/scanner.rs:413-440, src/gpu.rs:980-984,1106-1109) ----


def test_stop_flag_ends_a_hopeless_scan_promptly_with_whole_batches(vg):
    import ctypes
    import threading
    import time
    batch = 1 << 18
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=4)
    stop = ctypes.c_int32(0)
    seen = []

    def stopper():
        time.sleep(0.15)
        stop.value = 1

    t = threading.Thread(target=stopper)
    t0 = time.perf_counter()
    t.start()
    # 12 fixed Base58 characters: 1 in 58^12 keys — never within this test
    res = vg.scan_gpu_with_runner("^1zzzzzzzzzzzz", vg.ScanConfig(count=1, seed=9), r, progress_cb=seen.append, stop=stop)
    dt = time.perf_counter() - t0
    t.join()
    assert res.matches == [] and not res.complete
    assert res.operations > 0 and res.operations % batch == 0          # whole batches only (gpu.rs:1106)
    assert 0.1 < dt < 2.0, f"stopped scan took {dt:.2f} s"
    assert seen and seen[-1] == res.operations                          # the last callback carries the final count
    # a flag that is already set: nothing is dispatched at all (gpu.rs:980-984)
    res = vg.scan_gpu_with_runner("^1zzzzzzzzzzzz", vg.ScanConfig(count=1, seed=9), r, stop=stop)
    assert res.operations == 0 and res.matches == []
    r.close()


def test_progress_callback_once_per_batch_with_cumulative_operations(vg):
    batch = 65536
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=3)
    seen = []
    res = vg.scan_gpu_with_runner("^1zzzzzzzzzzzz", vg.ScanConfig(count=1, seed=3, max_batches=17), r, progress_cb=seen.append)
    assert seen == [batch * (i + 1) for i in range(17)] and res.operations == 17 * batch
    # two contexts, batches striped: one shared cumulative counter, still strictly increasing, one call per batch
    rs = [r, vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=3)]
    seen = []
    res = vg.scan_gpu_with_runner("^1zzzzzzzzzzzz", vg.ScanConfig(count=1, seed=3, max_batches=9), rs, progress_cb=seen.append)
    assert seen == [batch * (i + 1) for i in range(18)] and res.operations == 18 * batch
    # dump mode (host-side filtering, the reference's own mode) reports the same way
    seen = []
    res = vg.scan_gpu_with_runner("zzzzzzzz", vg.ScanConfig(format=vg.AddressFormat.P2wpkh, count=1, start=1, end=3 * 8192 - 1),
                                  vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2wpkh), progress_cb=seen.append)
    assert seen == [8192, 16384, 24576] and res.complete
    for x in rs:
        x.close()


def test_checkpoint_written_under_a_small_count_resumes_under_a_larger_one(vg, vo, tmp_path):
    import os
    import stat
    batch, lo, hi, pat = 8192, 1, 0x2FFFF, "^1[A-C]"
    want = [x["hex"] for x in vo.scan_range(0, pat, lo, hi, count=10**9)["matches"]]
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=4)
    ck = str(tmp_path / "count.ckpt")
    cfg = lambda n: vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=n, start=lo, end=hi, checkpoint_path=ck)
    a = vg.scan_gpu_with_runner(pat, cfg(1), r)
    assert [m.hex for m in a.matches] == want[:1] and not a.complete
    text = open(ck).read()
    assert "\ncomplete=0\n" in text                                    # stopping on count is not "range finished"
    assert stat.S_IMODE(os.stat(ck).st_mode) == 0o600                     # the file holds private keys
    # the committed batches keep ALL their matches: the ledger is the oracle's list up to the recorded position
    done = int([l for l in text.splitlines() if l.startswith("done=")][0][5:])
    ledger = [l[6:] for l in text.splitlines() if l.startswith("match=")]
    assert done >= 1 and ledger == [h for h in want if int(h, 16) < lo + done * batch]
    b = vg.scan_gpu_with_runner(pat, cfg(5), r)
    assert [m.hex for m in b.matches] == want[:5] and not b.complete
    c = vg.scan_gpu_with_runner(pat, cfg(None), r)
    assert c.complete and [m.hex for m in c.matches] == want
    # and across two contexts (shared counter): small count first, then everything
    ck2 = str(tmp_path / "count2.ckpt")
    rs = [r, vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=4)]
    cfg2 = lambda n: vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=n, start=lo, end=hi, checkpoint_path=ck2)
    a = vg.scan_gpu_with_runner(pat, cfg2(2), rs)
    assert len(a.matches) == 2 and not a.complete and set(m.hex for m in a.matches) <= set(want)
    c = vg.scan_gpu_with_runner(pat, cfg2(None), rs)
    assert c.complete and [m.hex for m in c.matches] == want
    for x in rs:
        x.close()


def test_frames_own_their_hardware_queues_without_environment_help(vg):
    import os
    assert "GPU_MAX_HW_QUEUES" not in os.environ or int(os.environ["GPU_MAX_HW_QUEUES"]) >= 4
    r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, frames=12)
    t = r.topology()
    assert t["streams"] == 12 and t["priority_levels"] * t["hw_queues"] >= 12 and not t["oversubscribed"]
    r.close()
    r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, frames=16)
    assert r.topology()["oversubscribed"]          # more frames than queues is allowed, and reported
    r.close()
    # contexts are created and destroyed repeatedly in one process (every context creates and destroys its own streams)
    for frames in (4, 8, 4, 12):
        r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, frames=frames)
        r.set_filter(None)
        for f in range(frames):
            r.dispatch(1 + f, f)
        for f in range(frames):
            r.await_result(f)
        r.close()


def test_permissive_pattern_grows_the_match_ring_instead_of_dumping(vg, vo):
    """A prefix 1 key in ~23 matches: far more candidates per batch than the default ring of 4096 holds.  The scan
    grows the ring (vgen_set_match_cap) and stays on the device filter; results equal the oracle's range scan, in
    order, and a pattern that overflows a ring sized from a wrong guess still loses nothing."""
    batch, lo, hi = 65536, 1, 6 * 65536
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=3, match_cap=256)
    for pat in ("^1[A-F]", "^1[2-9A-Za-z]", "1[A-D][a-z]"):
        want = vo.scan_range(0, pat, lo, hi, count=10**9)["matches"]
        assert len(want) > 2000
        res = vg.scan_gpu_with_runner(pat, vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, start=lo, end=hi), r)
        assert [(m.address, m.wif) for m in res.matches] == [(x["address"], x["wif"]) for x in want], pat
        assert res.operations == 6 * batch and res.complete
    r.close()


# (29 bits: a 138 GB table — part of the full suite: VGEN_TEST_FULL=1)
TABLE_WIDTHS = [8, 16, 20, 22, 24, 26, 25, 27] + ([29] if os.environ.get("VGEN_TEST_FULL") == "1" else [])


@pytest.mark.parametrize("bits", TABLE_WIDTHS)
def test_every_generator_table_width_gives_the_same_keys(vg, vo, bits, monkeypatch):
    """The arbitrary-scalar and taproot paths multiply through a fixed-window table of VGEN_GTAB_BITS-bit windows
    (8: the host-built 653 KB table; 16 / 20 / 22 / 24: built on the device from it, every entry as the sum of two
    entries of a table of half the width).  Every width must reproduce the oracle: explicit scalars incl. the
    extremes, one-digit scalars that hit the special entries of the two-level build in every window (low half zero,
    high half zero, all ones, group boundaries), and the P2TR tweak multiplication.  Odd widths (25 / 27 / 29) are SIGNED windows
    (core/ec.h: ec_mul_gen_signed — a table of magnitudes, negative digits take (x, p - y); 29 bits: 8 additions, 138 GB): their
    scalars also walk the sign threshold, carries rippling through all-ones windows and the top window's magnitude that stands
    for the scalar 2^256 itself."""
    import random
    rng = random.Random(bits)
    keys = [1, 2, N - 1, N - 2, 2**255, 0xFFFF, 0x10000, (1 << 200) + 5, (2**22 - 1) << 220, 2**256 - 1, 0, N] + [rng.randrange(1, N) for _ in range(500)]
    if bits % 2:
        st, half, nw = bits, 1 << (bits - 1), -(-257 // bits)
        h = (st - 1) // 2
        for w in range(nw):
            for digit in (1, 7, 8, 9, 2**h - 1, 2**h, 2**h + 1, (2**h - 1) << h, 3 << h, half - 8, half - 1, half, half + 1, half + 2**h, 2**st - 9,
                          2**st - 1, rng.randrange(1, 2**st)):
                k = digit << (w * st)
                if 0 < k < N:
                    keys.append(k)
                    keys.append((k + rng.randrange(1, N)) % N or 1)
        for w in range(1, nw):
            keys.append((1 << (st * w)) - 1)                                   # all ones below window w: the carry ripples up to it
        top = st * (nw - 1)
        keys.append((1 << 256) - (1 << top) + (half + 1) * (1 << (top - st)) + 5)     # top magnitude 2^(256 - top): the scalar 2^256, mod n
        keys.append((1 << 256) - (1 << top) + ((1 << st) - 1) * (1 << (top - st)))
    else:
        h = bits // 2
        for w in range((256 + bits - 1) // bits):
            for digit in (1, 7, 8, 9, 2**h - 1, 2**h, 2**h + 1, (2**h - 1) << h, 3 << h, 2**bits - 8, 2**bits - 1, rng.randrange(1, 2**bits)):
                k = digit << (w * bits)
                if 0 < k < N:
                    keys.append(k)
                    keys.append((k + rng.randrange(1, N)) % N or 1)      # the same digit among random ones: carries do not matter, digits are independent
    keys = [k for k in keys if k < 2**256]
    r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, table_bits=bits)      # the caller's choice: vgen_params.table_bits ...
    r.set_filter(None)
    r.dispatch_keys(keys, 0)
    blob, _, tested = r.await_result(0)
    assert tested == len(keys)
    for i, k in enumerate(keys):
        want = vo.payload(0, k) if 0 < k < N else bytes(20)
        assert blob[20 * i:20 * i + 20] == want, (bits, hex(k))
    assert r.resources()["table_bits"] == bits and r.resources()["note"] == ""      # the width asked for is the width in use
    r.close()
    monkeypatch.setenv("VGEN_GTAB_BITS", str(bits))                                  # ... and the test override, read at vgen_create
    r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2tr)
    start = vo.seed_key(bits, 3)
    assert dump(r, start) == vo.payload_seq(3, start, 8192)
    r.close()


# ---- endomorphism contexts (VGEN_FLAG_ENDO): six keys per curve point ---------------------------------------------

@pytest.mark.parametrize("fmt,pattern,ci", [(1, "cat", False), (1, "q[7x]q", False), (2, "Dog", False), (2, "ab$", False), (4, "1[Oo]r", False), (5, "dead0", True), (5, "bEEf", False)])
def test_endomorphism_with_the_whole_dfa_on_the_device(vg, vo, fmt, pattern, ci):
    """Unanchored / Base58-suffix patterns are matched on the device against the encoded address of every one of the six
    images (seq_bwd_kernel<FMT, FULL, ENDO>): the confirmed match set equals the pattern applied to the context's own
    dump, image by image."""
    batch = 1 << 16
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), endo=True, match_cap=16384)
    start = vo.seed_key(21, fmt)
    r.set_filter(None)
    r.dispatch(start, 0)
    blob, _, tested = r.await_result(0)
    assert tested == 6 * batch
    pat = vg.Pattern(pattern, ci, vg.AddressFormat(fmt))
    assert pat.device_kind == 4
    r.set_filter(pat)
    r.dispatch(start, 1)
    recs, n, tested = r.await_result(1)
    assert tested == 6 * batch and n == len(recs)
    # expected side: the ORACLE's regex over the ORACLE's encoding of every dumped payload (as the non-ENDO test above);
    # the dump itself is pinned to the oracle key by key in test_endomorphism_dump_is_the_oracle_on_all_six_images
    ore = vo.Regex(pattern, ci)
    want = [i for i in range(6 * batch) if ore.matches(vo.address_from_hash160(fmt, blob[20 * i:20 * i + 20]))]
    got = [idx for idx, pl in recs if ore.matches(vo.address_from_hash160(fmt, pl))]
    assert got == want and len(want) > 0
    for idx, pl in recs:
        assert pl == blob[20 * idx:20 * idx + 20]
    # and each reported index names a key that owns the address (variant * batch + i -> vgen_key_variant)
    for idx in got[:16]:
        k = variant_key(start + idx % batch, idx // batch)
        assert vo.generate(fmt, k)["address"] == vo.address_from_hash160(fmt, blob[20 * idx:20 * idx + 20])
    r.close()


@pytest.mark.parametrize("fmt", [5, 4])
def test_endomorphism_with_an_automaton_that_nearly_fills_the_lds(vg, vo, fmt):
    """The formats that park y beside the product tree hold 2 x 9 KiB of static LDS; with a 46.7 KB automaton the workgroup
    uses 65 136 of its 65 536 bytes.  Such a dispatch must still test all six images and report exactly the oracle's matches
    (a dispatch whose automaton left no room would test the plain keys only and say so in keys_tested: runtime.cpp)."""
    batch = 1 << 15
    pat = vg.Pattern("a.{9}b.c|d{12}", fmt == 5, vg.AddressFormat(fmt))
    assert pat.device_kind == 4 and 45000 < pat.dfa_bytes <= 64 * 1024 - 2 * 9 * 256 * 4
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), endo=True, match_cap=16384)
    start = vo.seed_key(33, fmt)
    r.set_filter(None)
    r.dispatch(start, 0)
    blob, _, tested = r.await_result(0)
    assert tested == 6 * batch
    r.set_filter(pat)
    r.dispatch(start, 1)
    recs, n, tested = r.await_result(1)
    assert tested == 6 * batch and n == len(recs)
    ore = vo.Regex("a.{9}b.c|d{12}", fmt == 5)
    want = [i for i in range(6 * batch) if ore.matches(vo.address_from_hash160(fmt, blob[20 * i:20 * i + 20]))]
    assert [idx for idx, _ in recs] == want and len(want) > 0
    for idx in want[:8]:
        assert vo.generate(fmt, variant_key(start + idx % batch, idx // batch))["address"] == vo.address_from_hash160(fmt, blob[20 * idx:20 * idx + 20])
    r.close()


LAMBDA = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72


def variant_key(k, v):
    kv = pow(LAMBDA, v % 3, N) * k % N
    return N - kv if v >= 3 else kv


@pytest.mark.parametrize("fmt", [0, 1, 2, 4, 5])
def test_endomorphism_dump_is_the_oracle_on_all_six_images(vg, vo, fmt):
    """A dispatch of an endomorphism context tests, for every base key k0 + i, the keys k, lambda k, lambda^2 k and
    their negations (mod n); entry variant * batch + i of the dump must be the oracle's payload of exactly that key."""
    batch = 8192
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), endo=True)
    r.set_filter(None)
    for start in (vo.seed_key(5, 0), 1, 2**200 + 12345):
        r.dispatch(start, 0)
        blob, _, tested = r.await_result(0)
        assert tested == 6 * batch and len(blob) == 6 * batch * 20
        for v in range(6):
            assert vg.key_variant(start + 77, v) == variant_key(start + 77, v)
            # the seeded start: EVERY entry of every image (6 x 8192 oracle keys, ~1 s); the other starts: a 1-in-97 sample
            idxs = range(batch) if start == vo.seed_key(5, 0) else list(range(0, batch, 97)) + [batch - 1]
            for i in idxs:
                want = vo.payload(fmt, variant_key(start + i, v))
                assert blob[20 * (v * batch + i):20 * (v * batch + i) + 20] == want, (hex(start), v, i)
        # images 0 (the keys themselves) in full against the sequential oracle
        assert blob[:20 * batch] == vo.payload_seq(fmt, start, batch)
    r.close()


def test_endomorphism_filter_mode_equals_dfa_over_its_own_dump(vg, vo):
    batch = 1 << 17
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, endo=True, match_cap=8192)
    start = vo.seed_key(11, 2)
    r.set_filter(None)
    r.dispatch(start, 0)
    blob, _, tested = r.await_result(0)
    assert tested == 6 * batch
    for v in range(6):       # the dump this test filters is the oracle's (sample; in full in the test above)
        for i in (0, 1, batch // 2, batch - 1, 4099 * (v + 1)):
            assert blob[20 * (v * batch + i):20 * (v * batch + i) + 20] == vo.payload(0, variant_key(start + i, v))
    for pattern in ("^1Cat", "^1[a-c]Z", "^1zz", "Cat", "zz$"):     # the last two: the whole DFA on the device, six images per point
        pat = vg.Pattern(pattern, False, vg.AddressFormat.P2pkh)
        assert pat.device_kind in ((1, 2) if pattern[0] == "^" else (4,))
        r.set_filter(pat)
        r.dispatch(start, 1)
        recs, n, tested = r.await_result(1)
        assert tested == 6 * batch and n == len(recs)
        ore = vo.Regex(pattern, False)      # expected side = oracle regex over oracle encoding, never product code
        want = [i for i in range(6 * batch) if ore.matches(vo.address_from_hash160(0, blob[20 * i:20 * i + 20]))]
        got_exact = [idx for idx, pl in recs if ore.matches(vo.address_from_hash160(0, pl))]
        assert got_exact == want and len(want) > 0, pattern
        for idx, pl in recs:
            assert pl == blob[20 * idx:20 * idx + 20]
    r.close()


def test_endomorphism_scan_returns_keys_that_really_own_their_addresses(vg, vo):
    r = vg.GpuRunner(batch_size=1 << 18, fmt=vg.AddressFormat.P2pkh, frames=4, endo=True)
    seen = []
    res = vg.scan_gpu_with_runner("^1Cat", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=8), r, progress_cb=seen.append)
    assert len(res.matches) == 8 and res.operations % (6 << 18) == 0 and seen[-1] == res.operations
    for m in res.matches:
        g = vo.generate(0, int(m.hex, 16))
        assert m.address.startswith("1Cat") and (g["address"], g["wif"]) == (m.address, m.wif)
    # some of eight random matches come from images other than the key walk itself (5 in 6 do on average)
    # a contiguous range is not what such a context tests
    for cfg in (vg.ScanConfig(count=1, seed=3), vg.ScanConfig(count=1, start=1, end=0xFFFF)):
        with pytest.raises(vg.VgenError) as e:
            vg.scan_gpu_with_runner("^1Cat", cfg, r)
        assert "contiguous key range" in str(e.value)
    # a pattern nearly every key matches goes through full dumps and host filtering: six images per point there too
    rs = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, frames=2, endo=True)
    res = vg.scan_gpu_with_runner("^1[2-9A-Za-z]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=300), rs)
    assert len(res.matches) == 300 and res.operations % (6 * 8192) == 0
    for m in res.matches:
        assert vo.generate(0, int(m.hex, 16))["address"] == m.address and re.match("^1[2-9A-Za-z]", m.address)
    assert len(set(m.hex for m in res.matches)) == 300
    rs.close()
    # two contexts (one per GPU in production): each walks from a random base of its own, one shared match counter
    r2 = vg.GpuRunner(batch_size=1 << 18, fmt=vg.AddressFormat.P2pkh, frames=4, endo=True)
    res = vg.scan_gpu_with_runner("^1Cat", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=6), [r, r2])
    assert len(res.matches) == 6 and len(set(m.hex for m in res.matches)) == 6
    for m in res.matches:
        assert vo.generate(0, int(m.hex, 16))["address"] == m.address and m.address.startswith("1Cat")
    r2.close()
    # a pattern that needs the on-device DFA tests the six images as well
    pat = vg.Pattern("Cat", False, vg.AddressFormat.P2pkh)
    assert pat.device_kind == 4
    r.set_filter(pat)
    r.dispatch(12345, 0)
    _, _, tested = r.await_result(0)
    assert tested == 6 << 18
    res = vg.scan_gpu_with_runner("CatS", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=5), r)
    assert len(res.matches) == 5 and res.operations % (6 << 18) == 0
    for m in res.matches:
        assert "CatS" in m.address and vo.generate(0, int(m.hex, 16))["address"] == m.address
    r.close()


# ---- independent random keys drawn on the device (vgen_dispatch_random) ------------------------------------------------

def test_random_dispatch_tests_exactly_the_oracles_candidates(vg, vo):
    """Lane i of vgen_dispatch_random(seed, stream, first_index) must test candidate first_index + i of the oracle's
    counter-based stream — the device draws its scalars itself (one SHA-256 per lane), nothing is uploaded.  3 000 lanes
    and the edge lanes of several (seed, stream, first_index) corners against the oracle's payload of the oracle's key."""
    batch = 8192
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=2)
    r.set_filter(None)
    corners = [(42, 0, 0, range(0, 3000)), (42, 0, batch, [0, 1, batch - 1]), (2**64 - 1, 2**32 - 1, 2**64 - batch, [0, 1, 4095, batch - 2, batch - 1]),
               (0, 0, 2**32 - 5, [0, 4, 5, 6, batch - 1]), (7, 3, 2**40 + 5, [0, 63, 64, 255, 256, batch - 1]),
               # the 24-byte seeds real searches use (vgen_dispatch_random_seed): every seed word must reach the hash
               (bytes(range(1, 25)), 2, 12345, range(0, 1000)), (b"\xff" * 24, 2**32 - 1, 2**64 - batch, [0, 1, batch - 1]),
               (bytes(20) + b"\x01\x02\x03\x04", 0, 0, [0, 1, 2, batch - 1]), (hashlib.sha256(b"seed").digest()[:24], 9, 2**33, range(0, 500))]
    for seed, stream, first, lanes in corners:
        r.dispatch_random(seed, stream, first, 0)
        blob, _, tested = r.await_result(0)
        assert tested == batch and len(blob) == 20 * batch
        for i in lanes:
            k = vo.random_key(seed, stream, first + i)
            want = vo.payload(0, k) if 0 < k < N else bytes(20)
            assert blob[20 * i:20 * i + 20] == want, (seed, stream, first, i)
            assert vg.random_key(seed, stream, first + i) == (k if 0 < k < N else None)
    with pytest.raises(vg.VgenError):
        r.dispatch_random(1, 0, 2**64 - batch + 1, 1)      # the index range would wrap
    # the other formats run through the same kernels (keys_bwd_kernel<FMT>)
    for fmt in (1, 2, 3, 4, 5):
        rf = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=1)
        rf.set_filter(None)
        rf.dispatch_random(99, fmt, 123456789, 0)
        blob, _, _ = rf.await_result(0)
        pl = 32 if fmt == 3 else 20
        for i in list(range(0, 200)) + [batch - 1]:
            assert blob[pl * i:pl * i + pl] == vo.payload(fmt, vo.random_key(99, fmt, 123456789 + i)), (fmt, i)
        rf.close()
    # filter mode reports exactly the candidates the oracle's regex accepts
    r.dispatch_random(5, 1, 0, 0)
    blob, _, _ = r.await_result(0)
    pat = vg.Pattern("^1[A-H]", False, vg.AddressFormat.P2pkh)
    r.set_filter(pat)
    r.dispatch_random(5, 1, 0, 1)
    recs, n, tested = r.await_result(1)
    ore = vo.Regex("^1[A-H]", False)
    want = [i for i in range(batch) if ore.matches(vo.address_from_hash160(0, blob[20 * i:20 * i + 20]))]
    assert [i for i, pl in recs if ore.matches(vo.address_from_hash160(0, pl))] == want and len(want) > 50
    r.close()


def test_random_key_scan_finds_what_the_oracles_random_walk_finds(vg, vo):
    """VGEN_SCAN_RANDOM_KEYS is the reference CPU path's shape (an independent random key per candidate,
    src/scanner.rs:118-169) on the device.  With a fixed seed the scan must return the matches of the oracle's
    scan_random walk of stream 0 — same keys, same addresses and WIFs, same order —, operations counted per batch."""
    r = vg.GpuRunner(batch_size=16384, fmt=vg.AddressFormat.P2pkh, frames=3)
    ref = vo.scan_random(0, "^1Ab", 42, count=6, threads=1)
    res = vg.scan_gpu_with_runner("^1Ab", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=6, seed=42, random_keys=True), r)
    assert [(m.address, m.wif, int(m.hex, 16)) for m in res.matches] == [(x["address"], x["wif"], x["key"]) for x in ref["matches"]]
    assert res.operations % 16384 == 0 and res.operations > 0
    # every match is a candidate of the stream (and none of them a neighbour of another: the keys are independent)
    keys = sorted(int(m.hex, 16) for m in res.matches)
    assert all(b - a > 2**128 for a, b in zip(keys, keys[1:]))
    # shards own streams of their own: shard 1 of 2 finds the oracle's stream-1 candidates
    res1 = vg.scan_gpu_with_runner("^1Ab", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=2, seed=42, random_keys=True, shard=1, n_shards=2), r)
    found = [int(m.hex, 16) for m in res1.matches]
    idx = [i for i in range(res1.operations) if vo.random_key(42, 1, i) in found][:2] if res1.operations <= 200000 else None
    if idx is not None:
        assert [vo.random_key(42, 1, i) for i in idx] == found
    for m in res1.matches:
        assert vo.generate(0, int(m.hex, 16))["address"] == m.address and m.address.startswith("1Ab")
    # unseeded: OS entropy, matches still re-derive on the oracle
    res2 = vg.scan_gpu_with_runner("^1A", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=3, random_keys=True), r)
    assert len(res2.matches) == 3
    for m in res2.matches:
        g = vo.generate(0, int(m.hex, 16))
        assert (g["address"], g["wif"]) == (m.address, m.wif)
    # refused combinations
    for cfg in (vg.ScanConfig(count=1, random_keys=True, start=5), vg.ScanConfig(count=1, random_keys=True, end=2**64)):
        with pytest.raises(vg.VgenError):
            vg.scan_gpu_with_runner("^1A", cfg, r)
    r.close()


# ---- a device that fails mid-scan (failure detection / recovery) -------------------------------------------

def test_multi_context_scan_survives_a_failing_context(vgh, vo, tmp_path):
    """Three contexts stripe one range; one of them starts failing after a few dispatches (vgen_debug_fail_after: what a
    device dropping off the bus looks like to the host loop).  The batches it had finished keep their matches, a surviving
    context takes its stripe over from the last finished batch, and the result is the oracle's scan of the WHOLE range —
    nothing lost, nothing twice.  The reference's answer to a failing GPU is its CPU fallback (src/lib.rs:727-746,1185-1198);
    here the other GPUs are the fallback.  (Fault injection exists only in the test build of the library: `vgh`.)"""
    vg = vgh
    batch = 8192
    lo, hi = 0x20000, 0x20000 + 30 * batch - 1
    want = [(x["address"], x["wif"]) for x in vo.scan_range(0, "^1[A-D]", lo, hi, count=10**9)["matches"]]
    assert len(want) > 500
    cfg = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, start=lo, end=hi)
    for victim, after in ((1, 3), (0, 0), (2, 7)):
        rs = [vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=2) for _ in range(3)]
        rs[victim].fail_after(after)
        res = vg.scan_gpu_with_runner("^1[A-D]", cfg, rs)
        assert [(m.address, m.wif) for m in res.matches] == want, (victim, after)
        assert res.failed_shards == 1 and res.complete and res.operations >= 30 * batch
        for r in rs:
            r.close()
    # two of three fail: the last one standing finishes all three stripes
    rs = [vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=2) for _ in range(3)]
    rs[0].fail_after(2)
    rs[2].fail_after(5)
    res = vg.scan_gpu_with_runner("^1[A-D]", cfg, rs)
    assert [(m.address, m.wif) for m in res.matches] == want and res.failed_shards == 2 and res.complete
    # all three fail: the call fails, and still hands over the matches of every batch that had finished
    for r, n in zip(rs, (1, 4, 2)):
        r.fail_after(n)
    rs[1].fail_after(4)
    with pytest.raises(vg.VgenError) as e:
        vg.scan_gpu_with_runner("^1[A-D]", cfg, rs)
    assert e.value.status == -3 and "injected device failure" in str(e.value)
    part = e.value.partial
    got = [(m.address, m.wif) for m in part.matches]
    assert not part.complete and part.failed_shards == 3 and 0 < len(got) < len(want)
    assert set(got) <= set(want) and part.operations % batch == 0 and 0 < part.operations < 30 * batch
    # a failed context is not poisoned: disarmed, the same contexts complete the scan
    for r in rs:
        r.fail_after(2**64 - 1)
    res = vg.scan_gpu_with_runner("^1[A-D]", cfg, rs)
    assert [(m.address, m.wif) for m in res.matches] == want and res.failed_shards == 0
    # a count-limited scan absorbs the failure as well
    rs[1].fail_after(1)
    cfg5 = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=7, start=lo, end=hi)
    res = vg.scan_gpu_with_runner("^1[A-D]", cfg5, rs)
    assert len(res.matches) == 7 and set((m.address, m.wif) for m in res.matches) <= set(want)
    # with a checkpoint the ledger carries the failed stripe's progress: the survivors finish it and the file says complete
    rs[1].fail_after(2)
    ck = str(tmp_path / "multi.ck")
    cfgk = vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, start=lo, end=hi, checkpoint_path=ck)
    res = vg.scan_gpu_with_runner("^1[A-D]", cfgk, rs)
    assert [(m.address, m.wif) for m in res.matches] == want and res.complete and res.failed_shards == 1
    assert "complete=1" in open(ck).read()
    # single-context scan: the error comes with the finished batches' matches
    rs[0].fail_after(6)
    with pytest.raises(vg.VgenError) as e:
        vg.scan_gpu_with_runner("^1[A-D]", cfg, rs[0])
    part = e.value.partial
    assert part.failed_shards == 1 and not part.complete and 0 < len(part.matches) < len(want)
    assert [(m.address, m.wif) for m in part.matches] == want[:len(part.matches)]     # a prefix: batches finish in order
    for r in rs:
        r.close()


def test_frames_on_every_stream_priority_level_make_progress_beside_a_competitor(vg, vo):
    """Twelve frames own twelve hardware queues because the context spreads its streams over the runtime's stream priority
    levels (runtime.cpp: create_stream): frames 4-7 sit on the high level, 8-11 on the low one.  That is an observation about
    ROCm 7.2, not a contract, so this checks what a host would notice if it broke: with another context (four
    normal-priority frames, a host application's own work) saturating the device at the same time, dispatches on EVERY
    level — the low one included — keep completing, none is starved, and both contexts still produce correct results."""
    import threading
    import time
    batch = 1 << 18
    a = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=12, timing=True)
    b = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=4, timing=False)
    t = a.topology()
    assert t["streams"] == 12 and not t["oversubscribed"]
    pat = vg.Pattern("^1Cat", False, vg.AddressFormat.P2pkh)
    a.set_filter(pat)
    b.set_filter(pat)
    k0 = vo.seed_key(77, 0)
    stop = threading.Event()
    done_b = [0]

    def competitor():
        step = 0
        for f in range(4):
            b.dispatch(k0 + (1 << 60) + step * batch, f)
            step += 1
        f = 0
        while not stop.is_set():
            b.wait(f)
            done_b[0] += 1
            b.dispatch(k0 + (1 << 60) + step * batch, f)
            step += 1
            f = (f + 1) % 4
        for f in range(4):
            b.wait(f)

    th = threading.Thread(target=competitor)
    th.start()
    per_frame_ms = [[] for _ in range(12)]
    counts = [0] * 12
    step = 0
    for f in range(12):
        a.dispatch(k0 + step * batch, f)
        step += 1
    t0 = time.time()
    f = 0
    while time.time() - t0 < 1.5:
        a.wait(f)
        per_frame_ms[f].append(a.dispatch_ms(f))
        counts[f] += 1
        a.dispatch(k0 + step * batch, f)
        step += 1
        f = (f + 1) % 12
    for f in range(12):
        a.wait(f)
    stop.set()
    th.join()
    assert min(counts) >= 20 and done_b[0] >= 80, (counts, done_b)            # everybody made real progress
    level = lambda lo: sum(sum(per_frame_ms[f]) / len(per_frame_ms[f]) for f in range(lo, lo + 4)) / 4
    normal, high, low = level(0), level(4), level(8)
    # the low-priority frames may wait longer for wave slots, but within the same order of magnitude: not starved
    assert low < 8 * min(normal, high) and max(normal, high, low) < 50.0, (normal, high, low)
    # results stay right under contention: one dispatch of each context against the oracle
    a.set_filter(None)
    a.dispatch(k0, 0)
    blob, _, _ = a.await_result(0)
    sample = list(range(0, batch, 4099))
    for i in sample:
        assert blob[20 * i:20 * i + 20] == vo.payload(0, k0 + i)
    a.close()
    b.close()


def test_dump_mode_pins_a_bounded_amount_of_host_memory(vg, vo):
    """Dump mode mirrors every payload of a dispatch in pinned host memory; an endomorphism context dumps six images per key,
    so at the CLI's defaults (2^20 keys, 12 frames) all frames together would pin 1.5 GB (24 GB at 2^24 keys per dispatch).
    The context therefore gives dump buffers to as many frames as fit ~1 GiB, never fewer than two: the other frames refuse a
    dump-mode dispatch with a state error naming the limit, filter-mode dispatches use all frames as before, and a scan that
    falls back to host filtering keeps to the frames that have buffers."""
    batch = 1 << 20
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=12, endo=True)
    r.set_filter(None)
    k0 = vo.seed_key(3, 3)
    r.dispatch(k0, 0)
    r.dispatch(k0 + batch, 7)                       # 8 frames x 6 x 2^20 x 20 B = 1.0 GB: frames 0..7 have buffers
    with pytest.raises(vg.VgenError) as e:
        r.dispatch(k0 + 2 * batch, 8)
    assert e.value.status == -5 and "dump mode serves frames 0..7" in str(e.value)
    blob, _, tested = r.await_result(0)
    assert tested == 6 * batch and blob[:20 * 4096] == vo.payload_seq(0, k0, 4096)
    r.await_result(7)
    r.set_filter(vg.Pattern("^1Cat", False, vg.AddressFormat.P2pkh))
    for f in (8, 11):                                # filter mode: every frame
        r.dispatch(k0, f)
        r.wait(f)
    res = vg.scan_gpu_with_runner("^1[2-9A-Za-z]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=50), r)   # host filtering of full dumps
    assert len(res.matches) == 50
    for m in res.matches[:10]:
        assert vo.generate(0, int(m.hex, 16))["address"] == m.address
    r.close()


def test_random_keys_on_an_endomorphism_context_test_six_keys_per_draw(vg, vo):
    """On a VGEN_FLAG_ENDO context the arbitrary-scalar path hashes the six images of every point too
    (keys_bwd_kernel<FMT, FULL, ENDO>): a draw k of the candidate stream is tested as k, lambda k, lambda^2 k and their negations,
    image v of candidate i at index v * batch + i.  Every entry of such a dump against the oracle's payload of exactly that key;
    filter mode against the oracle's regex over the dump; seeded scans against an oracle-side enumeration in index order."""
    batch = 8192
    for fmt in (0, 5, 4, 2):
        r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat(fmt), frames=2, endo=True, match_cap=32768)
        r.set_filter(None)
        r.dispatch_random(11, fmt, 5 * batch, 0)
        blob, _, tested = r.await_result(0)
        assert tested == 6 * batch and len(blob) == 6 * batch * 20
        lanes = range(batch) if fmt == 0 else list(range(0, batch, 131)) + [batch - 1]
        for v in range(6):
            for i in lanes:
                k = vo.random_key(11, fmt, 5 * batch + i)
                assert blob[20 * (v * batch + i):20 * (v * batch + i) + 20] == vo.payload(fmt, variant_key(k, v)), (fmt, v, i)
        if fmt == 0:
            # explicit scalars, fewer than a batch, one of them invalid: images at v * batch + i, nothing for the invalid one
            keys = [5, N - 7, 0, 2**200 + 1, vo.seed_key(1, 1)]
            r.dispatch_keys(keys, 1)
            blob2, _, tested2 = r.await_result(1)
            assert tested2 == 6 * len(keys)
            for v in range(6):
                for i, k in enumerate(keys):
                    want = vo.payload(0, variant_key(k, v)) if 0 < k < N else bytes(20)
                    assert blob2[20 * (v * batch + i):20 * (v * batch + i) + 20] == want, (v, i)
            # filter mode = the oracle's regex over the dump
            pat = vg.Pattern("^1[A-F]", False, vg.AddressFormat.P2pkh)
            r.set_filter(pat)
            r.dispatch_random(11, fmt, 5 * batch, 1)
            recs, n, _ = r.await_result(1)
            ore = vo.Regex("^1[A-F]", False)
            want = [i for i in range(6 * batch) if ore.matches(vo.address_from_hash160(0, blob[20 * i:20 * i + 20]))]
            assert n == len(recs) <= r.match_cap and [i for i, pl in recs if ore.matches(vo.address_from_hash160(0, pl))] == want and len(want) > 300
            # a seeded random-key scan: the first matches in (batch, index) order, keys re-derived through variant + stream
            res = vg.scan_gpu_with_runner("^1[A-F]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=40, seed=11, random_keys=True), r)
            blob0 = None
            r.set_filter(None)
            r.dispatch_random(11, 0, 0, 0)
            blob0, _, _ = r.await_result(0)
            exp = [i for i in range(6 * batch) if ore.matches(vo.address_from_hash160(0, blob0[20 * i:20 * i + 20]))][:40]
            assert len(exp) == 40 and res.operations == 6 * batch
            assert [int(m.hex, 16) for m in res.matches] == [variant_key(vo.random_key(11, 0, i % batch), i // batch) for i in exp]
            for m in res.matches[:8]:
                g = vo.generate(0, int(m.hex, 16))
                assert (g["address"], g["wif"]) == (m.address, m.wif)
        r.close()
    # two contexts: shard i walks stream i (never the same candidates twice)
    rs = [vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=2, endo=True) for _ in range(2)]
    res = vg.scan_gpu_with_runner("^1Ab", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=12, seed=5, random_keys=True), rs)
    keys = [int(m.hex, 16) for m in res.matches]
    assert len(keys) == 12 and len(set(keys)) == 12
    streams = set()
    for k in keys:
        assert vo.generate(0, k)["address"].startswith("1Ab")
    for r in rs:
        r.close()


def test_a_wide_generator_table_that_cannot_be_had_is_not_an_error(vgh, vo, monkeypatch):
    """The taproot and arbitrary-scalar paths multiply over a multi-gigabyte table built at first use.  When that table cannot
    be allocated or built (here: injected through VGEN_DEBUG_GTAB_FAIL, which only the test build of the library reads) the
    dispatch must not fail: the context steps down through the narrower widths — and, when none can be had, carries on with the
    always-present 8-bit table — slower, same keys; vgen_get_resources says what it got and why, vgen_last_error stays clean."""
    vg = vgh
    batch = 8192
    start = vo.seed_key(77, 3)
    # 24 bits "fail", 22 can be had: one step down
    monkeypatch.setenv("VGEN_DEBUG_GTAB_FAIL", "23")
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2tr, frames=2)
    assert r.resources()["table_bits"] == 0 and r.resources()["note"] == ""       # nothing built before the first dispatch
    assert dump(r, start) == vo.payload_seq(3, start, batch)
    res = r.resources()
    assert res["table_bits"] == 22 and res["table_bits_wanted"] == 24 and "continuing on a 22-bit table" in res["note"], res
    assert vg._L.vgen_last_error(r._h) in (None, b"")                               # a step-down is not an error
    r.close()
    # nothing wide can be had: the 8-bit table
    monkeypatch.setenv("VGEN_DEBUG_GTAB_FAIL", "1")
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2tr, frames=2)
    assert dump(r, start) == vo.payload_seq(3, start, batch)
    res = r.resources()
    assert res["table_bits"] == 8 and res["table_bits_wanted"] == 24 and "wide generator table unavailable" in res["note"] and "8-bit" in res["note"]
    assert vg._L.vgen_last_error(r._h) in (None, b"")
    assert dump(r, start + batch, frame=1) == vo.payload_seq(3, start + batch, batch)     # and again: no retry storm, still right
    r.close()
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=2)
    r.set_filter(None)
    r.dispatch_random(3, 0, 0, 0)
    blob, _, _ = r.await_result(0)
    for i in range(0, batch, 61):
        assert blob[20 * i:20 * i + 20] == vo.payload(0, vo.random_key(3, 0, i))
    r.close()
    # the command line says so too (what a scan absorbs must not pass unseen), and only then
    import os
    import subprocess
    from conftest import HOOKS_CLI
    p = subprocess.run([HOOKS_CLI, "generate", "-f", "p2tr", "-p", "^bc1pq", "-o", "minimal", "--seed", "5"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "Warning: device 0: wide generator table unavailable" in p.stderr and len(p.stdout.split()) == 1
    env = {k: v for k, v in os.environ.items() if k != "VGEN_DEBUG_GTAB_FAIL"}
    q = subprocess.run([HOOKS_CLI, "generate", "-f", "p2tr", "-p", "^bc1pq", "-o", "minimal", "--seed", "5"], capture_output=True, text=True, timeout=120, env=env)
    assert q.returncode == 0 and "Warning: device" not in q.stderr and q.stdout == p.stdout
    # the SHIPPED command line does not know the switch at all
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vgen_amd", "vgen-hip")
    s = subprocess.run([cli, "generate", "-f", "p2tr", "-p", "^bc1pq", "-o", "minimal", "--seed", "5"], capture_output=True, text=True, timeout=120)
    assert s.returncode == 0 and "Warning: device" not in s.stderr and s.stdout == p.stdout


def _scan_until_table(vg, r, pattern, cfg_kwargs, want_bits, timeout_s):
    """Runs a scan under the host's stop flag until the context reports `want_bits` in use (plus a few batches on it), a note says
    it will not come, or the timeout passes.  -> (result, [(t, operations, table_bits)])"""
    import ctypes
    import time
    stop = ctypes.c_int32(0)
    seen = []
    t0 = time.perf_counter()
    extra = [0]

    def progress(ops):
        rs = r.resources()
        seen.append((time.perf_counter(), ops, rs["table_bits"]))
        if rs["table_bits"] == want_bits or rs["note"]:
            extra[0] += 1
        if extra[0] >= 24 or time.perf_counter() - t0 > timeout_s:
            stop.value = 1
    res = vg.scan_gpu_with_runner(pattern, vg.ScanConfig(**cfg_kwargs), r, progress_cb=progress, stop=stop)
    return res, seen


def test_scan_picks_the_generator_table_its_expected_length_pays_for(vg, vo):
    """vgen_scan on the paths that multiply a scalar per key chooses the table by the keys it can expect to test: the default
    24-bit table for short scans, the 27-bit signed one (9 additions) from ~3 s, the 29-bit signed one (8 additions, 138 GB) from
    30 s — as a PREFERENCE the runtime checks against the device's free memory and builds behind the dispatches: the scan starts on the
    table it has and moves when the wider one is complete; a context never steps back down by itself; a device that is not empty may
    refuse the 138 GB (then a note says so and the scan stays where it is).  Results stay the oracle's either way."""
    fmt = vg.AddressFormat.P2tr
    batch = 8192
    r = vg.GpuRunner(batch_size=batch, fmt=fmt, frames=2)
    start = vo.seed_key(31, 3)
    res = vg.scan_gpu_with_runner("^bc1pq[qp]", vg.ScanConfig(format=fmt, count=None, start=start, end=start + 4 * batch - 1), r)
    assert res.complete and r.resources()["table_bits"] == 24 and r.resources()["table_bits_wanted"] == 24   # 32 768 keys: the default table
    want = [x["address"] for x in vo.scan_range(3, "^bc1pq[qp]", start, start + 4 * batch - 1, count=10**9)["matches"]]
    assert [m.address for m in res.matches] == want and len(want) > 10
    # 8.6 G keys = ~6 s at 1.35 Gkeys/s -> 27 bits, signed: arrives in the background within the first second
    res, seen = _scan_until_table(vg, r, "^bc1pq[qp]", dict(format=fmt, count=None, start=start, end=start + 2**33 - 1), 27, 20.0)
    rs = r.resources()
    assert rs["table_bits_wanted"] == 27 and (rs["table_bits"] == 27 or rs["note"]), rs
    assert [w for _, _, w in seen][0] == 24 and res.operations >= batch      # ... and the scan began on the table it had
    for m in res.matches[:40]:
        assert vo.generate(3, int(m.hex, 16))["address"] == m.address
    # ~13 minutes of keys -> 29 bits, signed (138 GB: only when half of the device's free memory covers it)
    bits_before = rs["table_bits"]
    res, seen = _scan_until_table(vg, r, "^bc1pq[qp]", dict(format=fmt, count=None, start=start, end=start + 2**40 - 1), 29, 60.0)
    rs = r.resources()
    mem = r.memory()
    assert rs["table_bits_wanted"] == 29, rs
    if rs["table_bits"] == 29:
        assert rs["note"] == "" and mem["table_bytes"] > 130 * 10**9
    else:   # a device somebody else is using: refused by the memory policy, said so, and the scan went on where it was
        assert rs["table_bits"] == bits_before and "half of the" in rs["note"], rs
    widths = [w for _, _, w in seen]
    assert widths[0] == bits_before and all(a <= b for a, b in zip(widths, widths[1:]))     # never down, never a gap in the reports
    assert [o for _, o, _ in seen] == [batch * (i + 1) for i in range(len(seen))]
    for m in res.matches[:40] + res.matches[-40:]:
        assert vo.generate(3, int(m.hex, 16))["address"] == m.address
    final = rs["table_bits"]
    res = vg.scan_gpu_with_runner("^bc1pq[qp]", vg.ScanConfig(format=fmt, count=None, start=start, end=start + 4 * batch - 1), r)
    assert r.resources()["table_bits"] == final and [m.address for m in res.matches] == want   # a short scan afterwards: no stepping down
    r.close()


def test_a_table_cap_and_a_memory_budget_bound_what_a_long_scan_takes(vg, vo):
    """The caller's side of the policy (vgen_scan_config.table_bits_max, vgen_params.device_mem_budget_bytes, vgen_params.table_bits):
    a scan that expects to run for 40 s stays on 24 bits when capped there, stays within a 1 GB budget on a table of at most 20
    bits (872 MB) — with the oracle's results —, a context created with table_bits = 22 uses exactly that, a budget the frames
    alone pass fails vgen_create, and vgen_get_memory accounts for all of it."""
    fmt = vg.AddressFormat.P2tr
    batch = 8192
    start = vo.seed_key(32, 4)
    want = [x["address"] for x in vo.scan_range(3, "^bc1pq[qp]", start, start + 4 * batch - 1, count=10**9)["matches"]]
    span = 2**36      # ~51 s of keys at 1.35 Gkeys/s: the scan would ask for 29 bits

    def short_run(r, **kw):
        import ctypes
        import threading
        stop = ctypes.c_int32(0)
        t = threading.Timer(0.6, lambda: setattr(stop, "value", 1))
        t.start()
        res = vg.scan_gpu_with_runner("^bc1pq[qp]", vg.ScanConfig(format=fmt, count=None, start=start, end=start + span - 1, **kw), r, stop=stop)
        t.cancel()
        return res
    # capped by the scan
    r = vg.GpuRunner(batch_size=batch, fmt=fmt, frames=2)
    res = short_run(r, table_bits_max=24)
    rs = r.resources()
    assert rs["table_bits"] == 24 and rs["table_bits_wanted"] == 24 and rs["note"] == "", rs
    assert [m.address for m in res.matches[:len(want)]] == want
    m = r.memory()
    assert 11 * 10**9 < m["table_bytes"] < 13 * 10**9 and m["budget_bytes"] == 0 and m["frames_bytes"] > 0 and m["device_total_bytes"] > 200 * 10**9
    r.close()
    # bounded by the context's budget: 1 GB holds the frames (a few MB at this batch size) and the 20-bit table, nothing wider
    r = vg.GpuRunner(batch_size=batch, fmt=fmt, frames=2, device_mem_budget_bytes=10**9)
    res = short_run(r)
    rs, m = r.resources(), r.memory()
    assert rs["table_bits"] <= 22 and rs["table_bits"] == 20 and "device_mem_budget_bytes" in rs["note"], rs
    assert m["frames_bytes"] + m["mode_bytes"] + m["table_bytes"] <= 10**9 and m["budget_bytes"] == 10**9, m
    assert [x.address for x in res.matches[:len(want)]] == want
    r.close()
    # a width by name
    r = vg.GpuRunner(batch_size=batch, fmt=fmt, frames=2, table_bits=22)
    res = short_run(r)
    assert r.resources()["table_bits"] == 22 and r.resources()["table_bits_wanted"] == 22
    assert [x.address for x in res.matches[:len(want)]] == want
    r.close()
    # a budget the frames themselves pass
    with pytest.raises(vg.VgenError, match="does not cover the frames"):
        vg.GpuRunner(batch_size=1 << 20, fmt=fmt, frames=12, device_mem_budget_bytes=10**8)


def test_a_scan_that_turns_out_long_moves_to_the_wide_table_on_the_way(vg, vo):
    """A taproot scan whose pattern gives the loop no estimate (the whole DFA on the device, a count it will not reach) starts on the
    default table; after five seconds it asks for the 29-bit signed table, which is allocated by a thread of its own and built in
    slices that ride in front of the scan's own dispatches — the scan never pauses (no gap between two finished batches beyond a few
    batch times), the matches from both sides of the switch are the oracle's, in key order, none lost or doubled at the seam.  (Here
    the scan is capped at 27 bits — 21.5 GB instead of 138 — so that the test also runs on a device that is not empty.)"""
    import ctypes
    import time
    fmt = vg.AddressFormat.P2tr
    r = vg.GpuRunner(batch_size=1 << 18, fmt=fmt, frames=4)
    pat = vg.Pattern("qqqqq", False, fmt)
    assert pat.device_kind == 4
    start = vo.seed_key(77, 7)
    stop = ctypes.c_int32(0)
    seen = []
    after = [0]

    def progress(ops):
        rs = r.resources()
        seen.append((time.perf_counter(), ops, rs["table_bits"]))
        if rs["table_bits"] == 27 or rs["note"]:
            after[0] += 1
        if after[0] > 1500 or time.perf_counter() - seen[0][0] > 40.0:
            stop.value = 1
    res = vg.scan_gpu_with_runner("qqqqq", vg.ScanConfig(format=fmt, count=10**9, start=start, table_bits_max=27), r, progress_cb=progress, stop=stop)
    widths = [w for _, _, w in seen]
    assert widths[0] == 24 and widths[-1] == 27 and sorted(set(widths)) == [24, 27], (sorted(set(widths)), r.resources())
    switch = next(i for i, w in enumerate(widths) if w == 27)
    assert seen[switch][0] - seen[0][0] > 5.0                           # asked for after five seconds, there some time later
    assert [o for _, o, _ in seen] == [(i + 1) << 18 for i in range(len(seen))]   # operations advance batch by batch across the seam
    # no pause: batch completions keep coming while the table is allocated, built and switched to
    gaps = [b[0] - a[0] for a, b in zip(seen, seen[1:])]
    steady = sorted(gaps[:switch // 2])[len(gaps[:switch // 2]) // 2]
    t_ask = seen[0][0] + 5.0
    during = [g for (a, g) in zip(seen, gaps) if a[0] >= t_ask - 0.5]
    print(f"table switch: steady batch interval {steady * 1e3:.3f} ms, longest interval from the request to the end {max(during) * 1e3:.3f} ms, "
          f"switch {seen[switch][0] - t_ask:.2f} s after the request")
    assert max(during) < max(8 * steady, 0.004), (steady, max(during))
    keys = [int(m.hex, 16) for m in res.matches]
    assert keys == sorted(keys) and len(set(keys)) == len(keys) and all(start <= k < start + res.operations for k in keys)
    ops_at_switch = seen[switch - 1][1]
    before = [k for k in keys if k < start + ops_at_switch]
    behind = [k for k in keys if k >= start + ops_at_switch]
    assert len(before) > 50 and len(behind) > 20, (len(before), len(behind))
    ore = vo.Regex("qqqqq", False)
    for m in res.matches[::max(1, len(res.matches) // 150)] + res.matches[-20:]:
        g = vo.generate(3, int(m.hex, 16))
        assert g["address"] == m.address and ore.matches(m.address)
    # nothing lost around the seam: the oracle's own scan of the two batches either side of it
    lo = start + ops_at_switch - (1 << 18)
    want = [x["key"] for x in vo.scan_range(3, "qqqqq", lo, lo + (2 << 18) - 1, count=10**9)["matches"]]
    assert [k for k in keys if lo <= k < lo + (2 << 18)] == want
    r.close()


def test_contexts_on_one_device_share_the_wide_generator_table(vg, vo):
    """The wide table (11.8 GB at the default width) is built once per device and shared by the process's contexts there,
    reference-counted: a second context's first dispatch builds nothing, both compute the oracle's keys, and the table goes
    away with its last user (device memory in use returns to where it was)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value
    batch = 8192
    start = vo.seed_key(5, 5)
    warm = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=1)     # (the runtime's own one-time allocations first)
    warm.close()
    free0 = free_bytes()
    a = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2tr, frames=2)
    assert dump(a, start) == vo.payload_seq(3, start, batch)
    ra = a.resources()
    assert ra["table_bits"] == ra["table_bits_wanted"] == 24 and ra["note"] == ""
    free1 = free_bytes()
    assert free0 - free1 > 11 * 2**30                      # the table is there
    b = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=2)
    b.set_filter(None)
    b.dispatch_random(3, 0, 0, 0)
    blob, _, _ = b.await_result(0)
    for i in range(0, batch, 97):
        assert blob[20 * i:20 * i + 20] == vo.payload(0, vo.random_key(3, 0, i))
    assert b.resources()["table_bits"] == 24
    free2 = free_bytes()
    assert free1 - free2 < 2 * 2**30                       # ... and was not built a second time
    a.close()                                              # the first user goes: the second keeps computing on the shared table
    b.dispatch_random(3, 0, batch, 1)
    blob, _, _ = b.await_result(1)
    for i in range(0, batch, 97):
        assert blob[20 * i:20 * i + 20] == vo.payload(0, vo.random_key(3, 0, batch + i))
    b.close()
    assert free0 - free_bytes() < 2**30                     # the last user took the table with it


def test_dump_mode_serves_the_frames_vgen_get_resources_names(vg, vo):
    """Dump mode's pinned host mirrors are bounded (~1 GiB): vgen_get_resources reports how many frames it serves, a
    dispatch beyond them is VGEN_E_STATE (not a crash, not a silent allocation), one below works."""
    batch = 1 << 20
    r = vg.GpuRunner(batch_size=batch, fmt=vg.AddressFormat.P2pkh, frames=12, endo=True)      # 6 x 20 MB per frame
    n = r.resources()["dump_frames"]
    assert n == 8, n
    r.set_filter(None)
    start = vo.seed_key(8, 1)
    r.dispatch(start, n - 1)
    blob, _, tested = r.await_result(n - 1)
    assert tested == 6 * batch and blob[:20 * 64] == vo.payload_seq(0, start, 64)
    with pytest.raises(vg.VgenError) as e:
        r.dispatch(start, n)
    assert e.value.status == -5 and "vgen_get_resources" in str(e.value)
    r.close()
    r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, frames=12)
    assert r.resources()["dump_frames"] == 12
    r.close()


def test_cli_progress_line_only_on_a_terminal(vg):
    """The reference's spinner (src/lib.rs:783-805) lives on stderr, only when that is a terminal and not with --quiet: the CLI
    redraws "[elapsed] Checked N addresses" there from the scan's progress callback and clears it before the results.
    (The GPU boxes have no pty devices: VGEN_PROGRESS=1 stands in for the terminal.)"""
    import os
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vgen_amd", "vgen-hip")
    args = [cli, "range", "-r", "1:3ffffffff", "-p", "^1ZZZZZZZZZ", "-o", "minimal"]      # 2^34 keys, no match: a second or so
    forced = dict(os.environ, VGEN_PROGRESS="1")
    p = subprocess.run(args, capture_output=True, text=True, timeout=120, env=forced)
    assert p.returncode == 0 and p.stdout == "" and "Checked " in p.stderr and " addresses" in p.stderr
    assert p.stderr.rstrip().endswith(")") and "No match found after 17,179,869,184 operations" in p.stderr.split("\x1b[K")[-1]
    p = subprocess.run(args + ["-q"], capture_output=True, text=True, timeout=120, env=forced)
    assert p.returncode == 0 and "Checked" not in p.stderr
    plain = {k: v for k, v in os.environ.items() if k != "VGEN_PROGRESS"}
    q = subprocess.run(args, capture_output=True, text=True, timeout=120, env=plain)      # stderr is a pipe: no progress line
    assert q.returncode == 0 and "Checked" not in q.stderr and "No match found after" in q.stderr


def test_random_key_scan_resumes_through_a_checkpoint(vg, vo, tmp_path):
    """VGEN_SCAN_RANDOM_KEYS with checkpoint_path: two legs (2 + 3 batches) find what one scan of five batches finds — the
    oracle-checked stream walk of test_random_key_scan_* —, also with six images per draw; the file names its mode and seed."""
    for endo in (False, True):
        r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, frames=3, endo=endo)
        whole = vg.scan_gpu_with_runner("^1[A-D][a-k]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, seed=21, random_keys=True, max_batches=5), r)
        ck = str(tmp_path / f"rk{int(endo)}.ckpt")
        a = vg.scan_gpu_with_runner("^1[A-D][a-k]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, seed=21, random_keys=True, max_batches=2, checkpoint_path=ck), r)
        b = vg.scan_gpu_with_runner("^1[A-D][a-k]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, seed=21, random_keys=True, max_batches=3, checkpoint_path=ck), r)
        assert 0 < len(a.matches) < len(b.matches) and b.resumed_operations == a.operations and a.operations + b.operations == whole.operations
        assert [(m.address, m.wif) for m in b.matches] == [(m.address, m.wif) for m in whole.matches] and len(whole.matches) > 100
        for m in whole.matches[::17]:
            assert vo.generate(0, int(m.hex, 16))["address"] == m.address
        text = open(ck).read()
        # base = eight zero bytes, then the 24 seed bytes: the u64 21 little-endian, sixteen zero bytes
        assert "mode=random-seed24" in text and ("base=" + "00" * 8 + "15" + "00" * 23) in text
        with pytest.raises(vg.VgenError):   # another seed: not this scan's file
            vg.scan_gpu_with_runner("^1[A-D][a-k]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, seed=22, random_keys=True, max_batches=1, checkpoint_path=ck), r)
        r.close()
    # An UNSEEDED random-key scan: the seed is all the secret its keys have, so it is 24 bytes of OS entropy, not 64 bits —
    # the file shows it; the found keys are candidates of that very seed; a second leg adopts it from the file.
    r = vg.GpuRunner(batch_size=8192, fmt=vg.AddressFormat.P2pkh, frames=2)
    ck = str(tmp_path / "unseeded.ckpt")
    a = vg.scan_gpu_with_runner("^1[A-D]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, random_keys=True, max_batches=1, checkpoint_path=ck), r)
    base = bytes.fromhex(re.search(r"^base=([0-9a-f]{64})$", open(ck).read(), re.M).group(1))
    seed = base[8:]
    assert base[:8] == bytes(8) and any(seed[8:16]) and any(seed[16:24])          # entropy beyond the first 64 bits
    assert len(a.matches) > 100
    cand = {vo.random_key(seed, 0, i) for i in range(8192)}
    assert all(int(m.hex, 16) in cand for m in a.matches)
    b = vg.scan_gpu_with_runner("^1[A-D]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=None, random_keys=True, max_batches=1, checkpoint_path=ck), r)
    cand2 = {vo.random_key(seed, 0, 8192 + i) for i in range(8192)}
    assert len(b.matches) > len(a.matches) and all(int(m.hex, 16) in cand | cand2 for m in b.matches)
    # two unseeded scans never share a seed
    c = vg.scan_gpu_with_runner("^1[A-D]", vg.ScanConfig(format=vg.AddressFormat.P2pkh, count=5, random_keys=True), r)
    assert not {int(m.hex, 16) for m in c.matches} & (cand | cand2)
    r.close()

