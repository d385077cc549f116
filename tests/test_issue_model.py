"""The issue-stage model of round 5 (tools/issue_model.py) against the probe measurements it was read from, and the per-class census
bench.py's `frac_of_slot_bound` rests on (tools/issue_classes.py -> profiles/r05_issue_classes.json) against the assembly of this build.

The probes themselves (tools/ubench_phase3.hip, one asm statement of 1 024 instructions per stream, the waves of a SIMD barrier-locked) run on
the MI355X; their committed results are data here."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import issue_model as im  # noqa: E402


def probe_patterns():
    src = open(os.path.join(ROOT, "tools", "ubench_phase3_gen.py")).read()
    body = src[src.index("pats = {"):src.index("# the same streams over 1 / 2 dependency chains")]
    scope = {}
    exec(body, scope)      # the dict literal of the generator (pattern strings only)
    return scope["pats"]


def measured(mode_barrier=1, waves=4):
    out = {}
    for line in open(os.path.join(ROOT, "profiles", "r05_phase3_ubench.jsonl")):
        if line.startswith("{"):
            d = json.loads(line)
            if d["barrier_per_block"] == mode_barrier and d["waves_per_simd"] == waves:
                out[d["stream"]] = d["cycles_per_waveinstr"]
    return out


def test_the_model_reproduces_the_barrier_locked_probes():
    pats, meas = probe_patterns(), measured()
    assert len(meas) >= 23 and set(meas) <= set(pats)
    worst = 0.0
    for name, m in meas.items():
        model = im.cycles_per_instruction(pats[name])
        worst = max(worst, abs(model - m) / m)
        assert abs(model - m) / m < 0.06, (name, model, m)
    assert worst > 0.001      # (it is a model, not a copy of the table)


def test_what_the_probes_say_in_words():
    pats, meas = probe_patterns(), measured()
    # any mix of full-rate and half-rate instructions under age-only arbitration: four cycles per instruction, the full-rate ones included
    for name in ("A3R1", "A1R1", "A4R4", "SHA", "M1A3"):
        assert 3.85 < meas[name] < 4.1
    # a priority change around the half-rate runs: the 1:1 stream at its bound, the SHA-256-like multiset a quarter faster
    assert meas["A1R1_P"] < 2.1 and meas["SHA_G"] < 3.05 and meas["SHA_P2"] < 3.2
    # priority on the WRONG class does nothing
    assert meas["A4R4_LO"] > 3.8
    assert im.slot_bound(0, 1, 1) == 1 and im.slot_bound(1, 0, 0) == 1 and im.slot_bound(512, 1299, 1139.5) == 1811


def test_the_class_census_is_the_one_of_this_build():
    """bench.py prices the steady state against X + max(C, (C + S) / 2) from profiles/r05_issue_classes.json: the file must describe the kernel
    the Makefile builds (a changed hash block or point arithmetic regenerates it: python tools/issue_classes.py > profiles/r05_issue_classes.json)."""
    committed = json.load(open(os.path.join(ROOT, "profiles", "r05_issue_classes.json")))
    fresh = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "issue_classes.py")]))
    for k, v in committed["per_key"].items():
        assert abs(fresh["per_key"][k] - v) <= 0.01 * max(1.0, v), (k, fresh["per_key"][k], v)
    assert abs(fresh["issue_slots_per_key_at_least"] - committed["issue_slots_per_key_at_least"]) < 10
    # and agrees with the counters: VALU instructions per key as rocprofv3 counted them (profiles/pmc_valu.json)
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))["p2pkh:1048576"]
    assert abs(committed["per_key"]["valu"] - pmc["valu_instr_per_key"]) / pmc["valu_instr_per_key"] < 0.01
