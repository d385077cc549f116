"""Host-side sanitizer runs ("host ASan/TSan builds of the C++ host"; the reference gets memory and thread
safety from `unsafe_code = "forbid"`, Cargo.toml:14).

The part of libvgen_hip.so that needs a device to run — scanner.cpp (worker pool, helper-thread ramp, shared counters,
checkpoint lock, multi-context threads, failure take-over) and cabi.cpp — is linked against a CPU stand-in of the runtime
interface (tests/native/fake_rt.cpp, over the product's own core/*.h, NOT the oracle) and built twice by g++: with
ThreadSanitizer and with AddressSanitizer + UBSan.  tests/native/fake_driver.cpp then drives the C ABI through twelve
scenarios (range scan with progress callback, stop flag from another thread, checkpoint / resume, three striped contexts,
ring growth + host-filter pool, a failing context taken over by the others, random keys (also checkpointed and resumed), endomorphism images and the other
formats, the frame-level API, ranges that are not whole batches incl. the end of the key space) and a randomised walk over
formats x pattern kinds x ranges x counts x contexts x ring sizes x injected failures (`fuzz`: 12 cases here, more with
VGEN_FAKE_FUZZ_CASES / VGEN_FAKE_FUZZ_SEED), and checks every result against the oracle.  CPU only: no sanitizer runs on the GPU box.
"""
import os
import subprocess

import pytest

from conftest import locked_make

HERE = os.path.dirname(os.path.abspath(__file__))
NATIVE = os.path.join(HERE, "native")
SCENARIOS = ["range_scan", "stop_flag", "checkpoint", "multi_context", "ring_growth", "failure_takeover", "random_keys",
             "endo_and_formats", "dispatch_api", "edge_ranges", "fuzz", "random_checkpoint"]


@pytest.fixture(scope="module")
def runs():
    locked_make("-s", "-C", os.path.join(HERE, "..", "oracle"))
    locked_make("-s", "-j4", "-C", NATIVE, "sanitizers")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    procs = {k: subprocess.Popen([os.path.join(NATIVE, f"fake_driver_{k}")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                 text=True, env=env) for k in ("tsan", "asan")}      # side by side: ~2 minutes in all
    out = {}
    for k, p in procs.items():
        try:
            so, se = p.communicate(timeout=1500)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
            se += "\nTIMEOUT"
        out[k] = (p.returncode, so, se)
    return out


@pytest.mark.parametrize("kind,marker", [("tsan", "ThreadSanitizer"), ("asan", "AddressSanitizer")])
def test_host_scan_loop_is_clean_under_the_sanitizer(runs, kind, marker):
    rc, so, se = runs[kind]
    assert marker not in se and "runtime error" not in se and "LeakSanitizer" not in se, se[-4000:]
    assert rc == 0, (so[-1500:], se[-2500:])
    for name in SCENARIOS:
        assert any(line.startswith(name) and " ok " in line for line in so.splitlines()), (name, so)
