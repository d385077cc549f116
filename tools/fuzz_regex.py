#!/usr/bin/env python3
"""Robustness fuzz of the pattern front end (regex_dfa.cpp + filter.cpp) under AddressSanitizer / UBSan on the CPU.

The pattern string is the one input of the C-ABI that comes straight from a user (vgen_filter_compile), so the
parser, the subset construction and the prefilter derivation must reject or compile ANY byte string without
touching memory they do not own and within the state cap.  This builds the host sources with
-fsanitize=address,undefined into /tmp, re-runs itself with the sanitizer runtimes preloaded and throws random
regex-shaped strings at filter_compile for every address format; any sanitizer report aborts the run.

  python tools/fuzz_regex.py [--n 20000] [--seed 1]

Prints the number of patterns tried / accepted and the slowest compile.  GPU sanitizers are not available on this
pool; the device side of the filter is covered by the parity tests instead.
"""
import argparse
import ctypes
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = "/tmp/libcoretest_asan.so"

ATOMS = ["1", "3", "b", "c", "q", "A", "Z", "a", "z", "0", "9", "x", "f", ".", "\\d", "\\w", "\\s", "\\D", "\\W", "\\b", "\\B",
         "\\<", "\\>", "\\b{start}", "\\b{end}", "\\b{start-half}", "\\b{", "\\pL", "\\p{Lu}", "\\P{Greek}", "\\p{", "\\p{^N}",
         "[[:alpha:]]", "[[:^digit:]]", "[[:", "[a-c]", "[^1-5]", "[a-z&&[^aeiou]]", "[0-9--4]", "[a-f~~c-h]", "[]a]", "[^]]",
         "[\\]]", "[a-", "[", "]", "(", ")", "(?:", "(?i)", "(?-i)", "(?i:", "(?x)", "(?s)", "(?m)", "(?U)", "(?u)", "(?-u)",
         "(?P<n>", "(?<n>", "(?", "|", "*", "+", "?", "*?", "+?", "??", "{2}", "{1,3}", "{,3}", "{3,}", "{", "}", "{99999}",
         "{0}", "{1000}", "^", "$", "\\A", "\\z", "\\x41", "\\x{41}", "\\x{110000}", "\\u0041", "\\u{1F600}", "\\U00000041",
         ".{20}$", "[ab]{14}$", "a.{13}\\b", "(a|bb).{17}", "\\x", "\\", "\\Q", "\\0", "\\1", "\\n", "\\t", " ", "#", "\n", "\x80", "\xff", "\xc3\xa9", "1Cat", "bc1q", "0xdead"]


def gen(rng):
    k = rng.choice([1, 2, 3, 5, 8, 13, 30])
    s = "".join(rng.choice(ATOMS) for _ in range(k))
    if rng.random() < 0.1:      # raw bytes
        s = "".join(chr(rng.randrange(1, 256)) for _ in range(rng.randrange(1, 40)))
    return s.encode("latin-1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    if os.environ.get("VGEN_FUZZ_CHILD") != "1":
        csrc = os.path.join(ROOT, "vgen_amd", "csrc")
        host = [os.path.join(csrc, "host", f) for f in
                ("host_ec.cpp", "encode.cpp", "regex_dfa.cpp", "filter.cpp", "pattern_info.cpp", "provider.cpp")]
        subprocess.check_call(["g++", "-O1", "-g", "-fPIC", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                               "-fno-sanitize-recover=undefined", "-Wno-unknown-pragmas", "-I" + os.path.join(ROOT, "include"),
                               "-shared", "-o", LIB, os.path.join(ROOT, "tests", "native", "core_shim.cpp")] + host + ["-lpthread"])
        rt = ":".join(subprocess.check_output(["gcc", "-print-file-name=" + n], text=True).strip() for n in ("libasan.so", "libubsan.so"))
        env = dict(os.environ, VGEN_FUZZ_CHILD="1", LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--n", str(a.n), "--seed", str(a.seed)], env=env))

    lib = ctypes.CDLL(LIB)
    lib.core_filter_ntests.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint]
    lib.core_regex_match.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    rng = random.Random(a.seed)
    ok = 0
    slow = (0.0, b"")
    for i in range(a.n):
        p = gen(rng)
        if b"\0" in p:
            continue
        fmt = rng.randrange(6)
        ci = rng.randrange(2)
        t0 = time.perf_counter()
        r = lib.core_filter_ntests(p, ci, fmt)
        lib.core_regex_match(p, ci, b"1CatXyZ bc1q-0xdead_9\xc3\xa9")
        dt = time.perf_counter() - t0
        ok += r >= 0
        if dt > slow[0]:
            slow = (dt, p)
        if i % 2000 == 0:
            print(f"[fuzz] {i} patterns, {ok} accepted, slowest {slow[0]*1e3:.0f} ms {slow[1]!r}", file=sys.stderr, flush=True)
    print(f"fuzz_regex: {a.n} patterns, {ok} accepted, no sanitizer report; slowest compile {slow[0]*1e3:.0f} ms for {slow[1]!r}")


if __name__ == "__main__":
    main()
