"""Wall-clock of whole CLI runs (process start, HIP initialisation, context, scan, output)."""
import os, subprocess, sys, time
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vgen_amd", "vgen-hip")
runs = [["list-gpus"], ["generate", "-p", "^1Cat", "-o", "minimal"], ["generate", "-p", "^1Cat", "-o", "minimal"],
        ["generate", "-p", "^1Cats", "-o", "minimal"], ["generate", "-p", "^1CatsX", "-o", "minimal"],
        ["generate", "-p", "^bc1pqqq", "-f", "p2tr", "-o", "minimal"], ["generate", "-p", "^0xdead", "-f", "ethereum", "-i", "-o", "minimal"],
        ["generate", "-p", "dead$", "-f", "p2wpkh", "-o", "minimal"], ["range", "--range", "1:fffff", "-p", "^1Cat", "-c", "0", "-o", "minimal"],
        ["generate", "-p", "^1Cat", "-o", "minimal", "--no-endo", "--seed", "7"]]
for a in runs:
    t = time.perf_counter()
    p = subprocess.run([exe] + a, capture_output=True, text=True, timeout=120)
    dt = time.perf_counter() - t
    print("%-62s rc=%d %6.0f ms  %s" % (" ".join(a), p.returncode, dt * 1e3, (p.stdout.strip().splitlines() or [p.stderr.strip()[-80:]])[0][:70]))
