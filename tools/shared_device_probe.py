"""The scan beside another tenant of the same GPU (a torch bf16 GEMM loop in a process of its own): what each gets.
usage: python tools/shared_device_probe.py"""
import os, re, signal, subprocess, sys, time
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vgen_amd", "vgen-hip")
GEMM = r'''
import sys, time, torch
secs = float(sys.argv[1])
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
for _ in range(5): (a @ b)
torch.cuda.synchronize()
n = 0; t0 = time.perf_counter()
while time.perf_counter() - t0 < secs:
    for _ in range(10): c = a @ b
    torch.cuda.synchronize(); n += 10
dt = time.perf_counter() - t0
print("GEMM %.1f TFLOP/s" % (n * 2 * 8192**3 / dt / 1e12), flush=True)
'''
def scan(frames, secs):
    return subprocess.Popen([exe, "range", "-p", "boha:b1000:66", "-l", "12", "-o", "json", "--frames", str(frames)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
def rate(p, secs):
    p.send_signal(signal.SIGINT)
    out, err = p.communicate(timeout=120)
    m = re.search(r"No match found after ([\d,]+) operations", err)
    return int(m.group(1).replace(",", "")) / secs / 1e9 if m else None
g = subprocess.run([sys.executable, "-c", GEMM, "6"], capture_output=True, text=True, timeout=300)
print("alone:", g.stdout.strip(), flush=True)
for frames in (12, 4):
    g = subprocess.Popen([sys.executable, "-c", GEMM, "14"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(6)            # torch import + warm-up
    p = scan(frames, 8); time.sleep(8); r = rate(p, 8)
    out, _ = g.communicate(timeout=300)
    print(f"scan with {frames} frames beside the GEMM loop: scan {r:.2f} Gkeys/s (part of the time alone), {out.strip()} (over 14 s, 8 of them shared)", flush=True)
