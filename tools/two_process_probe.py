"""Several independent CLI scans on ONE GPU at once: what each gets (12 frames each = 24 / 36 hardware queues on the device).
usage: python tools/two_process_probe.py"""
import os, re, signal, subprocess, time
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vgen_amd", "vgen-hip")
def run(n_proc, frames, secs=8.0):
    ps = [subprocess.Popen([exe, "range", "-p", "boha:b1000:66", "-l", "12", "-o", "json", "--frames", str(frames)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(n_proc)]
    time.sleep(secs)
    for p in ps: p.send_signal(signal.SIGINT)
    rates = []
    for p in ps:
        out, err = p.communicate(timeout=120)
        m = re.search(r"No match found after ([\d,]+) operations", err)
        rates.append(int(m.group(1).replace(",", "")) / secs / 1e9 if m else None)
    print(f"{n_proc} process(es) x {frames} frames: " + ", ".join("%.2f" % r if r else "?" for r in rates) + " Gkeys/s each"
          + (f" (sum {sum(r for r in rates if r):.2f})" if n_proc > 1 else ""), flush=True)
import sys
if len(sys.argv) > 1:      # explicit cases: PROCESSESxFRAMES ...
    for a in sys.argv[1:]:
        n, f = a.split('x')
        run(int(n), int(f))
else:
    run(1, 12)
    for fr in (12, 6, 4): run(2, fr)
    run(3, 4)
    run(4, 3)
