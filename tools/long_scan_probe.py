"""A long scan through the CLI, ended by Ctrl-C: sustained rate, resident memory, graceful stop.
usage: python tools/long_scan_probe.py [seconds]"""
import os, resource, signal, subprocess, sys, time
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vgen_amd", "vgen-hip")
t0 = time.time()
p = subprocess.Popen([exe, "range", "-p", "boha:b1000:66", "-l", "12", "-o", "json"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
rss = []
while time.time() - t0 < secs:
    time.sleep(10)
    try:
        for ln in open(f"/proc/{p.pid}/status"):
            if ln.startswith("VmRSS"):
                rss.append(int(ln.split()[1]) // 1024)
    except OSError:
        break
p.send_signal(signal.SIGINT)
out, err = p.communicate(timeout=60)
print("exit code", p.returncode, "after", round(time.time() - t0, 1), "s; stopped", round(time.time() - t0 - secs, 2), "s after SIGINT")
print("resident MiB every 10 s:", rss)
print("stdout:", out.strip()[:200])
print("stderr:", err.strip()[-300:])
print("peak RSS of the child (MiB):", resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss // 1024)
