#!/bin/bash
# Where a CLI run's wall time goes: HIP API timeline (AMD_LOG_LEVEL=3 prints a microsecond timestamp per call), largest gaps.
V=vgen_amd/vgen-hip
t0=$(date +%s.%N); $V generate -p "^1Cat" -o minimal > /dev/null; t1=$(date +%s.%N)
python3 -c "print(f'plain run wall {($t1-$t0)*1e3:.0f} ms')"
t0=$(date +%s.%N); LD_BIND_NOW=1 $V generate -p "^1Cat" -o minimal > /dev/null; t1=$(date +%s.%N)
python3 -c "print(f'LD_BIND_NOW wall {($t1-$t0)*1e3:.0f} ms')"
t0=$(date +%s.%N); $V list-gpus > /dev/null; t1=$(date +%s.%N)
python3 -c "print(f'list-gpus wall {($t1-$t0)*1e3:.0f} ms')"
t0=$(date +%s.%N); $V verify -k 01 > /dev/null 2>&1; t1=$(date +%s.%N)
python3 -c "print(f'verify (no HIP call) wall {($t1-$t0)*1e3:.0f} ms')"
AMD_LOG_LEVEL=3 $V generate -p "^1Cat" -o minimal > /dev/null 2> /tmp/hiplog.txt
python3 - <<'PY'
import re
rows=[]
for ln in open('/tmp/hiplog.txt', errors='replace'):
    m=re.search(r'\[(\d+)us\]|:(\d+)\s*us', ln)
    m2=re.match(r':\d+:[^:]+:\s*(\d+)\s*:\s*(\d+)\s*us', ln)
    if m2:
        rows.append((int(m2.group(2)), ln.strip()[:150]))
print(len(rows), "timestamped lines")
if rows:
    base=rows[0][0]
    gaps=sorted(((rows[i+1][0]-rows[i][0], i) for i in range(len(rows)-1)), reverse=True)[:14]
    for g,i in sorted(gaps, key=lambda x: x[1]):
        print(f"+{(rows[i][0]-base)/1e3:8.1f} ms  gap {g/1e3:7.1f} ms after: {rows[i][1][:120]}")
    print(f"last line at +{(rows[-1][0]-base)/1e3:.1f} ms")
PY
head -5 /tmp/hiplog.txt
