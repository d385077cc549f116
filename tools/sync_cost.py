"""What does the contract's closing torch.cuda.synchronize() cost when nothing is pending (12 frame streams alive)?"""
import sys, time
sys.path.insert(0, ".")
import torch
import vgen_amd as v
N = 1 << 20
r = v.GpuRunner(batch_size=N, fmt=v.AddressFormat.P2pkh, frames=12, timing=False)
r.set_filter(v.Pattern("^1Cat", False, v.AddressFormat.P2pkh))
key = 0x3a8ae174e51b7b1117ab406c6570970f453c4376b6d381977db7c02fb5a993e0
torch.cuda.synchronize()
for rep in range(5):
    for f in range(12):
        r.dispatch(key, f); key += N
    for f in range(12):
        r.wait(f)
    t = []
    for i in range(5):
        t0 = time.perf_counter(); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e6)
    print("torch.cuda.synchronize() with nothing pending: " + " ".join("%.0f" % x for x in t) + " us")
