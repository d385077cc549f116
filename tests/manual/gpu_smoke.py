"""Quick GPU parity probe used during bring-up (the formal tests live in tests/test_gpu_*.py)."""
import sys, time
sys.path.insert(0, ".")
import vgen_amd as v
from oracle import pyoracle as vo

fmt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
print("devices:", v.device_count(), v.device_name(0))
t0 = time.time()
r = v.GpuRunner(batch_size=batch, fmt=v.AddressFormat(fmt), frames=2)
print("create %.3fs batch %d" % (time.time() - t0, r.batch_size))
for start in (vo.seed_key(42, 0), 1, 2**65, 0xFFFFFFFFFFFF):
    r.dispatch(start, 0)
    blob, _, tested = r.await_result(0)
    ref = vo.payload_seq(fmt, start, batch)
    bad = [i for i in range(batch) if blob[20*i:20*i+20] != ref[20*i:20*i+20]]
    print("start %x: kernel %.3f ms, mismatches %d / %d" % (start, r.kernel_ms(0), len(bad), batch), bad[:8])
    if bad:
        i = bad[0]; print(i, blob[20*i:20*i+20].hex(), ref[20*i:20*i+20].hex())
