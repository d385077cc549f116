"""ctypes view of the parity oracle (oracle/libvgen_oracle.so) — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
vgen_amd package.  See oracle/vgen_oracle.h for what each entry point restates.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvgen_oracle.so")

FMT_P2PKH, FMT_P2WPKH, FMT_P2SH_P2WPKH, FMT_P2TR, FMT_P2PKH_UNCOMPRESSED, FMT_ETHEREUM = range(6)
FORMAT_NAMES = {
    "p2pkh": FMT_P2PKH, "p2wpkh": FMT_P2WPKH, "p2sh_p2wpkh": FMT_P2SH_P2WPKH, "p2tr": FMT_P2TR,
    "p2pkh_uncompressed": FMT_P2PKH_UNCOMPRESSED, "ethereum": FMT_ETHEREUM,
}


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


class Generated(ctypes.Structure):
    _fields_ = [("address", ctypes.c_char * 96), ("wif", ctypes.c_char * 72),
                ("hex", ctypes.c_char * 72), ("format", ctypes.c_int)]


class Match(ctypes.Structure):
    _fields_ = [("key", ctypes.c_uint8 * 32), ("gen", Generated)]


class ScanResult(ctypes.Structure):
    _fields_ = [("matches", ctypes.POINTER(Match)), ("n_matches", ctypes.c_size_t),
                ("operations", ctypes.c_uint64), ("elapsed_secs", ctypes.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.vo_regex_new.restype = ctypes.c_void_p
        L.vo_regex_new.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        L.vo_regex_is_match.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.vo_regex_free.argtypes = [ctypes.c_void_p]
        L.vo_scan_range.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p,
                                    ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int,
                                    ctypes.POINTER(ScanResult)]
        L.vo_scan_random.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint64,
                                     ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int,
                                     ctypes.POINTER(ScanResult)]
        L.vo_seed_key.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_char_p]
        L.vo_payload_seq.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int,
                                     ctypes.c_void_p]
        L.vo_key_add_u64.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p]
        _lib = L
    return _lib


def _key(k):
    return k.to_bytes(32, "big") if isinstance(k, int) else bytes(k)


def sha256(b):
    out = ctypes.create_string_buffer(32)
    lib().vo_sha256(bytes(b), len(b), out)
    return out.raw


def ripemd160(b):
    out = ctypes.create_string_buffer(20)
    lib().vo_ripemd160(bytes(b), len(b), out)
    return out.raw


def keccak256(b):
    out = ctypes.create_string_buffer(32)
    lib().vo_keccak256(bytes(b), len(b), out)
    return out.raw


def hash160(b):
    out = ctypes.create_string_buffer(20)
    lib().vo_hash160(bytes(b), len(b), out)
    return out.raw


def key_valid(k):
    return bool(lib().vo_key_valid(_key(k)))


def pubkey(k, naive=False):
    out = ctypes.create_string_buffer(65)
    fn = lib().vo_pubkey_naive if naive else lib().vo_pubkey
    return out.raw if fn(_key(k), out) else None


def generate(fmt, k):
    g = Generated()
    if not lib().vo_generate(fmt, _key(k), ctypes.byref(g)):
        return None
    return {"address": g.address.decode(), "wif": g.wif.decode(), "hex": g.hex.decode(), "format": g.format}


def payload(fmt, k):
    out = ctypes.create_string_buffer(32)
    n = lib().vo_payload(fmt, _key(k), out)
    return out.raw[:n] if n else None


def payload_seq(fmt, start, n, threads=0):
    """bytes of n payloads (20 B each, 32 for P2TR) for keys start .. start+n-1."""
    plen = 32 if fmt == FMT_P2TR else 20
    buf = ctypes.create_string_buffer(n * plen)
    lib().vo_payload_seq(fmt, _key(start), n, threads, buf)
    return buf.raw


def address_from_hash160(fmt, h160):
    out = ctypes.create_string_buffer(96)
    n = lib().vo_address_from_hash160(fmt, bytes(h160), out)
    return out.value.decode() if n > 0 else None


def segwit_addr(hrp, witver, prog):
    out = ctypes.create_string_buffer(128)
    n = lib().vo_segwit_addr(hrp.encode(), witver, bytes(prog), len(prog), out, 128)
    return out.value.decode() if n > 0 else None


def eip55(addr20):
    out = ctypes.create_string_buffer(43)
    lib().vo_eip55(bytes(addr20), out)
    return out.value.decode()


def wif(k, compressed=True):
    out = ctypes.create_string_buffer(64)
    lib().vo_wif(_key(k), 1 if compressed else 0, out)
    return out.value.decode()


def lift_x(x):
    out = ctypes.create_string_buffer(65)
    return out.raw if lib().vo_lift_x(bytes(x), out) else None


def taproot_output_key(pub65):
    out = ctypes.create_string_buffer(32)
    return out.raw if lib().vo_taproot_output_key(bytes(pub65), out) else None


def sha256_midstate(block64):
    st = (ctypes.c_uint32 * 8)()
    lib().vo_sha256_midstate(bytes(block64), st)
    return list(st)


def seed_key(seed, shard=0):
    out = ctypes.create_string_buffer(32)
    lib().vo_seed_key(seed, shard, out)
    return int.from_bytes(out.raw, "big")


def random_key(seed, stream, index):
    """The oracle's counter-based candidate stream (vo_scan.c: random_key, the keys vo_scan_random's worker `stream`
    walks): SHA-256("vgen-mi355x-rand" || seed[24] || u32le(stream) || u64le(index)) as a big-endian integer; seed: the 24
    bytes themselves, or an integer < 2^64 standing for u64le(seed) || sixteen zero bytes (the C ABI's 64-bit seeds).
    Restated here with hashlib, independently of both C implementations.  Invalid draws (0, >= n) yield no key."""
    import hashlib
    sb = bytes(seed) if isinstance(seed, (bytes, bytearray)) else seed.to_bytes(8, "little") + bytes(16)
    assert len(sb) == 24
    return int.from_bytes(hashlib.sha256(b"vgen-mi355x-rand" + sb + stream.to_bytes(4, "little")
                                         + index.to_bytes(8, "little")).digest(), "big")


class Regex:
    """Pattern::new / Pattern::matches (reference src/pattern.rs:21-45)."""

    def __init__(self, pattern, case_insensitive=False):
        err = ctypes.create_string_buffer(256)
        self._h = lib().vo_regex_new(pattern.encode(), 1 if case_insensitive else 0, err, 256)
        if not self._h:
            raise ValueError(err.value.decode() or "invalid pattern")

    def matches(self, text):
        return bool(lib().vo_regex_is_match(self._h, text.encode()))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vo_regex_free(self._h)
            self._h = None


def _collect(res):
    out = []
    for i in range(res.n_matches):
        m = res.matches[i]
        out.append({"key": int.from_bytes(bytes(m.key), "big"), "address": m.gen.address.decode(),
                    "wif": m.gen.wif.decode(), "hex": m.gen.hex.decode()})
    return out


def scan_range(fmt, pattern, start, end, count=1, ci=False, threads=0):
    res = ScanResult()
    rc = lib().vo_scan_range(fmt, pattern.encode(), int(ci), _key(start), _key(end), count, threads,
                             ctypes.byref(res))
    if rc != 0:
        raise ValueError(f"vo_scan_range failed ({rc})")
    out = {"matches": _collect(res), "operations": res.operations, "elapsed_secs": res.elapsed_secs}
    lib().vo_scan_free(ctypes.byref(res))
    return out


def scan_random(fmt, pattern, seed, count=1, max_keys=0, ci=False, threads=0):
    res = ScanResult()
    rc = lib().vo_scan_random(fmt, pattern.encode(), int(ci), seed, count, max_keys, threads,
                              ctypes.byref(res))
    if rc != 0:
        raise ValueError(f"vo_scan_random failed ({rc})")
    out = {"matches": _collect(res), "operations": res.operations, "elapsed_secs": res.elapsed_secs}
    lib().vo_scan_free(ctypes.byref(res))
    return out
