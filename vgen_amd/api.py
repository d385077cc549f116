"""ctypes binding of include/vgen_hip.h, shaped like the reference's Rust API.

  AddressFormat        src/address.rs:11-24
  GeneratedAddress     src/address.rs:63-72
  Pattern              src/pattern.rs:9-45         (new / matches)
  ScanConfig/Result    src/scanner.rs:17-68
  GpuRunner            src/gpu.rs:116-131,138,535,602   (new / dispatch / await_result)
  scan_gpu_with_runner src/gpu.rs:920-926

The extension is loaded from vgen_amd/libvgen_hip.so (in-tree).  If it is missing the import of
this module fails loudly — nothing here computes an address in Python.
"""
import ctypes
import enum
import os
from dataclasses import dataclass, field
from typing import Callable, List, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# (_SO_OVERRIDE: a second instance of this module bound to the TEST build of the library — the same sources with the
#  fault-injection hooks compiled in —, pre-set by tests/conftest.py: hooks_api() before it executes the module; never an
#  environment variable, never set in the package itself)
_SO = globals().get("_SO_OVERRIDE") or os.path.join(_HERE, "libvgen_hip.so")


class VgenError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"[vgen status {status}] {message}")
        self.status = status


def library_path():
    return _SO


if not os.path.exists(_SO):
    raise ImportError(f"{_SO} not found: build it with `make -C vgen_amd/csrc` (or __graft_entry__.build()); "
                      "vgen_amd has no pure-Python or CPU fallback")
_L = ctypes.CDLL(_SO)

OK, E_INVALID, E_NODEVICE, E_HIP, E_NOMEM, E_STATE, E_PATTERN, E_RANGE, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7, -8


class AddressFormat(enum.IntEnum):
    P2pkh = 0
    P2wpkh = 1
    P2shP2wpkh = 2
    P2tr = 3
    P2pkhUncompressed = 4
    Ethereum = 5

    def charset_name(self) -> str:
        """AddressFormat::charset_name (src/address.rs:39-45)."""
        return _L.vgen_format_charset_name(int(self)).decode()


class _Params(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("batch_size", ctypes.c_uint32),
                ("format", ctypes.c_uint32), ("frames", ctypes.c_uint32), ("match_cap", ctypes.c_uint32),
                ("flags", ctypes.c_uint32), ("table_bits", ctypes.c_uint32), ("device_mem_budget_bytes", ctypes.c_uint64)]


class _MemoryInfo(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("table_bits", ctypes.c_uint32), ("frames_bytes", ctypes.c_uint64),
                ("mode_bytes", ctypes.c_uint64), ("table_bytes", ctypes.c_uint64), ("pinned_host_bytes", ctypes.c_uint64),
                ("budget_bytes", ctypes.c_uint64), ("device_free_bytes", ctypes.c_uint64), ("device_total_bytes", ctypes.c_uint64)]


class _Match(ctypes.Structure):
    _fields_ = [("index", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("payload", ctypes.c_uint8 * 32)]


class _ScanConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("format", ctypes.c_uint32), ("count", ctypes.c_uint64),
                ("case_insensitive", ctypes.c_int32), ("has_start", ctypes.c_int32), ("start", ctypes.c_uint8 * 32),
                ("has_end", ctypes.c_int32), ("end", ctypes.c_uint8 * 32), ("seed", ctypes.c_uint64),
                ("shard", ctypes.c_uint32), ("n_shards", ctypes.c_uint32), ("max_batches", ctypes.c_uint64),
                ("checkpoint_path", ctypes.c_char_p), ("checkpoint_interval_ms", ctypes.c_uint32),
                ("flags", ctypes.c_uint32), ("table_bits_max", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


SCAN_RANDOM_KEYS = 1   # VGEN_SCAN_RANDOM_KEYS


class _Generated(ctypes.Structure):
    _fields_ = [("address", ctypes.c_char * 96), ("wif", ctypes.c_char * 72), ("hex", ctypes.c_char * 72),
                ("format", ctypes.c_uint32), ("key", ctypes.c_uint8 * 32)]


class _ScanResult(ctypes.Structure):
    _fields_ = [("matches", ctypes.POINTER(_Generated)), ("n_matches", ctypes.c_uint64),
                ("operations", ctypes.c_uint64), ("elapsed_secs", ctypes.c_double),
                ("resumed_operations", ctypes.c_uint64), ("complete", ctypes.c_int32), ("failed_shards", ctypes.c_int32)]


_PROGRESS = ctypes.CFUNCTYPE(None, ctypes.c_uint64, ctypes.c_void_p)

_L.vgen_last_error.restype = ctypes.c_char_p
_L.vgen_last_error.argtypes = [ctypes.c_void_p]
_L.vgen_create.argtypes = [ctypes.POINTER(_Params), ctypes.POINTER(ctypes.c_void_p)]
_L.vgen_destroy.argtypes = [ctypes.c_void_p]
_L.vgen_destroy.restype = None
_L.vgen_get_info.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_uint32)] * 3
_L.vgen_filter_compile.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
_L.vgen_filter_free.argtypes = [ctypes.c_void_p]
_L.vgen_filter_free.restype = None
_L.vgen_filter_matches.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
_L.vgen_filter_device_kind.argtypes = [ctypes.c_void_p]
_L.vgen_filter_dfa_bytes.argtypes = [ctypes.c_void_p]
_L.vgen_pattern_invalid_chars.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t,
                                          ctypes.POINTER(ctypes.c_size_t)]
_L.vgen_pattern_difficulty.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
_L.vgen_format_charset_name.argtypes = [ctypes.c_uint32]
_L.vgen_format_charset_name.restype = ctypes.c_char_p
_L.vgen_provider_resolve.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                                     ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32), ctypes.c_char_p,
                                     ctypes.c_char_p]
_L.vgen_provider_build_pattern.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
_L.vgen_set_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
_L.vgen_clock_probe_start.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
_L.vgen_clock_probe_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
_L.vgen_dispatch.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p]
_L.vgen_dispatch_keys.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32]
if hasattr(_L, "vgen_debug_fail_after"):   # only the test build (tests/native/libvgen_hip_hooks.so) has it
    _L.vgen_debug_fail_after.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
_L.vgen_dispatch_random_seed.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint64]
_L.vgen_random_key_seed.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_char_p]
_L.vgen_get_resources.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_uint32)] * 3 + [ctypes.c_char_p, ctypes.c_size_t]
_L.vgen_get_memory.argtypes = [ctypes.c_void_p, ctypes.POINTER(_MemoryInfo)]
_L.vgen_dispatch_random.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64]
_L.vgen_random_key.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_char_p]
_L.vgen_wait.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(_Match), ctypes.c_uint32,
                         ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)]
_L.vgen_read_dump.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
_L.vgen_dump_view.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
_L.vgen_get_topology.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_uint32)] * 3 + [ctypes.POINTER(ctypes.c_int32)]   # streams, hw_queues, priority_levels, oversubscribed
_L.vgen_frame_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float)]
_L.vgen_frame_clock.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
_L.vgen_frame_dispatch_ms.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float)]
_L.vgen_address_from_payload.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
_L.vgen_key_to_wif.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
_L.vgen_key_variant.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p]
_L.vgen_key_add.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p]
_L.vgen_derive.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                           ctypes.c_size_t]
_L.vgen_device_name.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
_L.vgen_scan.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(_ScanConfig), _PROGRESS, ctypes.c_void_p,
                         ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(_ScanResult)]
_L.vgen_scan_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_char_p,
                               ctypes.POINTER(_ScanConfig), _PROGRESS, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32),
                               ctypes.POINTER(_ScanResult)]
_L.vgen_scan_result_free.argtypes = [ctypes.POINTER(_ScanResult)]
_L.vgen_scan_result_free.restype = None


def _key(k):
    return k.to_bytes(32, "big") if isinstance(k, int) else bytes(k)


def _check(rc, ctx=None):
    if rc < 0:
        msg = _L.vgen_last_error(ctx)
        raise VgenError(rc, msg.decode() if msg else "")
    return rc


def abi_version():
    return _L.vgen_abi_version()


def device_count():
    n = ctypes.c_int(0)
    _check(_L.vgen_device_count(ctypes.byref(n)))
    return n.value


def device_name(device=0):
    buf = ctypes.create_string_buffer(256)
    _check(_L.vgen_device_name(device, buf, 256))
    return buf.value.decode()


def address_from_payload(fmt, payload):
    out = ctypes.create_string_buffer(128)
    _check(_L.vgen_address_from_payload(int(fmt), bytes(payload), out, 128))
    return out.value.decode()


def key_to_wif(fmt, key):
    out = ctypes.create_string_buffer(128)
    _check(_L.vgen_key_to_wif(int(fmt), _key(key), out, 128))
    return out.value.decode()


def key_add(key, amount):
    """increment_key (src/gpu.rs:951-968): None on overflow or an invalid scalar."""
    out = ctypes.create_string_buffer(32)
    rc = _L.vgen_key_add(_key(key), amount, out)
    return None if rc == E_RANGE else int.from_bytes(out.raw, "big")


def random_key(seed, stream, index):
    """vgen_random_key: candidate `index` of stream `stream` under `seed` of the counter-based scalar stream
    (vgen_dispatch_random); None when that draw is not a valid scalar.  seed: an integer < 2^64 (vgen_random_key) or the 24
    seed bytes themselves (vgen_random_key_seed)."""
    out = ctypes.create_string_buffer(32)
    if isinstance(seed, (bytes, bytearray)):
        assert len(seed) == 24
        rc = _L.vgen_random_key_seed(bytes(seed), stream, index, out)
    else:
        rc = _L.vgen_random_key(seed, stream, index, out)
    if rc == -7:
        return None
    _check(rc)
    return int.from_bytes(out.raw, "big")


def key_variant(key, variant):
    """vgen_key_variant: the key an endomorphism context tests as `variant` of base key `key`."""
    out = ctypes.create_string_buffer(32)
    _check(_L.vgen_key_variant(_key(key), variant, out))
    return int.from_bytes(out.raw, "big")


@dataclass
class GeneratedAddress:
    address: str
    wif: str
    hex: str
    format: AddressFormat


def derive(fmt, key) -> Optional[GeneratedAddress]:
    """AddressGenerator::generate on the host (src/address.rs:92-151); None for an invalid key."""
    a, w = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128)
    rc = _L.vgen_derive(int(fmt), _key(key), a, 128, w, 128)
    if rc == E_RANGE:
        return None
    _check(rc)
    return GeneratedAddress(a.value.decode(), w.value.decode(), _key(key).hex(), AddressFormat(int(fmt)))


class Pattern:
    """Pattern::new / matches (src/pattern.rs:21-45) plus the device prefilter derived for `fmt`."""

    def __init__(self, pattern: str, case_insensitive: bool = False, fmt: AddressFormat = AddressFormat.P2pkh):
        self.original, self.case_insensitive, self.format = pattern, case_insensitive, AddressFormat(int(fmt))
        h = ctypes.c_void_p()
        _check(_L.vgen_filter_compile(pattern.encode(), int(case_insensitive), int(fmt), ctypes.byref(h)))
        self._h = h

    def matches(self, address: str) -> bool:
        return _L.vgen_filter_matches(self._h, address.encode()) == 1

    @property
    def device_kind(self) -> int:
        return _L.vgen_filter_device_kind(self._h)

    @property
    def dfa_bytes(self) -> int:
        """vgen_filter_dfa_bytes: LDS bytes of the on-device automaton (device_kind 4), else 0."""
        return _L.vgen_filter_dfa_bytes(self._h)

    def validate_charset(self, fmt: Optional[AddressFormat] = None) -> List[str]:
        """Pattern::validate_charset (src/pattern.rs:49-177): characters that can never occur in `fmt` addresses."""
        fmt = self.format if fmt is None else fmt
        buf, n = ctypes.create_string_buffer(260), ctypes.c_size_t()
        _check(_L.vgen_pattern_invalid_chars(self.original.encode(), int(self.case_insensitive), int(fmt), buf, 260,
                                             ctypes.byref(n)))
        return list(buf.value.decode())

    def estimate_difficulty(self, fmt: Optional[AddressFormat] = None) -> int:
        """Pattern::estimate_difficulty (src/pattern.rs:183-253): 1 in N addresses is expected to match."""
        fmt = self.format if fmt is None else fmt
        d = ctypes.c_uint64()
        _check(_L.vgen_pattern_difficulty(self.original.encode(), int(self.case_insensitive), int(fmt), ctypes.byref(d)))
        return d.value

    def is_case_insensitive(self) -> bool:
        return self.case_insensitive

    def __del__(self):
        if getattr(self, "_h", None):
            _L.vgen_filter_free(self._h)
            self._h = None


@dataclass
class ProviderResult:
    """ProviderResult (src/provider.rs:6-10)."""
    address: str
    format: AddressFormat
    key_range: Optional[tuple] = None


def provider_resolve(pattern: str, table_path: Optional[str] = None) -> Optional[ProviderResult]:
    """provider::resolve (src/provider.rs:12-52): None when `pattern` is an ordinary regex."""
    addr, fmt, has = ctypes.create_string_buffer(128), ctypes.c_uint32(), ctypes.c_int32()
    lo, hi = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    rc = _L.vgen_provider_resolve(pattern.encode(), os.fsencode(table_path) if table_path else None, addr, 128,
                                  ctypes.byref(fmt), ctypes.byref(has), lo, hi)
    if rc == 0:
        return None
    _check(rc if rc < 0 else 0)
    rng = (int.from_bytes(lo.raw, "big"), int.from_bytes(hi.raw, "big")) if has.value else None
    return ProviderResult(addr.value.decode(), AddressFormat(fmt.value), rng)


def build_pattern(result: ProviderResult, prefix_length: int) -> str:
    """provider::build_pattern (src/provider.rs:54-58)."""
    if prefix_length < 1:
        raise ValueError("--prefix-length must be at least 1 for provider patterns")   # lib.rs:570-572
    out = ctypes.create_string_buffer(300)
    _check(_L.vgen_provider_build_pattern(result.address.encode(), prefix_length, out, 300))
    return out.value.decode()


def build_exact_pattern(result: ProviderResult) -> str:
    """provider::build_exact_pattern (src/provider.rs:60-62)."""
    out = ctypes.create_string_buffer(300)
    _check(_L.vgen_provider_build_pattern(result.address.encode(), 0, out, 300))
    return out.value.decode()


@dataclass
class ScanConfig:
    """src/scanner.rs:17-46 (fields the GPU path reads) + the build-side seed / shard additions."""
    format: AddressFormat = AddressFormat.P2pkh
    count: int = 1
    gpu_batch_size: Optional[int] = None
    start: Optional[int] = None
    end: Optional[int] = None
    case_insensitive: bool = False
    seed: int = 0
    shard: int = 0
    n_shards: int = 1
    max_batches: int = 0
    checkpoint_path: Optional[str] = None      # resumable scans (vgen_scan_config.checkpoint_path)
    checkpoint_interval_ms: int = 0
    random_keys: bool = False       # VGEN_SCAN_RANDOM_KEYS: an independent random key per candidate (scanner.rs:118-169's shape)
    table_bits_max: int = 0         # widest generator table the scan may move the context to (0 = the context's memory policy decides)


@dataclass
class ScanResult:
    matches: List[GeneratedAddress] = field(default_factory=list)
    operations: int = 0
    elapsed_secs: float = 0.0
    resumed_operations: int = 0     # operations recorded in the checkpoint this call resumed from
    complete: bool = False          # the key range ran out
    failed_shards: int = 0          # contexts that failed during the scan (their stripes were taken over by the others)

    def rate(self) -> float:   # src/scanner.rs:61-67
        return self.operations / self.elapsed_secs if self.elapsed_secs > 0 else 0.0


class GpuRunner:
    """GpuRunner (src/gpu.rs:116-131): one device, `frames` dispatches in flight."""

    def __init__(self, batch_size: int = 1 << 20, fmt: AddressFormat = AddressFormat.P2pkh, device: int = 0,
                 frames: int = 2, match_cap: int = 4096, timing: bool = True, endo: bool = False, table_bits: int = 0,
                 device_mem_budget_bytes: int = 0):
        # timing: VGEN_FLAG_TIMING — events around every dispatch so that kernel_ms() / dispatch_ms() work
        # endo: VGEN_FLAG_ENDO — six keys per curve point (vanity searches on compressed-key formats)
        # table_bits / device_mem_budget_bytes: the generator-table width and the device-memory bound of this context (0 = automatic)
        p = _Params(ctypes.sizeof(_Params), device, batch_size, int(fmt), frames, match_cap, (1 if timing else 0) | (2 if endo else 0),
                    table_bits, device_mem_budget_bytes)
        h = ctypes.c_void_p()
        _check(_L.vgen_create(ctypes.byref(p), ctypes.byref(h)))
        self._h = h
        b, f, m = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        _L.vgen_get_info(h, ctypes.byref(b), ctypes.byref(f), ctypes.byref(m))
        self.batch_size, self.frames, self.match_cap, self.format = b.value, f.value, m.value, AddressFormat(int(fmt))
        self.payload_bytes = 32 if self.format == AddressFormat.P2tr else 20
        self._pattern = None

    def topology(self) -> dict:
        """vgen_get_topology: streams of the context (one per frame), the HIP hardware-queue limit per stream priority
        level, the number of levels, and whether every stream owns a queue."""
        a, q, l, o = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_int32()
        _check(_L.vgen_get_topology(self._h, ctypes.byref(a), ctypes.byref(q), ctypes.byref(l), ctypes.byref(o)), self._h)
        return {"streams": a.value, "hw_queues": q.value, "priority_levels": l.value, "oversubscribed": bool(o.value)}

    def dump_view(self, frame: int) -> bytes:
        """vgen_dump_view: the frame's payloads from the pinned buffer its dump-mode dispatch copied itself into."""
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(_L.vgen_dump_view(self._h, frame, ctypes.byref(ptr), ctypes.byref(n)), self._h)
        return ctypes.string_at(ptr.value, n.value)

    def close(self):
        if getattr(self, "_h", None):
            _L.vgen_destroy(self._h)
            self._h = None

    __del__ = close

    def set_filter(self, pattern: Optional[Pattern]):
        _check(_L.vgen_set_filter(self._h, pattern._h if pattern else None), self._h)
        self._pattern = pattern

    def dispatch(self, start_key, frame: int):
        _check(_L.vgen_dispatch(self._h, frame, _key(start_key)), self._h)

    def dispatch_keys(self, keys, frame: int):
        """vgen_dispatch_keys: arbitrary scalars (ints, 32-byte strings, or one n*32-byte blob), full k*G per key."""
        if isinstance(keys, (bytes, bytearray)):     # already n * 32 big-endian bytes
            blob, n = bytes(keys), len(keys) // 32
        else:
            blob, n = b"".join(_key(k) for k in keys), len(keys)
        self._n_keys = n
        _check(_L.vgen_dispatch_keys(self._h, frame, blob, n), self._h)

    def fail_after(self, dispatches: int):
        """vgen_debug_fail_after: fault injection — dispatches fail with VGEN_E_HIP once `dispatches` more were accepted.
        Exists only in the test build of the library (tests/conftest.py: hooks_api)."""
        if not hasattr(_L, "vgen_debug_fail_after"):
            raise RuntimeError("fault injection is not part of libvgen_hip.so: load tests/native/libvgen_hip_hooks.so (tests/conftest.py: hooks_api)")
        _check(_L.vgen_debug_fail_after(self._h, dispatches), self._h)

    def resources(self):
        """vgen_get_resources -> dict(dump_frames, table_bits, table_bits_wanted, note)."""
        d, b, w = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        note = ctypes.create_string_buffer(256)
        _check(_L.vgen_get_resources(self._h, ctypes.byref(d), ctypes.byref(b), ctypes.byref(w), note, 256), self._h)
        return {"dump_frames": d.value, "table_bits": b.value, "table_bits_wanted": w.value, "note": note.value.decode()}

    def memory(self):
        """vgen_get_memory -> dict of the context's device / pinned host bytes, its budget and the device's free / total bytes."""
        m = _MemoryInfo()
        m.struct_size = ctypes.sizeof(_MemoryInfo)
        _check(_L.vgen_get_memory(self._h, ctypes.byref(m)), self._h)
        return {k: getattr(m, k) for k, _ in _MemoryInfo._fields_ if k != "struct_size"}

    def dispatch_random(self, seed, stream: int, first_index: int, frame: int):
        """vgen_dispatch_random: batch_size independent random keys drawn on the device from the counter-based stream.
        seed: an integer < 2^64 or the 24 seed bytes (vgen_dispatch_random_seed)."""
        self._n_keys = self.batch_size
        if isinstance(seed, (bytes, bytearray)):
            assert len(seed) == 24
            _check(_L.vgen_dispatch_random_seed(self._h, frame, bytes(seed), stream, first_index), self._h)
        else:
            _check(_L.vgen_dispatch_random(self._h, frame, seed, stream, first_index), self._h)

    def await_result(self, frame: int):
        """Filter mode: (list of (index, payload20), n_found, keys_tested).  Dump mode: (bytes, 0, keys_tested)."""
        recs = (_Match * self.match_cap)()
        n, tested = ctypes.c_uint32(), ctypes.c_uint64()
        _check(_L.vgen_wait(self._h, frame, recs, self.match_cap, ctypes.byref(n), ctypes.byref(tested)), self._h)
        if self._pattern is None:
            return self.dump_view(frame), 0, tested.value
        k = min(n.value, self.match_cap)
        return [(recs[i].index, bytes(recs[i].payload)[:self.payload_bytes]) for i in range(k)], n.value, tested.value

    def wait(self, frame: int):
        """vgen_wait without fetching anything (benchmark loop)."""
        n, tested = ctypes.c_uint32(), ctypes.c_uint64()
        _check(_L.vgen_wait(self._h, frame, None, 0, ctypes.byref(n), ctypes.byref(tested)), self._h)
        return n.value, tested.value

    def clock_probe_start(self, duration_ms: int):
        """Samples the shader clock for duration_ms on a side stream (vgen_clock_probe_start)."""
        _check(_L.vgen_clock_probe_start(self._h, duration_ms), self._h)

    def clock_probe_read(self) -> float:
        mhz = ctypes.c_double()
        _check(_L.vgen_clock_probe_read(self._h, ctypes.byref(mhz)), self._h)
        return mhz.value

    def kernel_ms(self, frame: int) -> float:
        """HIP-event duration of the dominant kernel (seq_bwd_kernel) of the frame's last dispatch."""
        ms = ctypes.c_float()
        _check(_L.vgen_frame_kernel_ms(self._h, frame, ctypes.byref(ms)), self._h)
        return ms.value

    def frame_clock(self, frame: int):
        """vgen_frame_clock: (shader-clock cycles, 100 MHz ticks) sampled by the frame's last seq_bwd launch."""
        c, t = ctypes.c_uint32(), ctypes.c_uint32()
        _check(_L.vgen_frame_clock(self._h, frame, ctypes.byref(c), ctypes.byref(t)), self._h)
        return c.value, t.value

    def dispatch_ms(self, frame: int) -> float:
        """HIP-event duration of the whole last dispatch (seq_fwd incl. the root inversions + seq_bwd)."""
        ms = ctypes.c_float()
        _check(_L.vgen_frame_dispatch_ms(self._h, frame, ctypes.byref(ms)), self._h)
        return ms.value


def scan_gpu_with_runner(pattern: str, config: ScanConfig, runner,
                         progress_cb: Optional[Callable[[int], None]] = None, stop=None, force_multi: bool = False) -> ScanResult:
    """scan_gpu_with_runner (src/gpu.rs:920-926).  `stop` is an optional ctypes.c_int32 flag.
    `runner` may be a list of GpuRunners (one per GPU): the batches are then striped over them
    (vgen_scan_multi; force_multi: also for a list of one)."""
    runners = list(runner) if isinstance(runner, (list, tuple)) else [runner]
    runner = runners[0]
    c = _ScanConfig()
    c.struct_size = ctypes.sizeof(_ScanConfig)
    c.format = int(config.format)
    c.count = config.count if config.count is not None else 2**64 - 1
    c.case_insensitive = int(config.case_insensitive)
    if config.start is not None:
        c.has_start = 1
        c.start = (ctypes.c_uint8 * 32)(*_key(config.start))
    if config.end is not None:
        c.has_end = 1
        c.end = (ctypes.c_uint8 * 32)(*_key(config.end))
    c.seed, c.shard, c.n_shards, c.max_batches = config.seed, config.shard, config.n_shards, config.max_batches
    if config.checkpoint_path:
        c.checkpoint_path = os.fsencode(config.checkpoint_path)
        c.checkpoint_interval_ms = config.checkpoint_interval_ms
    c.flags = SCAN_RANDOM_KEYS if config.random_keys else 0
    c.table_bits_max = config.table_bits_max
    res = _ScanResult()
    cb = _PROGRESS(lambda ops, _u: progress_cb(ops)) if progress_cb else ctypes.cast(None, _PROGRESS)
    stop_p = ctypes.byref(stop) if stop is not None else None
    if len(runners) == 1 and not force_multi:
        rc = _L.vgen_scan(runner._h, pattern.encode(), ctypes.byref(c), cb, None, stop_p, ctypes.byref(res))
    else:
        arr = (ctypes.c_void_p * len(runners))(*[r._h for r in runners])
        rc = _L.vgen_scan_multi(arr, len(runners), pattern.encode(), ctypes.byref(c), cb, None, stop_p, ctypes.byref(res))
    out = ScanResult(operations=res.operations, elapsed_secs=res.elapsed_secs,
                     resumed_operations=res.resumed_operations, complete=bool(res.complete), failed_shards=res.failed_shards)
    for i in range(res.n_matches):
        g = res.matches[i]
        out.matches.append(GeneratedAddress(g.address.decode(), g.wif.decode(), g.hex.decode(), AddressFormat(g.format)))
    _L.vgen_scan_result_free(ctypes.byref(res))
    if rc < 0:
        # a failing scan still hands over what its finished batches found (vgen_scan_result is filled, complete = 0)
        msg = _L.vgen_last_error(runner._h)
        err = VgenError(rc, msg.decode() if msg else "")
        err.partial = out
        raise err
    return out
