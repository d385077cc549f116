/*
 * vgen_hip.h — C ABI of libvgen_hip.so, the MI355X (gfx950) scan engine that stands in for the
 * wgpu/WGSL backend of oritwoen/vgen.
 *
 * The reference has no FFI layer (Cargo.toml:14 forbids unsafe code); its GPU path is entered
 * through the GpuRunner object and the scan_gpu_with_runner free function.  Each entry point below
 * names the reference interface it replaces (paths relative to the upstream repo root); the binding
 * a vgen fork would add is shown in INTEGRATION.md.
 *
 * Conventions: every call returns VGEN_OK (0) or a negative vgen_status; nothing throws across the
 * ABI; a vgen_ctx is bound to one HIP device and must be used from one host thread at a time
 * (different contexts are independent, as one GpuRunner per adapter would be); all multi-byte keys
 * are 32-byte BIG-endian scalars exactly as the reference passes them ([u8; 32], src/gpu.rs:535).
 * The library never falls back to a CPU implementation: without a usable HIP device vgen_create
 * fails with VGEN_E_NODEVICE.
 */
#ifndef VGEN_HIP_H
#define VGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGEN_ABI_VERSION 4   /* 2: frames = 0 selects 12; vgen_get_topology reports streams / priority levels; vgen_dispatch_random;
                                  vgen_scan_multi returns partial results beside an error
                                  3: vgen_get_resources; the fault-injection entry point left the library (test build only);
                                  random-key streams take a 24-byte seed (vgen_dispatch_random_seed / vgen_random_key_seed)
                                  4: the device-memory policy is the caller's: vgen_params.table_bits / device_mem_budget_bytes,
                                  vgen_scan_config.table_bits_max, vgen_get_memory; generator tables are replaced in the
                                  background, never by pausing a scan.  The ABI-3 layouts of vgen_params (28 bytes) and
                                  vgen_scan_config (136 bytes) are still accepted (struct_size tells): the new fields read as 0 */

typedef enum vgen_status {
    VGEN_OK = 0,
    VGEN_E_INVALID = -1,   /* bad argument */
    VGEN_E_NODEVICE = -2,  /* no usable HIP device / device index out of range */
    VGEN_E_HIP = -3,       /* a HIP runtime call failed (see vgen_last_error) */
    VGEN_E_NOMEM = -4,
    VGEN_E_STATE = -5,     /* e.g. vgen_wait on a frame with nothing in flight (gpu.rs:622-625) */
    VGEN_E_PATTERN = -6,   /* pattern empty / invalid / unsupported syntax (pattern.rs:21-33) */
    VGEN_E_RANGE = -7,     /* start key is 0 or >= n (SecretKey::from_slice failure, gpu.rs:903) */
    VGEN_E_UNSUPPORTED = -8
} vgen_status;

/* AddressFormat, src/address.rs:11-24 (same order). */
typedef enum vgen_format {
    VGEN_FMT_P2PKH = 0,
    VGEN_FMT_P2WPKH = 1,
    VGEN_FMT_P2SH_P2WPKH = 2,
    VGEN_FMT_P2TR = 3,
    VGEN_FMT_P2PKH_UNCOMPRESSED = 4,
    VGEN_FMT_ETHEREUM = 5
} vgen_format;

/* Parameters of vgen_create; replaces the arguments of GpuRunner::new(batch_size, backend)
 * (src/gpu.rs:138) plus the buffer sizing it derives from them (src/gpu.rs:391-500). */
/* vgen_params.flags */
#define VGEN_FLAG_ENDO 2u     /* vanity ("generate") searches only: every curve point of a dispatch is tested under its six
                                 endomorphism / negation images — keys k, lambda k, lambda^2 k and their negations, public
                                 keys (x, +-y), (beta x, +-y), (beta^2 x, +-y) — so a dispatch tests 6 x batch_size keys for
                                 one batch_size of point arithmetic (every format but P2TR, which runs as without the flag; with a
                                 prefilter pattern, an on-device DFA pattern or in dump mode).  The keys tested are NOT a contiguous range:
                                 vgen_scan refuses start / end / seed on such a context.  vgen_wait reports keys_tested
                                 = 6 x batch_size and match indices variant * batch_size + i (vgen_key_variant).  The
                                 arbitrary-scalar dispatches (vgen_dispatch_keys, vgen_dispatch_random) test the six images of
                                 every scalar's point in the same way (keys_tested = 6 x n, same index convention). */
#define VGEN_FLAG_TIMING 1u   /* record HIP events around every dispatch so that vgen_frame_kernel_ms /
                                 vgen_frame_dispatch_ms report durations (bench.py); without it a dispatch is
                                 three kernels and one copy, and the host loop is ~10 us per step cheaper */

#define VGEN_MAX_BATCH 16777216u   /* 2^24 keys per dispatch (the rate is flat from 2^20 up: profiles/r02_batch_sweep.txt) */

typedef struct vgen_params {
    uint32_t struct_size;  /* = sizeof(vgen_params) */
    int32_t device;        /* HIP device ordinal */
    uint32_t batch_size;   /* keys per dispatch; reference default 524288 (gpu.rs:83); must be a
                              multiple of 8192, at most VGEN_MAX_BATCH; 0 selects 1048576 */
    uint32_t format;       /* vgen_format */
    uint32_t frames;       /* dispatches that may be in flight; reference uses 2 (gpu.rs:399); 0 -> 12; max 20.  One dispatch is one wave per SIMD, so throughput grows with the frames
                              in flight: 9.7 / 14.7 / 15.4 / 16.5 Gkeys/s at 2 / 4 / 8 / 12 (P2PKH, 2^20 keys each; round 5).  The
                              first twelve frames get a hardware queue each (vgen_get_topology), whatever
                              GPU_MAX_HW_QUEUES is; multiples of 4 balance the queue pools */
    uint32_t match_cap;    /* match records kept per dispatch in filter mode; 0 -> 4096 */
    uint32_t flags;        /* VGEN_FLAG_* */
    /* ---- ABI 4: the device memory a context may take is the caller's decision, as in the reference, where a runner's device
     * memory is a pure function of batch_size (64 B x N table + 200 B x N per frame, limits requested up front: src/gpu.rs:216-231,391-402).
     * Here the frames are that function too (vgen_get_memory reports it); what is POLICY is the fixed-base generator table
     * of the paths that multiply a scalar per key (P2TR, vgen_dispatch_keys, vgen_dispatch_random): wider tables are faster
     * (24 bits: 11.8 GB, 10 additions per key; 27 signed: 21.5 GB, 9; 29 signed: 138 GB, 8: +12.5 %). */
    uint32_t table_bits;   /* generator-table width for this context: 0 = automatic (24 bits; vgen_scan moves to 27 / 29 bits
                              for scans long enough to pay for them, see vgen_scan_config.table_bits_max); else 8 | 16 | 20 |
                              22 | 24 | 26 (unsigned windows) or 25 | 27 | 29 (signed windows): exactly this width, stepping
                              down only when it cannot be allocated or passes the budget below */
    uint64_t device_mem_budget_bytes;   /* upper bound on the device memory this context holds (its frames, dump / key buffers and
                              the generator tables it references); 0 = automatic: the frames as sized by batch_size x frames, and
                              no generator table larger than HALF of the device memory free when the table is chosen
                              (hipMemGetInfo) — a device another tenant is using never loses 138 GB to a scan that "expects"
                              to be long.  With a budget, vgen_create fails with VGEN_E_NOMEM when the frames alone pass it,
                              later buffers (dump mode, arbitrary-scalar mode) fail the call that needs them, and the
                              generator table steps down to the widest width that fits (vgen_get_resources says so). */
} vgen_params;

/* One candidate reported by the device filter.  key = start_key + index.  payload is the 20-byte
 * hash160 / Ethereum address (memory order, as the reference's output buffer holds it,
 * src/gpu.rs:644-650) or the 32-byte x-only key for P2TR.  The device filter is a superset of the
 * pattern; the host confirms with vgen_filter_matches on the encoded address (the reference runs
 * pattern.matches on every key on the host, src/gpu.rs:1069). */
typedef struct vgen_match {
    uint32_t index;
    uint32_t reserved;
    uint8_t payload[32];
} vgen_match;

typedef struct vgen_ctx vgen_ctx;
typedef struct vgen_filter vgen_filter;

/* ---- library / device -------------------------------------------------------------------------- */

int vgen_abi_version(void);
/* Number of HIP devices; replaces the adapter enumeration of list_gpus / GpuRunner::new
 * (src/gpu.rs:149-207, src/lib.rs:979-1036). */
int vgen_device_count(int *n);
/* Device name into buf (NUL-terminated, truncated to cap). */
int vgen_device_name(int device, char *buf, size_t cap);

/* ---- context: GpuRunner (src/gpu.rs:116-131) ------------------------------------------------------ */

/* GpuRunner::new (src/gpu.rs:138-533): binds the device, allocates the frames and builds the shared
 * offset table (the reference's init_table dispatch, src/gpu.rs:502-517). */
int vgen_create(const vgen_params *p, vgen_ctx **out);
void vgen_destroy(vgen_ctx *ctx);
/* Message of the last failure on this context (or of the last failed vgen_create /
 * vgen_filter_compile on this thread when ctx is NULL).  Valid until the next call. */
const char *vgen_last_error(const vgen_ctx *ctx);
/* The batch size / frame count actually in use (after defaults). */
int vgen_get_info(const vgen_ctx *ctx, uint32_t *batch_size, uint32_t *frames, uint32_t *match_cap);
/* How the context reaches the device (the reference's wgpu queue, src/gpu.rs:116-131, has no counterpart to
 * tune): one stream per frame (*streams = frames).  The HIP runtime keeps one pool of GPU_MAX_HW_QUEUES (default 4,
 * reported in *hw_queues) hardware queues per stream priority level (*priority_levels: 3 on ROCm 7.2), and the
 * context spreads its streams over the levels: up to *priority_levels x *hw_queues streams own a queue each.
 * *oversubscribed == 1 reports that the context has more frames than that (frames > 12 by default): the surplus
 * streams share queues, which costs little (11.6 Gkeys/s on 4 queues against 12.1 on 12) but buys nothing.
 * Side effect to know about: frames 4-7 run on the runtime's HIGH priority level and 8-11 on its LOW one, relative to
 * the host application's own default-priority streams.  Any pointer may be NULL. */
int vgen_get_topology(const vgen_ctx *ctx, uint32_t *streams, uint32_t *hw_queues, uint32_t *priority_levels,
                      int32_t *oversubscribed);

/* What the context has — or will get — of two resources it sizes itself (the reference's GpuRunner allocates fixed
 * buffers for its two frames, src/gpu.rs:391-500, and has no tables).  Any pointer may be NULL.
 *   *dump_frames        frames that can be dispatched in DUMP mode (vgen_set_filter(NULL), or a filter of device kind 0):
 *                       all of them unless their pinned host mirrors would pass ~1 GiB (never fewer than 2) — e.g. 8 of 12
 *                       on a VGEN_FLAG_ENDO context at 2^20 keys per dispatch (6 x 20 MB per frame).  vgen_dispatch* on a
 *                       frame >= *dump_frames in dump mode is VGEN_E_STATE; vgen_scan drives only the frames that have a buffer.
 *   *table_bits         window width of the fixed-base generator table the scalar-multiplication paths (P2TR, vgen_dispatch_keys,
 *                       vgen_dispatch_random) use: 0 = none built yet (first use builds it), 8 = the 8-bit table only,
 *                       16 / 20 / 22 / 24 / 26 = the wide table in use (shared by all contexts of the process on this device);
 *                       odd values are SIGNED windows: 25 / 27 / 29 bits = 10 / 9 / 8 additions in 5.9 / 21.5 / 138 GB.
 *   *table_bits_wanted  the width asked for: vgen_params.table_bits when set; else what vgen_scan chose for the scan's expected length
 *                       (>= 3 s of keys: 27, >= 30 s: 29, capped by vgen_scan_config.table_bits_max; a context never steps back
 *                       down by itself); else the default, 24.  (The VGEN_GTAB_BITS environment variable overrides all three: a
 *                       test switch, read once at vgen_create.)  A wider table than the one in use is built IN THE BACKGROUND — on a
 *                       low-priority stream, by a thread of its own, while dispatches go on with the old table — and taken into
 *                       use by the first dispatch after it is ready: *table_bits < *table_bits_wanted with an empty note means
 *                       "on its way".  With a note it means the wider tables could not be had — allocation or build failed, the
 *                       budget or the half-of-free-memory rule forbids them — and the context stepped down in order of
 *                       (additions per multiplication, bytes): 29s -> 27s -> 26 -> 25s -> 24 -> 22 -> 20 -> 16 -> 8 (8 / 9 / 9 / 10 /
 *                       10 / 11 / 12 / 15 / 31 additions): same keys, fewer per second.
 *   note                why it stepped down ("" when it did not), NUL-terminated, truncated to note_cap.
 * A step-down is not an error: the dispatch that caused it returned VGEN_OK and vgen_last_error is untouched. */
int vgen_get_resources(const vgen_ctx *ctx, uint32_t *dump_frames, uint32_t *table_bits, uint32_t *table_bits_wanted,
                       char *note, size_t note_cap);

/* Device and pinned host memory of a context (the reference requests its buffer limits up front, src/gpu.rs:216-231, and sizes every
 * buffer from batch_size, src/gpu.rs:391-402).  Fill struct_size before the call. */
typedef struct vgen_memory_info {
    uint32_t struct_size;        /* = sizeof(vgen_memory_info) */
    uint32_t table_bits;         /* generator table in use (as vgen_get_resources) */
    uint64_t frames_bytes;       /* device memory of the frames and per-context tables: scratch, match rings, offset table, filter (fixed at vgen_create) */
    uint64_t mode_bytes;         /* device memory of buffers made at first use: dump mode's payload buffers, the arbitrary-scalar path's keys + scratch */
    uint64_t table_bytes;        /* generator tables this context references (shared by the process's contexts on the device: counted in full
                                    by each), including a retired one a dispatch in flight still reads and one being built */
    uint64_t pinned_host_bytes;  /* page-locked host memory: match-ring mirrors, dump mirrors */
    uint64_t budget_bytes;       /* vgen_params.device_mem_budget_bytes (0 = automatic) */
    uint64_t device_free_bytes;  /* hipMemGetInfo now */
    uint64_t device_total_bytes;
} vgen_memory_info;
int vgen_get_memory(const vgen_ctx *ctx, vgen_memory_info *out);

/* ---- pattern: Pattern::new / Pattern::matches (src/pattern.rs:21-45) -------------------------------- */

/* Compiles `pattern` (prefixed with "(?i)" when case_insensitive, pattern.rs:26-30) into (a) a DFA
 * that decides Pattern::matches exactly for ASCII address strings and (b) a device prefilter for
 * `format`.  Syntax: the regex crate's, as far as it can matter on ASCII haystacks (flags i m s x U u, word
 * boundaries, Unicode / POSIX classes by their ASCII members, nested classes with && -- ~~); the CRLF flag R
 * and Unicode class names outside the built-in table are VGEN_E_PATTERN, as are empty or invalid patterns
 * (pattern.rs:22-24,32-33). */
int vgen_filter_compile(const char *pattern, int case_insensitive, uint32_t format, vgen_filter **out);
void vgen_filter_free(vgen_filter *f);
/* Pattern::matches (pattern.rs:43-45): unanchored regex search over the address string. 1 / 0. */
int vgen_filter_matches(const vgen_filter *f, const char *address);
/* How the device evaluates this filter: 0 = every key is reported to the host (no usable
 * prefilter; reference-equivalent host filtering), 1 = hash160 range test (Base58 prefixes),
 * 2 = masked-bits test (Bech32 / hex prefixes and suffixes), 3 = match-all, 4 = the pattern's whole DFA runs on the
 * device over the encoded address (unanchored patterns, Base58 suffixes). */
int vgen_filter_device_kind(const vgen_filter *f);
/* Size in bytes of the automaton a kind-4 filter stages into LDS (at most 48 KiB; 0 for the other kinds).  With the
 * product tree (and, on a VGEN_FLAG_ENDO context of the uncompressed / Ethereum formats, the parked y coordinate)
 * beside it a workgroup must stay within 64 KiB of LDS: an ENDO dispatch whose automaton does not leave room for the
 * parked coordinate tests the plain keys only (vgen_wait then reports keys_tested = batch_size for it). */
int vgen_filter_dfa_bytes(const vgen_filter *f);
/* Pattern::validate_charset(format) (src/pattern.rs:49-177): the characters of `pattern` that can never
 * occur in an address of `format` (literals outside classes; members of non-negated classes with no
 * valid member), in order of first appearance.  Writes up to cap-1 characters + NUL into `out` and the
 * full count into *n.  The caller warns "pattern will NEVER match" when *n > 0 (src/lib.rs:684-706). */
int vgen_pattern_invalid_chars(const char *pattern, int case_insensitive, uint32_t format, char *out, size_t cap,
                               size_t *n);
/* Pattern::estimate_difficulty(format) (src/pattern.rs:183-253): "1 in N" heuristic = alphabet size
 * (58 | 34 case-insensitive | 32 | 16) to the number of fixed alphanumeric pattern characters, less the
 * characters of the format's constant prefix when the pattern is anchored on it; saturates at 2^64-1. */
int vgen_pattern_difficulty(const char *pattern, int case_insensitive, uint32_t format, uint64_t *out);
/* AddressFormat::charset_name (src/address.rs:39-45): "Base58" | "Bech32" | "Hex"; NULL for an unknown format. */
const char *vgen_format_charset_name(uint32_t format);
/* Selects the filter for subsequent dispatches; NULL = dump mode (every payload is written,
 * index order — the reference kernel's behaviour, src/shaders/search.wgsl:2-31).  Dump mode serves the first
 * *dump_frames frames of the context (vgen_get_resources): its pinned host mirrors are bounded to ~1 GiB. */
int vgen_set_filter(vgen_ctx *ctx, const vgen_filter *f);

/* Resizes the match rings of all frames to match_cap records per dispatch (clamped to [256, batch_size], 6 x batch_size on a
 * VGEN_FLAG_ENDO context); only while
 * no dispatch is in flight.  vgen_scan uses it to keep permissive patterns on the device filter: the reference hands
 * EVERY hash to the host (src/gpu.rs:602-658); here the ring grows to hold the expected candidates instead. */
int vgen_set_match_cap(vgen_ctx *ctx, uint32_t match_cap);

/* ---- measurement aid -------------------------------------------------------------------------------------- */

/* Starts a one-wave probe on its own stream that, for duration_ms, compares the shader-clock counter with
 * the constant 100 MHz counter; vgen_clock_probe_read waits for it and returns the clock (MHz) the CUs ran at
 * meanwhile — i.e. under whatever the frames were executing.  The probe's stream takes a hardware queue from the
 * normal-priority pool, which it may share with a frame and then holds up that frame's kernels: for measurements
 * beside a saturated scan prefer vgen_frame_clock, which costs no queue. */
int vgen_clock_probe_start(vgen_ctx *ctx, uint32_t duration_ms);
int vgen_clock_probe_read(vgen_ctx *ctx, double *mhz);

/* ---- dispatch / readback ----------------------------------------------------------------------------- */

/* GpuRunner::dispatch(start_key, frame) (src/gpu.rs:535-600): asynchronously tests the keys
 * start_key + i, i in [0, batch_size).  Keys >= n yield no result (increment_key -> None,
 * src/gpu.rs:963).  start_key must be a valid scalar. */
int vgen_dispatch(vgen_ctx *ctx, uint32_t frame, const uint8_t start_key_be[32]);
/* Arbitrary-scalar mode (the CPU path's "independent random key" shape, src/scanner.rs:151-155):
 * tests keys_be[32*i], i < n <= batch_size, with a full fixed-base multiplication per key.
 * Invalid scalars (0 or >= n) yield no result (address.rs:93).  The upload rides the frame's stream: a pageable
 * keys_be has been consumed when the call returns, a pinned (hipHostMalloc / hipHostRegister) one must stay
 * unchanged until vgen_wait(frame) returns. */
int vgen_dispatch_keys(vgen_ctx *ctx, uint32_t frame, const uint8_t *keys_be, uint32_t n);
/* Independent random keys — the shape of the reference's CPU hot loop, which draws 32 fresh bytes per candidate
 * (rng.fill, src/scanner.rs:144-152) — without any upload: lane i of the dispatch tests candidate first_index + i of
 * the counter-based scalar stream
 *     key(seed, stream, index) = SHA-256("vgen-mi355x-rand" || seed[24] || u32le(stream) || u64le(index)),
 * computed on the device (one SHA-256 compression per lane); draws that are 0 or >= n yield no result, as the
 * reference skips them (src/address.rs:93).  A match reports index i: vgen_random_key_seed re-derives its key on the host.
 * `stream` separates scanners that share a seed (one per GPU / shard).  Always batch_size candidates.
 * THE SEED IS ALL THE SECRET THERE IS: stream and index are small, so a key found by this mode is exactly as hard to
 * recover from its address as the seed is to guess.  Real searches use the 24-byte form with 24 bytes of OS entropy
 * (vgen_scan does when vgen_scan_config.seed == 0; the reference seeds a 256-bit StdRng from the OS, src/scanner.rs:144);
 * 24 rather than 32 bytes keep the message in one SHA-256 block.  The uint64_t forms stand for seed = u64le(seed) ||
 * sixteen zero bytes and exist for reproducible runs and tests — never for keys that will hold value. */
int vgen_dispatch_random_seed(vgen_ctx *ctx, uint32_t frame, const uint8_t seed[24], uint32_t stream, uint64_t first_index);
int vgen_dispatch_random(vgen_ctx *ctx, uint32_t frame, uint64_t seed, uint32_t stream, uint64_t first_index);
/* The key vgen_dispatch_random*'s lane (index - first_index) tests; VGEN_E_RANGE when that draw is not a valid scalar. */
int vgen_random_key_seed(const uint8_t seed[24], uint32_t stream, uint64_t index, uint8_t key_be[32]);
int vgen_random_key(uint64_t seed, uint32_t stream, uint64_t index, uint8_t key_be[32]);
/* GpuRunner::await_result(frame) (src/gpu.rs:602-658): blocks until the frame's dispatch is done.
 * Filter mode: copies up to cap records to out (ascending index), stores the number found in
 * *n_matches (may exceed cap: the surplus was dropped) and the number of keys tested in *keys_tested.
 * Dump mode: n_matches = 0; fetch the payloads with vgen_read_dump.  Any pointer may be NULL. */
int vgen_wait(vgen_ctx *ctx, uint32_t frame, vgen_match *out, uint32_t cap, uint32_t *n_matches,
              uint64_t *keys_tested);
/* Dump mode only, after vgen_wait: copies the frame's payloads (batch_size * 20 bytes, or * 32 for
 * P2TR; zeroed for invalid keys) — the Vec<[u8;20]> await_result returns (src/gpu.rs:644-650). */
int vgen_read_dump(vgen_ctx *ctx, uint32_t frame, uint8_t *out, size_t out_len);
/* (Both dump readers: frames 0 .. dump_frames-1 only, see vgen_get_resources.)
 * The same payloads without the copy: a dump-mode dispatch ends with its own asynchronous transfer into pinned
 * host memory, and this returns that buffer (valid until the frame is dispatched again) — what the host-side
 * filter loop of scan_gpu_with_runner reads (src/gpu.rs:1030-1093). */
int vgen_dump_view(vgen_ctx *ctx, uint32_t frame, const uint8_t **ptr, size_t *len);
/* Device time of the dominant kernel (seq_bwd_kernel: tree walk-down, point additions, hashes,
 * filter) of the frame's last completed dispatch, from HIP events on the frame's own stream. */
int vgen_frame_kernel_ms(vgen_ctx *ctx, uint32_t frame, float *ms);
/* Device time of the whole dispatch (seq_fwd incl. the root inversions + seq_bwd, incl. the gap between them). */
int vgen_frame_dispatch_ms(vgen_ctx *ctx, uint32_t frame, float *ms);

/* Shader-clock sample of the frame's last completed filter-mode dispatch: the first wave of its seq_bwd_kernel
 * launch read the shader-clock counter and the constant 100 MHz counter when it started and when it ended;
 * *cycles / *ticks_100mhz x 100 = the MHz the CUs ran at while that kernel executed (0 / 0 after a dump-mode or
 * P2TR dispatch).  No extra launch, no extra queue: bench.py sums the samples of its sustained leg. */
int vgen_frame_clock(vgen_ctx *ctx, uint32_t frame, uint32_t *cycles, uint32_t *ticks_100mhz);

/* ---- host-side derivation (what the Rust host obtains from rust-bitcoin) ----------------------------- */

/* Address string from a device payload (src/gpu.rs:1034-1060).  Returns length or negative status. */
int vgen_address_from_payload(uint32_t format, const uint8_t *payload, char *out, size_t cap);
/* AddressGenerator::bytes_to_wif (src/address.rs:168-172); uncompressed form for
 * VGEN_FMT_P2PKH_UNCOMPRESSED, hex for Ethereum (address.rs:110).  Returns length or negative status. */
int vgen_key_to_wif(uint32_t format, const uint8_t key_be[32], char *out, size_t cap);
/* increment_key (src/gpu.rs:951-968): out = key + amount; VGEN_E_RANGE on overflow or invalid scalar. */
int vgen_key_add(const uint8_t key_be[32], uint64_t amount, uint8_t out_be[32]);
/* Endomorphism contexts (VGEN_FLAG_ENDO) test six keys per curve point; a match's index is variant * batch_size + i.
 * This maps the base key start_key + i to the key of `variant`: 0..2 = lambda^variant * k, 3..5 = the negations
 * (mod n; lambda * (x, y) = (beta * x, y) on secp256k1).  VGEN_E_RANGE for an invalid key. */
int vgen_key_variant(const uint8_t key_be[32], uint32_t variant, uint8_t out_be[32]);
/* AddressGenerator::generate (src/address.rs:92-151) on the host, for single keys (verify-style
 * use and tests). address cap >= 96, wif cap >= 72. VGEN_E_RANGE for an invalid key. */
int vgen_derive(uint32_t format, const uint8_t key_be[32], char *address, size_t acap, char *wif, size_t wcap);

/* ---- provider patterns (src/provider.rs) -------------------------------------------------------------- */

/* provider::resolve (src/provider.rs:12-52): "boha:b1000:66" / "boha:b1000/66" -> the puzzle's target
 * address, its address format and (when known) its key range.  The reference reads puzzles from the
 * un-vendored `boha` crate; this build has a static table — every b1000 puzzle's range (2^(N-1)..2^N-1)
 * and the addresses the reference itself pins (puzzles 1 and 66) — extended by the optional CSV file
 * `table_path` (rows "collection/id,address,kind[,start_hex,end_hex]", kind p2pkh|p2wpkh|p2tr|p2sh).
 * Returns 1 = resolved, 0 = not a provider pattern (treat it as a regex), VGEN_E_INVALID = unknown
 * puzzle / bad table (message in vgen_last_error(NULL)). */
int vgen_provider_resolve(const char *pattern, const char *table_path, char *address, size_t acap, uint32_t *format,
                          int32_t *has_range, uint8_t start_be[32], uint8_t end_be[32]);
/* provider::build_pattern / build_exact_pattern (src/provider.rs:54-62): prefix_length 0 = the exact
 * pattern "^address$", otherwise "^" + the first prefix_length characters (clamped); metacharacters escaped. */
int vgen_provider_build_pattern(const char *address, uint32_t prefix_length, char *out, size_t cap);

/* ---- scanner: scan_gpu_with_runner (src/gpu.rs:920-1125) ----------------------------------------------- */

/* ScanConfig (src/scanner.rs:17-46) restricted to the fields the GPU path reads. */
typedef struct vgen_scan_config {
    uint32_t struct_size;
    uint32_t format;
    uint64_t count;          /* matches to find; UINT64_MAX = unbounded (lib.rs:524) */
    int32_t case_insensitive;
    int32_t has_start;       /* 0: random base key (gpu.rs:936-945) drawn from `seed` */
    uint8_t start[32];
    int32_t has_end;
    uint8_t end[32];
    uint64_t seed;           /* build-side addition (the reference seeds from OS entropy): base key
                                k0(seed, shard) of BASELINE.md §4; seed 0 = OS entropy */
    uint32_t shard;          /* this scanner's index among n_shards batch-striped scanners */
    uint32_t n_shards;       /* 0/1 = single device; >1: this context takes global batches b with
                                b % n_shards == shard */
    uint64_t max_batches;    /* stop after this many dispatches per shard IN THIS CALL (0 = no limit): a resumed scan
                                counts from where the checkpoint left off */
    /* Build-side addition (the reference cannot resume): when non-NULL, the scan records the finished
     * batches of every shard and the matches found in them in this file (rewritten atomically, at most
     * every checkpoint_interval_ms and once when the scan returns) and, when the file already exists and
     * describes the same scan (pattern, format, batch size, sharding, end, and the base key if start or
     * seed pin it), resumes after the recorded batches, with the recorded matches counting towards
     * `count`.  An unseeded random scan adopts the file's base key.  A file of a different scan is
     * VGEN_E_INVALID. */
    const char *checkpoint_path;
    uint32_t checkpoint_interval_ms;   /* 0 = 10 s */
    uint32_t flags;                    /* VGEN_SCAN_* */
    /* ---- ABI 4 ---- */
    uint32_t table_bits_max;           /* scans that multiply a scalar per key (P2TR, VGEN_SCAN_RANDOM_KEYS): the widest generator table this
                                          scan may move the context to — compared by additions per multiplication, so 24 also forbids 25 / 27 / 29 —;
                                          0 = no cap beyond the context's memory policy (vgen_params.device_mem_budget_bytes).  The scan asks
                                          for 27 bits from ~3 s and 29 bits from ~30 s of expected (or, after 5 s, observed) scanning; the table
                                          is built in the background and the scan never waits for it. */
    uint32_t reserved;                 /* 0 */
} vgen_scan_config;

/* vgen_scan_config.flags */
#define VGEN_SCAN_RANDOM_KEYS 1u   /* scan_with_progress's shape (src/scanner.rs:118-169): every candidate an independent random key
                                      (vgen_dispatch_random: batch b tests candidates b * batch_size .. of stream `shard` under
                                      `seed`; seed 0 = 24 bytes of OS entropy, the only setting for keys that will hold value: a
                                      nonzero 64-bit seed makes the run reproducible and its keys guessable, see
                                      vgen_dispatch_random_seed) instead of the reference GPU path's walk from one base key.
                                      A full scalar multiplication per key: ~10x slower than the walk.  No start / end; a
                                      checkpoint records the seed and the batches done per stream.  With a fixed seed (and without VGEN_FLAG_ENDO) the matches are those of
                                      the oracle's scan_random walk of the same stream, in the same order.  On a VGEN_FLAG_ENDO
                                      context every draw is tested as six keys (the candidate and its lambda / negation images,
                                      one multiplication for the six: 5.5 instead of 1.35 Gkeys/s); seeds and shards keep their
                                      meaning there (they name streams of candidates, not ranges). */

/* GeneratedAddress (src/address.rs:63-72). */
typedef struct vgen_generated {
    char address[96];
    char wif[72];
    char hex[72];
    uint32_t format;
    uint8_t key[32];
} vgen_generated;

/* ScanResult (src/scanner.rs:50-68). */
typedef struct vgen_scan_result {
    vgen_generated *matches;  /* vgen_scan_result_free */
    uint64_t n_matches;
    uint64_t operations;      /* of this call */
    double elapsed_secs;
    uint64_t resumed_operations;  /* operations recorded in the checkpoint this call resumed from (else 0) */
    int32_t complete;         /* 1: the key range ran out (every shard reached `end` / the end of the key space) */
    int32_t failed_shards;    /* contexts that failed during the scan (0 normally).  vgen_scan_multi with survivors: their
                                 stripes were taken over and the call still returned VGEN_OK; a failing call (negative status)
                                 fills this structure all the same, with the matches of every batch finished before the
                                 failure and complete = 0 */
} vgen_scan_result;

typedef void (*vgen_progress_cb)(uint64_t operations, void *user); /* ProgressCallback, scanner.rs:71 */

/* scan_gpu_with_runner(pattern, config, progress_cb, stop, runner) (src/gpu.rs:920-926): drives the
 * frames of `ctx` round-robin, confirms device candidates on the host, collects up to count matches
 * in ascending key order within a batch (gpu.rs:1095-1104), counts batch_size operations per
 * completed batch (gpu.rs:1106) and honours *stop between batches (gpu.rs:980-984,1007-1011). */
int vgen_scan(vgen_ctx *ctx, const char *pattern, const vgen_scan_config *cfg, vgen_progress_cb cb,
              void *user, const volatile int32_t *stop, vgen_scan_result *out);
/* The same scan over several contexts (one per GPU of the node; the reference is single-adapter,
 * src/gpu.rs:161-165): one host thread per context, global batch b goes to context b mod n_ctx, a
 * shared match counter / stop flag, matches merged in ascending key order and truncated to count,
 * operations summed.  No device-to-device traffic.  All contexts must share batch_size
 * and format; cfg->shard / n_shards are ignored.
 * A context whose device fails mid-scan retires; the matches of its finished batches are kept, and a context that has
 * finished its own stripe takes the failed one over from its last finished batch (the reference's answer to a failing GPU
 * is its CPU fallback, src/lib.rs:727-746,1185-1198; there is no CPU path here, the other GPUs are).  The call returns
 * VGEN_OK with out->failed_shards > 0 when the survivors covered everything (or `count` / the stop flag ended the scan),
 * and the first failure's status — with out still filled, complete = 0 — when no context was left to do so. */
int vgen_scan_multi(vgen_ctx **ctxs, uint32_t n_ctx, const char *pattern, const vgen_scan_config *cfg,
                    vgen_progress_cb cb, void *user, const volatile int32_t *stop, vgen_scan_result *out);
void vgen_scan_result_free(vgen_scan_result *r);

#ifdef __cplusplus
}
#endif
#endif
