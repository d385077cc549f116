// device_types.h — plain-data structures shared between the HIP kernels and the host runtime.
#pragma once
#include <stdint.h>

namespace vg {

// vgen_format values (include/vgen_hip.h), usable as template arguments in device code
enum : int {
    VGF_P2PKH = 0,
    VGF_P2WPKH = 1,
    VGF_P2SH_P2WPKH = 2,
    VGF_P2TR = 3,
    VGF_P2PKH_UNCOMPRESSED = 4,
    VGF_ETHEREUM = 5
};


// Device-side prefilter program (built by host/filter.cpp from the pattern's DFA).
// The test runs on the 160-bit payload seen as five BIG-endian words H[0..4] (H[0] = bytes 0..3).
enum : uint32_t {
    DEVF_HOST_ALL = 0,   // no usable prefilter: every key is a candidate (dump + host filter)
    DEVF_RANGES = 1,     // any r:  lo_r <= H <= hi_r           (Base58 prefixes)
    DEVF_MASKED = 2,     // any t:  (H & mask_t) == value_t     (Bech32 / hex prefixes & suffixes)
    DEVF_ALL = 3,        // pattern accepts every address
    DEVF_DFA = 4         // full match on the device: encode the address, walk the DFA (core/dfa_eval.h)
};

constexpr uint32_t DEVF_MAX_TESTS = 64;
constexpr uint32_t DEVF_FLAG_BECH32_CHK = 1u;   // masked tests also constrain the bech32 checksum

constexpr int DEVF_WORDS = 8;   // payload words a test covers: 5 used for 20-byte payloads, 8 for P2TR

struct DevFilterTest {
    uint32_t a[DEVF_WORDS];   // ranges: lo      masked: mask
    uint32_t b[DEVF_WORDS];   // ranges: hi      masked: value
    uint32_t chk_mask;   // bech32 checksum (30 bits, first symbol in bits 29..25)
    uint32_t chk_value;
};

struct DevFilter {
    uint32_t kind;
    uint32_t count;
    uint32_t flags;
    uint32_t witver;     // bech32: witness version symbol (0 for P2WPKH)
    // Bech32 checksum as an affine map of the payload bytes: chk = chk_base ^ XOR_i chk_lut[i*256 + byte_i]
    // (20 x 256 words; 32 x 256 for the 32-byte P2TR payload).  Device pointer in the device copy, host pointer in the host copy; nullptr = use
    // the step-by-step polymod.
    const uint32_t *chk_lut;
    uint32_t chk_base;
    uint32_t dfa_bytes;  // DEVF_DFA: size of the DFA blob (<= 48 KiB, staged in LDS by the kernel)
    // DEVF_DFA: the blob (layout in core/dfa_eval.h); device pointer in the device copy
    const uint32_t *dfa_blob;
    DevFilterTest tests[DEVF_MAX_TESTS];
};

// Match ring of one frame (device memory).  count may run past cap; records beyond cap are dropped.
struct DevMatch {
    uint32_t index;
    uint32_t reserved;
    uint32_t payload[8];
};

struct DevMatchHeader {
    uint32_t count;
    uint32_t cap;
    uint32_t clk_cycles;   // running sums (mod 2^32) over the frame's seq_bwd launches, first wave of each launch:
    uint32_t clk_ticks;    // shader-clock cycles and 100 MHz ticks it ran for (kernels.hip: "shader-clock sample")
};

// Per-dispatch uniform points of the sequential kernel: Q_j = (k0 + N/2 - S/2 + j)*G and the
// canonical negations of their coordinates, nine 29-bit limbs each.
struct DevSeqQ {
    uint32_t qx[9], qy[9], nqx[9], nqy[9];
};

// Building the offset table on the device (rtab_build_kernel): lane u computes R_u = base + u * step from the
// doublings pw[b] = 2^b * step (affine, canonical limbs, passed by value), bits of u selecting the summands.
struct DevAffine {
    uint32_t x[9], y[9];
};
struct RtabArgs {
    uint32_t *rtab;            // [18][lanes], limb-major
    uint32_t lanes;
    uint32_t nbits;            // ceil(log2(lanes)) <= 24
    DevAffine base;
    DevAffine pw[24];
};

constexpr uint32_t SEQ_MAX_S = 16;   // keeps SeqArgs (passed by value as kernel arguments) under 4 KB
constexpr int SEQ_WG = 256;          // lanes per workgroup of the seq_* kernels

struct SeqArgs {
    const uint32_t *rtab;      // offset table, limb-major: [18][lanes]  (x limbs 0..8, y limbs 0..8)
    const DevFilter *filter;   // used when dump == nullptr
    uint32_t *dump;            // dump mode: N * 5 words
    DevMatchHeader *mhdr;      // filter mode: monotonic candidate counter ...
    DevMatch *mrec;            // ... and this frame's record ring
    uint32_t *pre;             // scratch: prefix products [S][9][lanes]
    uint32_t *tree;            // scratch: product-tree nodes [groups][9][SEQ_WG]
    uint32_t *root;            // scratch: tree roots / their inverses [9][groups]
    uint32_t *xs;              // scratch of the split form (compressed-key formats): [9][N] — eight words of x and the prefix byte 0x02 | parity(y) of
                               // every key, slot (2j + sgn) * lanes + u; nullptr = the fused seq_bwd_kernel does everything
    uint32_t lanes;            // N / (2*S)
    uint32_t groups;           // lanes / SEQ_WG
    uint32_t n;                // N
    uint32_t s;                // S
    uint32_t match_base;       // value of mhdr->count when this dispatch was enqueued
    uint32_t match_cap;
    const uint32_t *dfa_blob;  // DEVF_DFA: the automaton (device memory), staged into LDS by the kernel
    const uint32_t *gtab;      // P2TR: 8-bit fixed-window generator table (global memory) for the tweak multiplication
    const uint32_t *gtab16;    // ... and the wide-window one (built on the device at first use); nullptr = use gtab
    uint32_t gtab_bits;        // window width of gtab16 (16 | 20 | 22 | 24)
    uint32_t dfa_bytes;        // 0 = prefilter mode
    uint32_t fmt;              // VGF_* of the context (the DFA path needs the exact address format)
    uint32_t hash_kpl;         // split form: keys per lane of seq_hash_kernel (a divisor of 2S)
    uint32_t endo;             // seq_bwd: test the six endomorphism / negation images of every point (kernels.hip: ENDO)
    uint32_t lone;             // seq_bwd: at most one other frame of the context was in flight when this dispatch was issued: launch the variant without issue-slot yields
    // P2TR only: the tweaked points Q = P + t*G of a dispatch wait for a second shared inversion.
    uint32_t *tq;              // [2S key steps][27][lanes]: X(Q), Z(Q), running product of the lane's Z's
    uint32_t *tq_flag;         // [2S][lanes]: 1 = the key has an address (valid tweak, Q finite)
    uint32_t *tree2;           // product tree / roots of the lanes' final products, laid out like tree / root
    uint32_t *root2;
    DevSeqQ q[SEQ_MAX_S];      // per-dispatch uniform points, by value (scalar loads from the kernarg segment)
};

// Arbitrary-scalar path (keys_fwd_kernel -> seq_inv_kernel -> keys_bwd_kernel): one key per lane, full
// fixed-base multiplication over the 8-bit window table (32 windows x 255 points x 20 words = 652 800 B in
// global memory, core/ec.h); the Jacobian results go through global scratch and share their inversions exactly as
// the sequential path does (workgroup product tree, one root per lane in seq_inv_kernel).
constexpr int KEYS_WG = 256;

struct KeysArgs {
    const uint32_t *gtab;      // [32][255][20]: x limbs 0..8, y limbs 9..17 of d * 256^w * G (d = 1..255)
    const uint32_t *gtab16;    // wide windows: [windows][2^bits - 1][16 words] (core/ec.h: ec_mul_gen_wide); nullptr = use gtab
    uint32_t gtab_bits;        // window width of gtab16 (16 | 20 | 22)
    const uint8_t *keys_be;    // n * 32 bytes big-endian (uploaded, or drawn on the device by rnd_fill_kernel), or nullptr: key i = base + i
    const DevFilter *filter;
    uint32_t *dump;            // dump mode: n * 5 words (zeroed for invalid keys)
    DevMatchHeader *mhdr;
    DevMatch *mrec;
    uint32_t base[8];          // sequential variant: little-endian words of the first key
    uint32_t n;
    uint32_t match_base;
    uint32_t match_cap;
    uint32_t fmt;              // VGF_* of the context
    const uint32_t *dfa_blob;  // DEVF_DFA (see SeqArgs)
    uint32_t dfa_bytes;
    uint32_t groups;           // workgroups = ceil(n / KEYS_WG)
    uint32_t endo;             // keys_bwd: test the six endomorphism / negation images of every point (kernels.hip: ENDO) ...
    uint32_t vstride;          // ... image `variant` of key i is reported / dumped at variant * vstride + i (= the context's batch size)
    uint32_t *pts;             // P2TR: affine internal keys in key order, [n][16] words (written by keys_bwd_kernel<P2TR>,
                               // read by p2tr_tweak_kernel); y = 0 marks "no key"
    uint32_t *xyz;             // scratch: Jacobian results, limb-major [27][groups * KEYS_WG] (X, Y, Z limbs; the taproot stage: X, validity word, Z)
    uint32_t *tree;            // scratch: product-tree nodes [groups][9][KEYS_WG]
    uint32_t *root;            // scratch: tree roots / their inverses [9][groups]
};

}  // namespace vg
