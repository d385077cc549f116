"""Generator of device/hash_blocks.inc: the address hashes of the scan kernels as scheduled gfx950 instruction blocks.

Each hash160 of the path (SHA-256 of the serialized key followed by RIPEMD-160) is executed SYMBOLICALLY here: a value is a
Python int when the padded message, the IVs or the round constants determine it, the name of a virtual register otherwise.
Known values fold (the constant tail of the message schedule, the first rounds on the IV, K[r] + W[r] for constant W), so the
emitted list is the floor of the computation: 6 rotates + 4 three-input booleans + 5 additions per SHA-256 round, 1 boolean
+ 2 rotates + 2 additions per RIPEMD-160 step.  The list is register-allocated by a linear scan (a pair lives in 28 VGPRs
and one SGPR for literals) and written out as ONE `asm` statement per hash, so the order below IS the order the SIMD sees.

Two things are decided here rather than by hipcc, both measured in the real kernels on the MI355X (EXPERIMENTS.md):
  * the ORDER: alternating RUNS of half-rate (v_alignbit, v_add3, v_perm) and of full-rate instructions (VOP2 add / shift, v_bitop3), made by
    a list scheduler over a window of 4 places of the dependency order (`by_class`; 29 registers instead of 28).
  * the PRIORITY: an `s_setprio` at every class boundary — half-rate runs at wave priority 1 (the level the kernels' point arithmetic runs
    at), full-rate runs at 0.  A gfx950 SIMD fills a 4-cycle issue slot with the next instruction of its highest-priority ready wave and, behind
    it, ONE full-rate instruction of another wave; half-rate instructions only take the first place.  Under age-only arbitration the oldest
    wave owns that place and a mixed stream costs 4 cycles per instruction, the full-rate ones included (tools/ubench_phase*.hip); with the
    priorities the full-rate runs of three waves ride behind the fourth's half-rate run: the hash pair alone 18.6 -> 23.4 Gpairs/s at four
    waves per SIMD, the scan 13.2 -> 15.5 Gkeys/s (profiles/r05_prio_ab.txt).  Round 4's form — dependency order with an `s_nop 0` after every
    third instruction, which makes the oldest wave skip a slot now and then (+5.7 %) — is `--class-window 0 --prio none --yield every:3`.
    A launch that has the chip to itself at ONE wave per SIMD pays 4 cycles per priority change: such dispatches launch a twin of the kernel
    that keeps hipcc's schedule of core/hash.h (kernels.hip: LONE).

The same instruction lists run in Python (`evaluate`, `evaluate_allocated`) for the CPU test-suite: tests/test_hashgen.py
checks them, before and after register allocation, against hashlib and the oracle.

usage: python3 hashgen.py [--order natural|grouped] [--class-window W] [--prio HALF:FULL[:EXIT]|none] [--class-max-run N] [--class-distance D]
       [--split-kadd m:n] [--split-add3 m:n] [--yield every:N|dep|none] [--yield-for FUNCTION=MODE] [--with FUNCTION] [--window W --distance D] > hash_blocks.inc
Algorithms restated from core/hash.h (reference: src/shaders/sha256.wgsl:43-170, src/shaders/ripemd160.wgsl:10-100)."""
import sys

M = 0xFFFFFFFF
K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
     0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
     0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
     0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
     0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
     0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
SHA_IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
RMD_IV = [0x67452301, 0xEFCDAB89, 0x98BADCFE, 0x10325476, 0xC3D2E1F0]
RL = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 7, 4, 13, 1, 10, 6, 15, 3, 12, 0, 9, 5, 2, 14, 11, 8, 3, 10, 14, 4, 9, 15, 8, 1, 2, 7,
      0, 6, 13, 11, 5, 12, 1, 9, 11, 10, 0, 8, 12, 4, 13, 3, 7, 15, 14, 5, 6, 2, 4, 0, 5, 9, 7, 12, 2, 10, 14, 1, 3, 8, 11, 6, 15, 13]
RR = [5, 14, 7, 0, 9, 2, 11, 4, 13, 6, 15, 8, 1, 10, 3, 12, 6, 11, 3, 7, 0, 13, 5, 10, 14, 15, 8, 12, 4, 9, 1, 2, 15, 5, 1, 3, 7, 14, 6, 9, 11,
      8, 12, 2, 10, 0, 4, 13, 8, 6, 4, 1, 3, 11, 15, 0, 5, 12, 2, 13, 9, 7, 10, 14, 12, 15, 10, 4, 1, 5, 8, 7, 6, 2, 13, 14, 0, 3, 9, 11]
SL = [11, 14, 15, 12, 5, 8, 7, 9, 11, 13, 14, 15, 6, 7, 9, 8, 7, 6, 8, 13, 11, 9, 7, 15, 7, 12, 15, 9, 11, 7, 13, 12, 11, 13, 6, 7, 14, 9, 13,
      15, 14, 8, 13, 6, 5, 12, 7, 5, 11, 12, 14, 15, 14, 15, 9, 8, 9, 14, 5, 6, 8, 6, 5, 12, 9, 15, 5, 11, 6, 8, 13, 12, 5, 12, 13, 14, 11, 8,
      5, 6]
SR = [8, 9, 9, 11, 13, 15, 15, 5, 7, 7, 8, 11, 14, 14, 12, 6, 9, 13, 15, 7, 12, 8, 9, 11, 7, 7, 12, 7, 6, 15, 13, 11, 9, 7, 15, 11, 8, 6, 6, 14,
      12, 13, 5, 14, 13, 13, 7, 5, 15, 5, 8, 11, 14, 14, 6, 14, 6, 9, 12, 9, 12, 5, 15, 8, 8, 5, 12, 9, 12, 5, 14, 6, 8, 13, 6, 5, 15, 13, 11,
      11]
KL = [0x00000000, 0x5A827999, 0x6ED9EBA1, 0x8F1BBCDC, 0xA953FD4E]
KR = [0x50A28BE6, 0x5C4DD124, 0x6D703EF3, 0x7A6D76E9, 0x00000000]
KECCAK_RHO = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]   # [x][y]
KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b, 0x0000000080000001,
             0x8000000080008081, 0x8000000000008009, 0x000000000000008a, 0x0000000000000088, 0x0000000080008009, 0x000000008000000a,
             0x000000008000808b, 0x800000000000008b, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
             0x000000000000800a, 0x800000008000000a, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
TT_L = [0x96, 0xCA, 0x59, 0xE4, 0x2D]   # f1..f5 as v_bitop3 truth tables (core/hash.h VG_RMD_F1..F5)
TT_R = [0x2D, 0xE4, 0x59, 0xCA, 0x96]


def rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M


def bswap(x):
    return int.from_bytes(x.to_bytes(4, "big"), "little")


def tt_eval(tt, a, b, c):
    """v_bitop3_b32: bit i of the result is bit (4a_i + 2b_i + c_i) of the truth table."""
    r = 0
    for i in range(8):
        if (tt >> i) & 1:
            r |= (a if i & 4 else ~a) & (b if i & 2 else ~b) & (c if i & 1 else ~c)
    return r & M


def known(v):
    return isinstance(v, int)


CLASS_MAX_RUN = 0   # (--class-max-run n: by_class changes class after n instructions if the other class has work)
CLASS_DISTANCE = 0  # (--class-distance d: by_class prefers instructions that do not read one of the last d results)
SPLIT_ADD3 = None   # (m, n): m of every n three-variable sums as two v_add_u32 (--split-add3 m:n)
SPLIT_KADD = None   # (m, n): m of every n `acc + v + K` sums as two v_add_u32 instead of s_mov + v_add3_u32 (--split-kadd m:n)


class Program:
    """A straight-line list of instructions over virtual registers.

    ins: (op, dst, srcs, imm) with op in alignbit(hi, lo; n) lshr(src; n) bitop3(a, b, c; tt) mov(k) add3(a, b, c) add(a, b) xor(a, b)
    bswap(src); a source is a virtual register name or an int (at most one int per instruction, never in lshr/bswap)."""

    def __init__(self, grouped=False):
        self.ins, self.inputs, self.outputs, self.n, self.grouped = [], [], [], 0, grouped

    def input(self, name):
        self.inputs.append(name)
        return name

    def emit(self, op, srcs, imm=None):
        self.n += 1
        d = f"t{self.n}"
        self.ins.append((op, d, tuple(srcs), imm))
        return d

    # ---- folding constructors ----------------------------------------------------------------------------------------
    def rotr(self, x, n):
        return rotr(x, n) if known(x) else self.emit("alignbit", (x, x), n)

    def funnel(self, hi, lo, n):
        """Low word of the 64-bit value hi:lo shifted right by n (0 < n < 32): v_alignbit_b32."""
        if known(hi) and known(lo):
            return ((hi << 32 | lo) >> n) & M
        return self.emit("alignbit", (hi, lo), n)

    def xor(self, a, b):
        if known(a) and known(b):
            return a ^ b
        if known(a):
            a, b = b, a
        if known(b) and b == 0:
            return a
        return self.emit("xor", (a, b))

    def rotl(self, x, n):
        return self.rotr(x, (32 - n) & 31) if n & 31 else x

    def shr(self, x, n):
        return x >> n if known(x) else self.emit("lshr", (x,), n)

    def bswap(self, x):
        return bswap(x) if known(x) else self.emit("bswap", (x,))

    def bitop3(self, tt, a, b, c):
        ops = [a, b, c]
        kn = [known(o) for o in ops]
        if all(kn):
            return tt_eval(tt, a, b, c)
        while sum(kn) > 1:           # a second constant needs a register of its own (first rounds on the IV only)
            i = kn.index(True)
            ops[i] = self.emit("mov", (ops[i],))
            kn[i] = False
        return self.emit("bitop3", ops, tt)

    def add(self, *terms):
        kc = sum(t for t in terms if known(t)) & M
        vs = [t for t in terms if not known(t)]
        if not vs:
            return kc
        acc, rest = vs[0], vs[1:]
        while len(rest) >= 2:
            self.add3s = getattr(self, "add3s", 0) + 1
            if SPLIT_ADD3 and self.add3s % SPLIT_ADD3[1] < SPLIT_ADD3[0]:      # (A/B: a three-input sum as two full-rate adds)
                acc = self.emit("add", (self.emit("add", (acc, rest[0])), rest[1]))
            else:
                acc = self.emit("add3", (acc, rest[0], rest[1]))
            rest = rest[2:]
        if rest and kc:
            # acc + v + K: one half-rate v_add3 behind an s_mov of K, or (SPLIT_KADD: every n-th of them) two full-rate adds, the second
            # with K as a literal - a place in the FIRST half of an issue slot traded for two in the second halves (see by_class)
            self.kadds = getattr(self, "kadds", 0) + 1
            if SPLIT_KADD and not inline_const(kc) and self.kadds % SPLIT_KADD[1] < SPLIT_KADD[0]:
                return self.emit("add", (self.emit("add", (acc, rest[0])), kc))
            return self.emit("add3", (acc, rest[0], kc))
        if rest:
            return self.emit("add", (acc, rest[0]))
        if kc:
            return self.emit("add", (acc, kc))
        return acc

    # ---- SHA-256: one compression of the 16 words w (ints or registers) into the chaining value st -------------------------
    def sha256_compress(self, st_in, w):
        w = list(w)
        st = list(st_in)
        for r in range(64):
            if r >= 16:
                i15, i2, i7, i0 = (r - 15) & 15, (r - 2) & 15, (r - 7) & 15, r & 15
                if self.grouped:
                    a7, a18 = self.rotr(w[i15], 7), self.rotr(w[i15], 18)
                    b17, b19 = self.rotr(w[i2], 17), self.rotr(w[i2], 19)
                    a3, b10 = self.shr(w[i15], 3), self.shr(w[i2], 10)
                    s0, s1 = self.bitop3(0x96, a7, a18, a3), self.bitop3(0x96, b17, b19, b10)
                else:
                    b17, b19, b10 = self.rotr(w[i2], 17), self.rotr(w[i2], 19), self.shr(w[i2], 10)
                    s1 = self.bitop3(0x96, b17, b19, b10)
                    a7, a18, a3 = self.rotr(w[i15], 7), self.rotr(w[i15], 18), self.shr(w[i15], 3)
                    s0 = self.bitop3(0x96, a7, a18, a3)
                w[i0] = self.add(w[i0], s1, w[i7], s0)
            a, b, c, d, e, f, g, h = (st[(j - r) & 7] for j in range(8))
            if self.grouped:
                r6, r11, r25 = self.rotr(e, 6), self.rotr(e, 11), self.rotr(e, 25)
                r2, r13, r22 = self.rotr(a, 2), self.rotr(a, 13), self.rotr(a, 22)
                s1, ch = self.bitop3(0x96, r6, r11, r25), self.bitop3(0xCA, e, f, g)
                s0, mj = self.bitop3(0x96, r2, r13, r22), self.bitop3(0xE8, a, b, c)
                t1 = self.add(h, s1, ch, K[r], w[r & 15])
                nh = self.add(t1, s0, mj)
                nd = self.add(d, t1)
            else:
                r6, r11, r25 = self.rotr(e, 6), self.rotr(e, 11), self.rotr(e, 25)
                s1, ch = self.bitop3(0x96, r6, r11, r25), self.bitop3(0xCA, e, f, g)
                t1 = self.add(h, s1, ch, K[r], w[r & 15])
                r2, r13, r22 = self.rotr(a, 2), self.rotr(a, 13), self.rotr(a, 22)
                s0, mj = self.bitop3(0x96, r2, r13, r22), self.bitop3(0xE8, a, b, c)
                nd = self.add(d, t1)
                nh = self.add(t1, s0, mj)
            st[(3 - r) & 7], st[(7 - r) & 7] = nd, nh
        return [self.add(st[i], st_in[i]) for i in range(8)]

    # ---- RIPEMD-160: one compression of the 16 little-endian words x into the IV -----------------------------------------
    def ripemd160_compress_iv(self, x):
        left, right = list(RMD_IV), list(RMD_IV)
        for j in range(80):
            g, o = j // 16, (5 - j % 5) % 5
            for v, tt, msg, kk, sh in ((left, TT_L[g], x[RL[j]], KL[g], SL[j]), (right, TT_R[g], x[RR[j]], KR[g], SR[j])):
                a, b, c, d, e = (v[(o + i) % 5] for i in range(5))
                f = self.bitop3(tt, b, c, d)
                s = self.add(a, f, msg, kk)
                if self.grouped:
                    nc = self.rotl(c, 10)
                    s = self.rotl(s, sh)
                    na = self.add(s, e)
                else:
                    s = self.rotl(s, sh)
                    na = self.add(s, e)
                    nc = self.rotl(c, 10)
                v[o], v[(o + 2) % 5] = na, nc
        return [self.add(RMD_IV[1], left[2], right[3]), self.add(RMD_IV[2], left[3], right[4]), self.add(RMD_IV[3], left[4], right[0]),
                self.add(RMD_IV[4], left[0], right[1]), self.add(RMD_IV[0], left[1], right[2])]

    def ripemd160_of_sha(self, sha):
        """RIPEMD-160 of a 32-byte SHA-256 digest given as eight big-endian words -> five little-endian words (memory order)."""
        x = [self.bswap(s) for s in sha] + [0x80, 0, 0, 0, 0, 0, 256, 0]
        return self.ripemd160_compress_iv(x)

    def prune(self):
        """Drop the instructions no output depends on (a digest of which only the first word is wanted)."""
        live = {o for o in self.outputs if not known(o)}
        keep = []
        for ins in reversed(self.ins):
            if ins[1] in live:
                keep.append(ins)
                live.update(x for x in ins[2] if not known(x))
        self.ins = keep[::-1]
        return self

    # ---- Keccak-f[1600] on 25 lanes of (lo, hi) word pairs ------------------------------------------------------------------
    def rotl64(self, lane, n):
        lo, hi = lane
        if n == 0:
            return lane
        if n == 32:
            return (hi, lo)
        m = n & 31
        a = self.funnel(hi, lo, 32 - m)     # (hi << m) | (lo >> (32 - m))
        b = self.funnel(lo, hi, 32 - m)     # (lo << m) | (hi >> (32 - m))
        return (b, a) if n < 32 else (a, b)

    def bitop3_64(self, tt, a, b, c):
        return (self.bitop3(tt, a[0], b[0], c[0]), self.bitop3(tt, a[1], b[1], c[1]))

    def keccak_f1600(self, a):
        """a: 25 lanes, index x + 5y.  Per round: the column parities as two xor3 each; `a ^ d` of every lane as one more xor3
        (d = c[x-1] ^ rotl(c[x+1], 1) is never materialised); rho and pi; chi as the truth table 0xD2; one output row at a time,
        so that a row's five source lanes die as they are read (core/hash.h keccak_round, restated)."""
        a = list(a)
        for rnd in range(24):
            c = [self.bitop3_64(0x96, self.bitop3_64(0x96, a[x], a[x + 5], a[x + 10]), a[x + 15], a[x + 20]) for x in range(5)]
            rc = [self.rotl64(c[x], 1) for x in range(5)]
            n = [None] * 25
            for Y in range(5):
                b = []
                for X in range(5):
                    y = X
                    x = ((Y - 3 * X) * 3) % 5            # pi: B[y][2x + 3y] = rot(A[x][y])
                    t = self.bitop3_64(0x96, a[x + 5 * y], c[(x + 4) % 5], rc[(x + 1) % 5])
                    b.append(self.rotl64(t, KECCAK_RHO[x][y]))
                for X in range(5):
                    n[X + 5 * Y] = self.bitop3_64(0xD2, b[X], b[(X + 1) % 5], b[(X + 2) % 5])
            n[0] = (self.xor(n[0][0], KECCAK_RC[rnd] & M), self.xor(n[0][1], KECCAK_RC[rnd] >> 32))
            a = n
        return a

    # ---- statistics ------------------------------------------------------------------------------------------------------
    def census(self):
        c = {}
        for op, _, _, _ in self.ins:
            c[op] = c.get(op, 0) + 1
        return c


# ---- the hashes of the path: (C prologue computing the inputs, program) -------------------------------------------------------

def prog_pub33_h160(grouped=False):
    """hash160 of the compressed key: inputs m0..m8 = the nine message words that depend on the key
    (core/hash.h sha256_pub33 + ripemd160_of_sha; payload of P2PKH / P2WPKH and first half of P2SH-P2WPKH)."""
    p = Program(grouped)
    w = [p.input(f"m{i}") for i in range(9)] + [0] * 6 + [33 * 8]
    p.outputs = p.ripemd160_of_sha(p.sha256_compress(SHA_IV, w))
    prologue = ["u32 m0 = (prefix << 24) | (xw[7] >> 8);"]
    prologue += [f"u32 m{i} = (xw[{8 - i}] << 24) | (xw[{7 - i}] >> 8);" for i in range(1, 8)]
    prologue += ["u32 m8 = (xw[0] << 24) | 0x00800000u;"]
    return p, "u32 prefix, const u32 xw[8]", prologue


def prog_script22_h160(grouped=False):
    """hash160 of the P2WPKH redeem script 0x00 0x14 || h160: inputs m0..m5 (core/hash.h sha256_script22 + ripemd160_of_sha)."""
    p = Program(grouped)
    w = [p.input(f"m{i}") for i in range(6)] + [0] * 9 + [22 * 8]
    p.outputs = p.ripemd160_of_sha(p.sha256_compress(SHA_IV, w))
    prologue = ["u32 b0 = bswap32(h[0]), b1 = bswap32(h[1]), b2 = bswap32(h[2]), b3 = bswap32(h[3]), b4 = bswap32(h[4]);",
                "u32 m0 = 0x00140000u | (b0 >> 16);"]
    prologue += [f"u32 m{i} = (b{i - 1} << 16) | (b{i} >> 16);" for i in range(1, 5)]
    prologue += ["u32 m5 = (b4 << 16) | 0x00008000u;"]
    return p, "const u32 h[5]", prologue


def prog_pub65_h160(grouped=False):
    """hash160 of the uncompressed key 0x04 || X || Y: two SHA-256 blocks (core/hash.h sha256_pub65 + ripemd160_of_sha)."""
    p = Program(grouped)
    w = [p.input(f"m{i}") for i in range(17)]
    mid = p.sha256_compress(SHA_IV, w[:16])
    sha = p.sha256_compress(mid, [w[16]] + [0] * 14 + [65 * 8])
    p.outputs = p.ripemd160_of_sha(sha)
    prologue = ["u32 m0 = (0x04u << 24) | (xw[7] >> 8);"]
    prologue += [f"u32 m{i} = (xw[{8 - i}] << 24) | (xw[{7 - i}] >> 8);" for i in range(1, 8)]
    prologue += ["u32 m8 = (xw[0] << 24) | (yw[7] >> 8);"]
    prologue += [f"u32 m{8 + i} = (yw[{8 - i}] << 24) | (yw[{7 - i}] >> 8);" for i in range(1, 8)]
    prologue += ["u32 m16 = (yw[0] << 24) | 0x00800000u;"]
    return p, "const u32 xw[8], const u32 yw[8]", prologue


def prog_base58_check(grouped=False):
    """Base58Check checksum: the first word of SHA-256(SHA-256(version || hash160)); inputs m0..m5, the six message words of the
    21 bytes (core/dfa_eval.h base58_checksum; the on-device match of unanchored patterns needs the whole address string)."""
    p = Program(grouped)
    w = [p.input(f"m{i}") for i in range(6)] + [0] * 9 + [21 * 8]
    first = p.sha256_compress(SHA_IV, w)
    p.outputs = p.sha256_compress(SHA_IV, first + [0x80000000] + [0] * 6 + [32 * 8])[:1]
    p.prune()
    prologue = ["u32 m0 = (version << 24) | (H[0] >> 8);"]
    prologue += [f"u32 m{i} = (H[{i - 1}] << 24) | (H[{i}] >> 8);" for i in range(1, 5)]
    prologue += ["u32 m5 = (H[4] << 24) | 0x00800000u;"]
    return p, "u32 version, const u32 H[5]", prologue


def prog_keccak_addr(grouped=False):
    """Ethereum address: the low 20 bytes of Keccak-256(X || Y); inputs m0..m15 = the 16 message words, lane i = (m[2i], m[2i+1])
    (core/hash.h keccak256_pub64_addr: one rate block, pre-SHA-3 padding); out = five words in memory order."""
    p = Program(grouped)
    lanes = [(p.input(f"m{2 * i}"), p.input(f"m{2 * i + 1}")) for i in range(8)] + [(0, 0)] * 17
    lanes[8] = (1, 0)
    lanes[16] = (0, 0x80000000)
    a = p.keccak_f1600(lanes)
    p.outputs = [a[1][1], a[2][0], a[2][1], a[3][0], a[3][1]]
    p.prune()
    prologue = []
    for i in range(4):
        prologue.append(f"u32 m{2 * i} = bswap32(xw[{7 - 2 * i}]), m{2 * i + 1} = bswap32(xw[{6 - 2 * i}]);")
    for i in range(4):
        prologue.append(f"u32 m{8 + 2 * i} = bswap32(yw[{7 - 2 * i}]), m{9 + 2 * i} = bswap32(yw[{6 - 2 * i}]);")
    return p, "const u32 xw[8], const u32 yw[8]", prologue


# keccak_addr_block: in round 4, in dependency order, the block ran no faster than hipcc's rolled rounds (same 5 003 instructions per key); as runs by
# issue class with the priority changes it is worth +28 % on every Ethereum configuration (profiles/r05_keccak_ab.txt): 1 351 half-rate funnel
# shifts against 2 844 full-rate booleans, which now ride in the second places of the issue slots.
PROGRAMS = {"hash160_pub33_block": prog_pub33_h160, "hash160_script22_block": prog_script22_h160,
            "hash160_pub65_block": prog_pub65_h160, "base58_check_block": prog_base58_check, "keccak_addr_block": prog_keccak_addr}
OPTIONAL = {}    # (--with NAME: blocks generated for an A/B only)
# yields per function where they differ from the default
YIELDS = {"keccak_addr_block": "none"}


# ---- the Python model of the instruction list (CPU tests) ---------------------------------------------------------------------

def evaluate(p, inputs):
    """Run program p on concrete input words (dict name -> int); -> list of output words."""
    v = dict(inputs)

    def val(s):
        return s if known(s) else v[s]

    for op, d, srcs, imm in p.ins:
        s = [val(x) for x in srcs]
        if op == "alignbit":
            r = ((s[0] << 32 | s[1]) >> imm) & M
        elif op == "xor":
            r = s[0] ^ s[1]
        elif op == "lshr":
            r = s[0] >> imm
        elif op == "bitop3":
            r = tt_eval(imm, *s)
        elif op == "mov":
            r = s[0]
        elif op in ("add3", "add"):
            r = sum(s) & M
        elif op == "bswap":
            r = bswap(s[0])
        else:
            raise ValueError(op)
        v[d] = r
    return [val(o) for o in p.outputs]


def evaluate_allocated(p, reg, nreg, inputs):
    """The same on the PHYSICAL registers `allocate` chose (what the asm statement does): -> list of output words."""
    rf = [None] * nreg
    for name in p.inputs:
        rf[reg[name]] = inputs[name]
    q = Program()
    q.ins = [(op, ("r", reg[d]), tuple(s if known(s) else ("r", reg[s]) for s in srcs), imm) for op, d, srcs, imm in p.ins]
    for op, (_, d), srcs, imm in q.ins:
        one = Program()
        one.ins = [(op, "d", tuple(s if known(s) else f"s{i}" for i, s in enumerate(srcs)), imm)]
        one.outputs = ["d"]
        rf[d] = evaluate(one, {f"s{i}": rf[s[1]] for i, s in enumerate(srcs) if not known(s)})[0]
    return [o if known(o) else rf[reg[o]] for o in p.outputs]


# ---- instruction order ----------------------------------------------------------------------------------------------------------

def spread(p, window=24, distance=1):
    """Reorder p.ins so that no instruction reads a result written by one of the `distance` instructions before it, wherever the
    dependency graph allows.  Greedy list scheduling over a window of the original order (which bounds the extra register
    pressure): the earliest complete instruction among those farthest from their operands goes next.  An option of the A/B,
    not the default: on the MI355X it changes nothing (12.62 against 12.62 Gkeys/s without yields, 13.10 against 13.14 with).
    -> number of dependent neighbours that remain."""
    ins = p.ins
    n = len(ins)
    producer = {d: i for i, (_, d, _, _) in enumerate(ins)}
    preds = [[producer[s] for s in srcs if not known(s) and s in producer] for _, _, srcs, _ in ins]
    done = [False] * n
    pos = {}                    # original index -> position in the new order
    order = []
    head = 0
    remaining = 0
    while len(order) < n:
        while head < n and done[head]:
            head += 1
        best, pick = -1, None          # the earliest complete instruction among those farthest (up to `distance`) from their operands
        for i in range(head, min(n, head + window)):
            if done[i] or not all(done[q] for q in preds[i]):
                continue
            gap = min([len(order) - pos[q] - 1 for q in preds[i]] + [distance])
            if gap > best:
                best, pick = gap, i
                if gap == distance:
                    break
        if best == 0:
            remaining += 1
        done[pick] = True
        pos[pick] = len(order)
        order.append(pick)
    p.ins = [ins[i] for i in order]
    return remaining


# Instruction classes of the gfx950 issue stage (tools/ubench_phase*.hip, profiles/r05_phase*_ubench.jsonl): a SIMD fills one
# 4-cycle slot with the next instruction of its highest-priority (then oldest) ready wave and, if that leaves room, with a
# FULL-RATE instruction (VOP2 add / logic / shift, v_bitop3) of another wave.  A half-rate instruction (v_alignbit, v_add3,
# v_perm) can only take the first place, so in a mixed stream under age-only arbitration the oldest wave owns that place and
# every instruction of every wave costs a slot of its own: 4.0 cycles, the full-rate ones included.
HALF_RATE = {"alignbit", "add3", "bswap"}


def by_class(p, window=40, distance=0):
    """Reorder p.ins into alternating runs of half-rate and full-rate instructions: greedy list scheduling that stays in the
    current class while any instruction of it within `window` places of the oldest unscheduled one has its operands ready
    (the window bounds the extra register pressure).  asm_lines(prio=...) then raises the wave's priority for the half-rate
    runs, which gives them the slots' first places while other waves' full-rate runs fill the second ones.
    -> list of run lengths."""
    ins = p.ins
    n = len(ins)
    producer = {d: i for i, (_, d, _, _) in enumerate(ins)}
    preds = [[producer[s] for s in srcs if not known(s) and s in producer] for _, _, srcs, _ in ins]
    done = [False] * n
    order, runs, pos = [], [], {}
    head, cls = 0, None
    while len(order) < n:
        while head < n and done[head]:
            head += 1
        ready = [i for i in range(head, min(n, head + window)) if not done[i] and all(done[q] for q in preds[i])]
        same = [i for i in ready if (ins[i][0] in HALF_RATE) == cls]
        if same and CLASS_MAX_RUN and runs and runs[-1] >= CLASS_MAX_RUN and len(same) < len(ready):
            same = []      # (A/B only: cap the run although more of its class are ready)
        if same:
            # (distance > 0, A/B only: prefer an instruction that does not read one of the last `distance` results)
            far = [i for i in same if all(len(order) - pos[q] > distance for q in preds[i])] if distance else same
            pick = (far or same)[0]
            runs[-1] += 1
        else:
            pick = ready[0]
            cls = ins[pick][0] in HALF_RATE
            runs.append(1)
        done[pick] = True
        pos[pick] = len(order)
        order.append(pick)
    p.ins = [ins[i] for i in order]
    return runs


# ---- register allocation and the asm text -------------------------------------------------------------------------------------

YIELD_INSN = "s_nop 0"     # (--yield-insn: A/B of other ways to give the slot away)


def inline_const(k):
    """gfx9 inline integer constants: 0..64 and -16..-1."""
    return k <= 64 or k >= (M + 1 - 16)


def allocate(p):
    """Linear scan over the straight-line list.  -> (reg of every virtual name, number of registers).
    Inputs occupy r0..r(n-1) on entry; a register is free again after the last read of its value, and the instruction that
    reads it last may write its own result there (in-order issue: the read precedes the write)."""
    last = {}
    for i, (_, _, srcs, _) in enumerate(p.ins):
        for s in srcs:
            if not known(s):
                last[s] = i
    for o in p.outputs:
        if not known(o):
            last[o] = len(p.ins)
    reg, free, nreg = {}, [], 0
    for name in p.inputs:
        reg[name] = nreg
        nreg += 1
    for name in p.inputs:
        if name not in last:
            free.append(reg[name])
    for i, (_, d, srcs, _) in enumerate(p.ins):
        for s in dict.fromkeys(srcs):
            if not known(s) and last[s] == i:
                free.append(reg[s])
        if d not in last:
            raise ValueError(f"dead instruction {i}: {p.ins[i]}")
        if free:
            reg[d] = free.pop(0)       # oldest free register first: spreads the writes over the pool
        else:
            reg[d] = nreg
            nreg += 1
    return reg, nreg


def asm_lines(p, reg, yields="every:3", prio=None):
    """-> (list of asm lines with %[rN] / %[k] operands, VALU count, s_mov count, yield count).
    yields: "every:N" puts an `s_nop 0` after every N-th VALU instruction, "dep" between an instruction and a successor that
    reads its result (what hipcc does around single-instruction asm statements), "none" nowhere.
    prio: None, or (p_half, p_full[, p_exit]): an `s_setprio` wherever the stream changes class (by_class made the runs), p_exit (0) at the end."""
    out, valu, salu, nyield = [], 0, 0, 0
    cur_class = None
    prev_dst = None
    last_yield_at = last_salu_at = 0

    def r(s):
        return f"%[r{reg[s]}]"

    def const_operand(k):
        nonlocal salu
        if inline_const(k):
            return str(k if k <= 64 else k - (M + 1))
        out.append(f"s_mov_b32 %[k], 0x{k:08x}")
        salu += 1
        return "%[k]"

    for op, d, srcs, imm in p.ins:
        D = r(d)
        if yields.startswith("every:"):
            # "every:N": after every N-th VALU instruction; "every:N:salu": N VALU instructions after the last yield OR s_mov
            # (a scalar instruction of the wave's own gives the slot away just as well)
            parts = yields.split(":")
            if len(parts) == 4:          # "every:N:from:to": only between VALU instructions `from` and `to` (A/B of the two hashes of a pair)
                want = prev_dst is not None and int(parts[2]) <= valu < int(parts[3]) and valu % int(parts[1]) == 0
            elif len(parts) == 3:
                since = valu - max(last_yield_at, last_salu_at)
                want = prev_dst is not None and since >= int(parts[1])
            else:
                want = prev_dst is not None and valu % int(parts[1]) == 0
        else:
            want = yields == "dep" and prev_dst is not None and prev_dst in srcs
        if want:
            out.append(YIELD_INSN)
            nyield += 1
            last_yield_at = valu
        prev_dst = d
        if prio is not None and (op in HALF_RATE) != cur_class:
            cur_class = op in HALF_RATE
            out.append(f"s_setprio {prio[0] if cur_class else prio[1]}")
        salu_before = salu
        if op == "alignbit":
            o = [const_operand(x) if known(x) else r(x) for x in srcs]
            line = f"v_alignbit_b32 {D}, {o[0]}, {o[1]}, {imm}"
        elif op == "xor":        # VOP2: a 32-bit literal is allowed in src0
            a, b = srcs
            line = f"v_xor_b32 {D}, {('0x%08x' % b) if known(b) else r(b)}, {r(a)}"
        elif op == "lshr":
            line = f"v_lshrrev_b32 {D}, {imm}, {r(srcs[0])}"
        elif op == "mov":
            line = f"v_mov_b32 {D}, 0x{srcs[0]:08x}"
        elif op == "bswap":
            line = f"v_perm_b32 {D}, 0, {r(srcs[0])}, {const_operand(0x00010203)}"
        elif op == "bitop3":
            o = [const_operand(s) if known(s) else r(s) for s in srcs]
            line = f"v_bitop3_b32 {D}, {o[0]}, {o[1]}, {o[2]} bitop3:0x{imm:02x}"
        elif op == "add3":
            o = [const_operand(s) if known(s) else r(s) for s in srcs]
            line = f"v_add3_u32 {D}, {o[0]}, {o[1]}, {o[2]}"
        elif op == "add":
            a, b = srcs
            if known(b):
                a, b = b, a
            if known(a):     # VOP2: a 32-bit literal is allowed in src0
                line = f"v_add_u32 {D}, {a if a <= 64 else ('0x%08x' % a)}, {r(b)}"
            else:
                line = f"v_add_u32 {D}, {r(a)}, {r(b)}"
        else:
            raise ValueError(op)
        out.append(line)
        valu += 1
        if salu != salu_before:
            last_salu_at = valu - 1      # the s_mov sits in front of the instruction just emitted
    if prio is not None:
        out.append(f"s_setprio {prio[2] if len(prio) > 2 else 0}")     # the level the caller runs at
    return out, valu, salu, nyield


FILLER = None   # (--filler seq:N | mix:N, micro-benchmark only: N multiply-adds of four independent chains after / spread through hash160_pub33_block)


def add_filler(lines, mode, n):
    """tools/ubench_hash_yield.hip: what the point arithmetic's multiply-adds cost beside the hash - in a phase of their own or mixed in."""
    mads = [f"v_mad_u64_u32 %[p{i % 4}], vcc, %[q{i % 2}], %[q{(i // 2) % 2}], %[p{i % 4}]" for i in range(n)]
    if mode == "seq":
        return lines + mads
    out, step, k = [], max(1, len(lines) // max(1, n)), 0
    for i, l in enumerate(lines):
        out.append(l)
        if (i + 1) % step == 0 and k < n and l.startswith("v_"):     # never between an s_mov and the instruction that reads it
            out.append(mads[k])
            k += 1
    return out + mads[k:]


def function_source(name, grouped=False, yields="every:3", window=0, distance=1, prio=None, class_window=0):
    p, params, prologue = {**PROGRAMS, **OPTIONAL}[name](grouped)
    left = spread(p, window, distance) if window else None
    runs = by_class(p, class_window, CLASS_DISTANCE) if class_window else None
    reg, nreg = allocate(p)
    lines, valu, salu, nyield = asm_lines(p, reg, yields, prio)
    filler = FILLER if name == "hash160_pub33_block" else None
    if filler:
        lines = add_filler(lines, filler[0], filler[1])
        params += ", u32 *fill"
    nin = len(p.inputs)
    nout = len(p.outputs)
    c = p.census()
    src = f"// {name}: {valu} VALU ({', '.join(f'{c[k]} {k}' for k in sorted(c))}) + {salu} s_mov + {nyield} yields,\n"
    src += f"// {nreg} VGPRs + 1 SGPR, {'grouped' if grouped else 'dependency'} order"
    src += f", spread over a window of {window}: {left} dependent neighbours left" if window else ""
    if runs:
        src += f";\n// runs by issue class over a window of {class_window}: {len(runs)} runs, mean {sum(runs) / len(runs):.1f}, s_setprio {prio}"
    src += ".\n"
    src += f"__device__ __forceinline__ void {name}({params}, u32 out[{nout}]) {{\n"
    for l in prologue:
        src += f"    {l}\n"
    for i, nm in enumerate(p.inputs):
        src += f"    u32 r{i} = {nm};\n"
    if nreg > nin:
        src += "    u32 " + ", ".join(f"r{i}" for i in range(nin, nreg)) + ";\n"
    src += "    u32 k;\n"
    if filler:
        src += "    u32 q0 = prefix | 1u, q1 = xw[0] | 1u;\n    unsigned long long p0 = q0, p1 = q1, p2 = q0 + 2u, p3 = q1 + 2u;\n"
    src += "    asm(\n"
    for l in lines:
        src += f'        "{l}\\n\\t"\n'
    ops = [f'[r{i}] "+v"(r{i})' for i in range(nin)] + [f'[r{i}] "=&v"(r{i})' for i in range(nin, nreg)] + ['[k] "=&s"(k)']
    if filler:
        ops += [f'[p{i}] "+v"(p{i})' for i in range(4)]
        src += "        : " + ", ".join(ops) + ' : [q0] "v"(q0), [q1] "v"(q1) : "vcc");\n'
        src += "    fill[0] ^= (u32)(p0 ^ p1 ^ p2 ^ p3) ^ (u32)((p0 ^ p1 ^ p2 ^ p3) >> 32);\n"
    else:
        src += "        : " + ", ".join(ops) + ");\n"
    for j, o in enumerate(p.outputs):
        src += f"    out[{j}] = {('0x%08xu' % o) if known(o) else 'r%d' % reg[o]};\n"
    src += "}\n\n"
    return src


# What the Makefile builds (round 5): runs by issue class over a window of 4 (29 VGPRs), half-rate runs at wave priority 1, full-rate runs at 0,
# back to 1 - the level the scan kernels run their point arithmetic at - on the way out; no yields (a priority change gives the slot away as well).
# The round-4 blocks are `--class-window 0 --prio none --yield every:3`.
DEFAULT_CLASS_WINDOW, DEFAULT_PRIO, DEFAULT_YIELD = 4, "1:0:1", "none"


def generate(grouped=False, yields=DEFAULT_YIELD, window=0, distance=1, overrides=None, extra=(),
             prio=tuple(int(x) for x in DEFAULT_PRIO.split(":")), class_window=DEFAULT_CLASS_WINDOW):
    src = "// GENERATED by device/hashgen.py (`make -C vgen_amd/csrc hashblocks`) - do not edit.\n"
    src += "// The address hashes of the scan kernels as single asm statements of gfx950 instructions; see hashgen.py.\n\n"
    for name in list(PROGRAMS) + list(extra):
        src += function_source(name, grouped, (overrides or {}).get(name, YIELDS.get(name, yields)), window, distance, prio, class_window)
    return src


def main(argv):
    def opt(name, default):
        return argv[argv.index(name) + 1] if name in argv else default

    global YIELD_INSN, FILLER, SPLIT_KADD, SPLIT_ADD3, CLASS_DISTANCE, CLASS_MAX_RUN
    if "--split-add3" in argv:
        SPLIT_ADD3 = tuple(int(x) for x in opt("--split-add3", "").split(":"))
    CLASS_MAX_RUN = int(opt("--class-max-run", "0"))
    CLASS_DISTANCE = int(opt("--class-distance", "0"))
    if "--split-kadd" in argv:
        SPLIT_KADD = tuple(int(x) for x in opt("--split-kadd", "").split(":"))
    YIELD_INSN = opt("--yield-insn", YIELD_INSN)
    if "--filler" in argv:
        mode, n = opt("--filler", "").split(":")
        FILLER = (mode, int(n))
    overrides = dict(a.split("=", 1) for i, a in enumerate(argv) if i and argv[i - 1] == "--yield-for")   # --yield-for name=mode
    sys.stdout.write(generate(opt("--order", "natural") == "grouped", opt("--yield", DEFAULT_YIELD),
                              int(opt("--window", "0")), int(opt("--distance", "1")), overrides,
                              [a for i, a in enumerate(argv) if i and argv[i - 1] == "--with"],
                              None if opt("--prio", DEFAULT_PRIO) == "none" else tuple(int(x) for x in opt("--prio", DEFAULT_PRIO).split(":")),
                              int(opt("--class-window", str(DEFAULT_CLASS_WINDOW)))))


if __name__ == "__main__":
    main(sys.argv[1:])
