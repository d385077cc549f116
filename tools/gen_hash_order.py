"""Generator of tools/hash_order_gen.inc: SHA-256(33-byte key) -> RIPEMD-160 as a flat list of pinned (asm volatile)
gfx950 instructions, in several ORDERS of the same instruction multiset, for tools/ubench_hash_order.hip.

The generator executes the two compressions symbolically: a value is a Python int (known at generation time: the padded
message words, the IVs, everything that depends only on them) or the name of a register variable.  Known values fold
exactly as hipcc folds them in core/hash.h, so the pinned variants run the product's instruction multiset (2 211 VALU
per hash pair; tools/hash_order_census.py counts both).  Every emitted instruction belongs to a BLOCK (one round's
half-rate burst, its full-rate part, ...); orders are made by arranging blocks and, for two keys per lane, by merging the
two chains' instruction lists.

  natural      one chain, dependency order as the round macros of core/hash.h write it
  grouped      one chain, per round: rotates (v_alignbit) first, then booleans (v_bitop3), then the adds
  x2           two chains merged instruction by instruction (natural order each)
  x2_grouped   two chains, per round: both chains' rotates, both chains' booleans, both chains' adds

usage: python tools/gen_hash_order.py   (writes tools/hash_order_gen.inc — ~750 KB of generated code, not kept in the repository)
Reference for the algorithms: src/shaders/sha256.wgsl:43-170, src/shaders/ripemd160.wgsl:10-100 (via core/hash.h)."""
import os

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
TT_L = [0x96, 0xCA, 0x59, 0xE4, 0x2D]
TT_R = [0x2D, 0xE4, 0x59, 0xCA, 0x96]


def rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M


def tt_eval(tt, a, b, c):
    r = 0
    for i in range(8):
        if (tt >> i) & 1:
            r |= (a if i & 4 else ~a) & (b if i & 2 else ~b) & (c if i & 1 else ~c)
    return r & M


class Chain:
    """Symbolic executor of one key's hash pair.  self.blocks: list of lists of C statements (one pinned instruction each)."""

    def __init__(self, tag, grouped):
        self.tag, self.grouped, self.n = tag, grouped, 0
        self.blocks = [[]]
        self.decl = []

    def new_block(self):
        if self.blocks[-1]:
            self.blocks.append([])

    def tmp(self):
        self.n += 1
        return f"{self.tag}{self.n}"

    def emit(self, expr):
        t = self.tmp()
        self.blocks[-1].append(f"const u32 {t} = {expr};")
        return t

    def rotr(self, x, n):
        return rotr(x, n) if isinstance(x, int) else self.emit(f"i_alignbit({x}, {x}, {n})")

    def shr(self, x, n):
        return x >> n if isinstance(x, int) else self.emit(f"i_lshr({x}, {n})")

    def bitop3(self, tt, a, b, c):
        ops = [a, b, c]
        known = [isinstance(o, int) for o in ops]
        if all(known):
            return tt_eval(tt, a, b, c)
        if sum(known) == 2:   # the second constant needs a register (first SHA rounds only)
            i = known.index(True)
            ops[i] = self.emit(f"i_mov(0x{ops[i]:08x}u)")
            known[i] = False
        if any(known):
            i = known.index(True)
            args = ", ".join(f"0x{o:08x}u" if isinstance(o, int) else o for o in ops)
            return self.emit(f"i_bitop3_s{i}<0x{tt:02X}>({args})")
        return self.emit(f"i_bitop3<0x{tt:02X}>({a}, {b}, {c})")

    def add(self, *terms):
        kc = sum(t for t in terms if isinstance(t, int)) & M
        vs = [t for t in terms if not isinstance(t, int)]
        if not vs:
            return kc
        acc = vs[0]
        rest = vs[1:]
        while len(rest) >= 2:
            acc = self.emit(f"i_add3({acc}, {rest[0]}, {rest[1]})")
            rest = rest[2:]
        if rest and kc:
            return self.emit(f"i_add3k({acc}, {rest[0]}, 0x{kc:08x}u)")
        if rest:
            return self.emit(f"i_add({acc}, {rest[0]})")
        if kc:
            return self.emit(f"i_addk({acc}, 0x{kc:08x}u)")
        return acc

    # ---- SHA-256 of prefix || X -----------------------------------------------------------------------------------
    def sha256_pub33(self, prefix, xw):
        t = self.tag
        w = []
        self.blocks[-1].append(f"const u32 {t}m0 = ({prefix} << 24) | ({xw}[7] >> 8);")
        w.append(f"{t}m0")
        for i in range(1, 8):
            self.blocks[-1].append(f"const u32 {t}m{i} = ({xw}[{8 - i}] << 24) | ({xw}[{7 - i}] >> 8);")
            w.append(f"{t}m{i}")
        self.blocks[-1].append(f"const u32 {t}m8 = ({xw}[0] << 24) | 0x00800000u;")
        w.append(f"{t}m8")
        w += [0] * 6 + [33 * 8]
        st = list(SHA_IV)
        for r in range(64):
            self.new_block()
            if r >= 16:
                i15, i2, i7, i0 = (r - 15) & 15, (r - 2) & 15, (r - 7) & 15, r & 15
                if self.grouped:
                    a7, a18 = self.rotr(w[i15], 7), self.rotr(w[i15], 18)
                    b17, b19 = self.rotr(w[i2], 17), self.rotr(w[i2], 19)
                    self.new_block()
                    a3, b10 = self.shr(w[i15], 3), self.shr(w[i2], 10)
                    s0, s1 = self.bitop3(0x96, a7, a18, a3), self.bitop3(0x96, b17, b19, b10)
                else:
                    b17, b19, b10 = self.rotr(w[i2], 17), self.rotr(w[i2], 19), self.shr(w[i2], 10)
                    s1 = self.bitop3(0x96, b17, b19, b10)
                    a7, a18, a3 = self.rotr(w[i15], 7), self.rotr(w[i15], 18), self.shr(w[i15], 3)
                    s0 = self.bitop3(0x96, a7, a18, a3)
                w[i0] = self.add(w[i0], s1, w[i7], s0)
                self.new_block()
            a, b, c, d, e, f, g, h = (st[(j - r) & 7] for j in range(8))
            if self.grouped:
                r6, r11, r25 = self.rotr(e, 6), self.rotr(e, 11), self.rotr(e, 25)
                r2, r13, r22 = self.rotr(a, 2), self.rotr(a, 13), self.rotr(a, 22)
                self.new_block()
                s1, ch = self.bitop3(0x96, r6, r11, r25), self.bitop3(0xCA, e, f, g)
                s0, mj = self.bitop3(0x96, r2, r13, r22), self.bitop3(0xE8, a, b, c)
                self.new_block()
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
        self.new_block()
        return [self.add(st[i], SHA_IV[i]) for i in range(8)]

    # ---- RIPEMD-160 of the 32-byte digest -------------------------------------------------------------------------
    def ripemd160_of_sha(self, sha):
        t = self.tag
        self.new_block()
        x = []
        for i in range(8):
            self.blocks[-1].append(f"const u32 {t}x{i} = bswap32({sha[i]});")
            x.append(f"{t}x{i}")
        x += [0x80, 0, 0, 0, 0, 0, 256, 0]
        left, right = list(RMD_IV), list(RMD_IV)
        for j in range(80):
            g, o = j // 16, (5 - j % 5) % 5
            for v, tt, msg, kk, sh in ((left, TT_L[g], x[RL[j]], KL[g], SL[j]), (right, TT_R[g], x[RR[j]], KR[g], SR[j])):
                self.new_block()
                a, b, c, d, e = (v[(o + i) % 5] for i in range(5))
                f = self.bitop3(tt, b, c, d)
                if self.grouped:
                    self.new_block()
                    s = self.add(a, f, msg, kk)
                    nc = self.rotr(c, 22)
                    s = self.rotr(s, 32 - sh)
                    na = self.add(s, e)
                else:
                    s = self.add(a, f, msg, kk)
                    s = self.rotr(s, 32 - sh)
                    na = self.add(s, e)
                    nc = self.rotr(c, 22)
                v[o], v[(o + 2) % 5] = na, nc
        self.new_block()
        return [self.add(RMD_IV[1], left[2], right[3]), self.add(RMD_IV[2], left[3], right[4]), self.add(RMD_IV[3], left[4], right[0]),
                self.add(RMD_IV[4], left[0], right[1]), self.add(RMD_IV[0], left[1], right[2])]


def hash_pair(tag, grouped, prefix, xw):
    c = Chain(tag, grouped)
    sha = c.sha256_pub33(prefix, xw)
    out = c.ripemd160_of_sha(sha)
    return c, out


def function(name, chains, grouped):
    """-> C source of `__device__ __forceinline__ void name(const u32 *prefix, const u32 (*xw)[8], u32 (*out)[5])`."""
    cs = [hash_pair(f"c{i}_", grouped, f"prefix[{i}]", f"xw[{i}]") for i in range(chains)]
    body = []
    if chains == 1:
        for blk in cs[0][0].blocks:
            body += blk
    elif grouped:   # block by block: both chains' rotates, both chains' booleans, ...
        nb = max(len(c.blocks) for c, _ in cs)
        for i in range(nb):
            for c, _ in cs:
                if i < len(c.blocks):
                    body += c.blocks[i]
    else:           # instruction by instruction
        flat = [[s for blk in c.blocks for s in blk] for c, _ in cs]
        for i in range(max(len(f) for f in flat)):
            for f in flat:
                if i < len(f):
                    body.append(f[i])
    for i, (_, out) in enumerate(cs):
        for j, o in enumerate(out):
            body.append(f"out[{i}][{j}] = {o if not isinstance(o, int) else hex(o) + 'u'};")
    n = sum(1 for s in body if "i_" in s)
    src = f"// {name}: {chains} chain(s), {'grouped' if grouped else 'natural'} order, {n} pinned instructions\n"
    src += f"__device__ __forceinline__ void {name}(const u32 *prefix, const u32 (*xw)[8], u32 (*out)[5]) {{\n"
    src += "".join(f"    {s}\n" for s in body) + "}\n\n"
    return src


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hash_order_gen.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_hash_order.py — do not edit.  Included by tools/ubench_hash_order.hip.\n\n")
        f.write(function("hashpair_natural", 1, False))
        f.write(function("hashpair_grouped", 1, True))
        f.write(function("hashpair_x2", 2, False))
        f.write(function("hashpair_x2_grouped", 2, True))
    print("wrote", out)


if __name__ == "__main__":
    main()
