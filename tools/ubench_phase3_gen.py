"""Writes tools/ubench_phase3_blocks.inc for tools/ubench_phase3.hip: asm statements of 1 024 VALU instructions over eight independent chains,
with wave-priority changes / yields between them.  Pattern letters: a v_add_u32, r v_alignbit_b32, m v_mad_u64_u32, b v_bitop3_b32, 3 v_add3_u32 (VALU);
P s_setprio 1, Q s_setprio 2, p s_setprio 0, n s_nop 0 (not counted)."""
import sys
N = 1024
OPS = {"a": "v_add_u32 %{c}, %{c}, %16", "r": "v_alignbit_b32 %{c}, %{c}, %{c}, 7", "m": "v_mad_u64_u32 %{a}, vcc, %16, %17, %{a}",
       "b": "v_bitop3_b32 %{c}, %{c}, %16, %17 bitop3:0x96", "3": "v_add3_u32 %{c}, %{c}, %16, %17",
       "l": "v_add_u32 %{c}, 0x5a827999, %{c}",                                         # VOP2 with a 32-bit literal (RIPEMD-160's K)
       "k": "s_mov_b32 s30, 0x71374491\\n\\tv_add3_u32 %{c}, %{c}, %16, s30",           # SHA-256's K through an SGPR
       "s": "v_lshrrev_b32 %{c}, 3, %{c}", "x": "v_xor_b32 %{c}, %{c}, %16"}
CTL = {"P": "s_setprio 1", "Q": "s_setprio 2", "p": "s_setprio 0", "n": "s_nop 0"}
def block(pattern, chains=8):
    out, u, i = [], 0, 0
    while u < N:
        ch = pattern[i % len(pattern)]
        i += 1
        if ch in CTL:
            out.append(CTL[ch])
            continue
        c = u % chains
        out.append(OPS[ch].format(c=c, a=c + 8))
        u += 1
    out.append("s_setprio 0")
    return "\\n\\t".join(out)
pats = {
    "A3R1": "aaar", "A3R1_P": "aaaPrp", "A3R1_N": "aaanr",
    "A1R1": "ar", "A1R1_P": "aPrp", "A1R1_N": "arn",
    "A4R4": "aaaarrrr", "A4R4_P": "aaaaPrrrrp", "A4R4_LO": "PaaaaprrrrP",
    "A16R16": "a" * 16 + "r" * 16, "A16R16_P": "a" * 16 + "P" + "r" * 16 + "p",
    "A1R3_P": "aPrrrp",
    # a SHA-256-round-like multiset: 6 alignbit, 4 bitop3, 3 add3, 1 add, in dependency-like order
    "SHA": "rrrbb3rrrbb3a3", "SHA_P": "Prrrpbb" + "P3rrrpbb" + "P3pa" + "P3p", "SHA_N3": "rrrnbb3nrrrnbb3na3n",
    "SHA_P2": "Prrrp" + "bb" + "P3rrrp" + "bba" + "P33p",
    # grouped: all complex of two rounds together, then all simple
    "SHA_G": "P" + "rrrrrr3333rrrrrr33" + "p" + "bbbbbbbbaa",
    "M3A1": "mmma", "M3A1_P": "Pmmmpa", "M1A1_P": "Pmpa", "M1A3_P": "Pmpaaa", "M1A3": "maaa",
    "M1B1R1_P": "PmrpbPmrpa",
}
# the same streams over 1 / 2 dependency chains instead of 8 (does a wave that must wait for its own last result lose its place?)
DEP = {"R": "r", "A": "a", "A1R1_P": pats["A1R1_P"], "A4R4_P": pats["A4R4_P"], "SHA_P2": pats["SHA_P2"], "SHA_G": pats["SHA_G"], "SHA": pats["SHA"]}
with open(sys.argv[1], "w") as f:
    if len(sys.argv) > 2 and sys.argv[2] == "ops":
        pats = {"A1R1_P": "aPrp", "L1R1_P": "lPrp", "B1R1_P": "bPrp", "S1R1_P": "sPrp", "X1R1_P": "xPrp", "A1K1_P": "aPkp", "A13_P": "aP3p",
                "A3R3_P": "aaaPrrrp", "L3R3_P": "lllPrrrp", "B3R3_P": "bbbPrrrp", "A3K3_P": "aaaPkkkp", "B3K3_P": "bbbPkkkp",
                "BAL_R3K_P": "balPrr3kp", "L": "l", "B": "b", "K": "k", "AB": "ab", "AL": "al", "BL": "bl"}
        for k, p in pats.items():
            f.write('#define BLK_%s "%s"\n' % (k, block(p)))
    elif len(sys.argv) > 2 and sys.argv[2] == "dep":
        pats = {}
        for k, p in DEP.items():
            for ch in (1, 2, 4, 8):
                pats["%s_D%d" % (k, ch)] = (p, ch)
        for k, (p, ch) in pats.items():
            f.write('#define BLK_%s "%s"\n' % (k, block(p, ch)))
    else:
        for k, p in pats.items():
            f.write('#define BLK_%s "%s"\n' % (k, block(p)))
    f.write("#define ALL_BLOCKS(X) " + " ".join("X(%s)" % k for k in pats) + "\n")
