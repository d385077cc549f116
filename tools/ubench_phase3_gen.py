"""Writes tools/ubench_phase3_blocks.inc for tools/ubench_phase3.hip: asm statements of 1 024 VALU instructions over eight independent chains,
with wave-priority changes / yields between them.  Pattern letters: a v_add_u32, r v_alignbit_b32, m v_mad_u64_u32, b v_bitop3_b32, 3 v_add3_u32 (VALU);
P s_setprio 1, Q s_setprio 2, p s_setprio 0, n s_nop 0 (not counted)."""
import sys
N = 1024
OPS = {"a": "v_add_u32 %{c}, %{c}, %16", "r": "v_alignbit_b32 %{c}, %{c}, %{c}, 7", "m": "v_mad_u64_u32 %{a}, vcc, %16, %17, %{a}",
       "b": "v_bitop3_b32 %{c}, %{c}, %16, %17 bitop3:0x96", "3": "v_add3_u32 %{c}, %{c}, %16, %17"}
CTL = {"P": "s_setprio 1", "Q": "s_setprio 2", "p": "s_setprio 0", "n": "s_nop 0"}
def block(pattern):
    out, u, i = [], 0, 0
    while u < N:
        ch = pattern[i % len(pattern)]
        i += 1
        if ch in CTL:
            out.append(CTL[ch])
            continue
        c = u % 8
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
with open(sys.argv[1], "w") as f:
    for k, p in pats.items():
        f.write('#define BLK_%s "%s"\n' % (k, block(p)))
    f.write("#define ALL_BLOCKS(X) " + " ".join("X(%s)" % k for k in pats) + "\n")
