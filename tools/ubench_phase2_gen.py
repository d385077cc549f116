"""Writes tools/ubench_phase2_blocks.inc for tools/ubench_phase2.hip: 1 024-instruction asm statements, eight independent chains.
Operands: %0-%7 x[8], %8-%15 acc[8] (64-bit), %16 y, %17 z.  Patterns: 'a' v_add_u32, 'r' v_alignbit_b32, 'm' v_mad_u64_u32, 'b' v_bitop3_b32,
'x' v_xor_b32, 's' v_lshrrev_b32, '3' v_add3_u32."""
import sys
N = 1024
OPS = {"a": "v_add_u32 %{c}, %{c}, %16", "r": "v_alignbit_b32 %{c}, %{c}, %{c}, 7", "m": "v_mad_u64_u32 %{a}, vcc, %16, %17, %{a}",
       "b": "v_bitop3_b32 %{c}, %{c}, %16, %17 bitop3:0x96", "x": "v_xor_b32 %{c}, %{c}, %16", "s": "v_lshrrev_b32 %{c}, 3, %{c}",
       "3": "v_add3_u32 %{c}, %{c}, %16, %17"}
def block(pattern):
    out = []
    for u in range(N):
        c = u % 8
        out.append(OPS[pattern[u % len(pattern)]].format(c=c, a=c + 8))
    return "\\n\\t".join(out)
pats = {"A": "a", "R": "r", "M": "m", "B": "b", "A7R1": "aaaaaaar", "A3R1": "aaar", "A1R3": "arrr", "A15R1": "a" * 15 + "r", "A31R1": "a" * 31 + "r",
        "A63R1": "a" * 63 + "r", "B3R1": "bbbr", "A3M1": "aaam", "A7M1": "aaaaaaam", "X": "x", "S": "s", "T": "3", "A3T1": "aaa3"}
with open(sys.argv[1], "w") as f:
    for k, p in pats.items():
        f.write('#define BLK_%s "%s"\n' % (k, block(p)))
