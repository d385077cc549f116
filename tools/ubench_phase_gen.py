"""Writes tools/ubench_phase_blocks.inc: one asm string per (run length, class P, class Q) of tools/ubench_phase.hip — 2 048 instructions
as runs of R of class P then R of class Q over eight independent chains, ONE asm statement each (hipcc puts an s_nop between
single-instruction asm statements, which is itself an issue-slot yield).  Operands: %0-%7 x[8], %8-%15 acc[8] (64-bit), %16 y, %17 z."""
import sys
N = 2048
OPS = {0: "v_add_u32 %{c}, %{c}, %16", 1: "v_alignbit_b32 %{c}, %{c}, %{c}, 7", 2: "v_mad_u64_u32 %{a}, vcc, %16, %17, %{a}",
       3: "v_bitop3_b32 %{c}, %{c}, %16, %17 bitop3:0x96"}
def block(r, p, q, yield_every=0):
    out = []
    for u in range(N):
        cls = q if (u // r) & 1 else p
        c = u % 8
        out.append(OPS[cls].format(c=c, a=c + 8))
        if yield_every and (u + 1) % yield_every == 0:
            out.append("s_nop 0")
    return "\\n\\t".join(out)
cases = [(8, 0, 0), (8, 1, 1), (8, 2, 2), (1, 0, 1), (4, 0, 1), (16, 0, 1), (64, 0, 1), (256, 0, 1), (1024, 0, 1), (16, 3, 1), (16, 0, 2), (256, 0, 2), (16, 1, 2)]
with open(sys.argv[1], "w") as f:
    for r, p, q in cases:
        for y in (0, 3):
            f.write('#define BLK_%d_%d_%d_Y%d "%s"\n' % (r, p, q, y, block(r, p, q, y)))
            if p != q:
                f.write('#define BLK_%d_%d_%d_Y%d "%s"\n' % (r, q, p, y, block(r, q, p, y)))
