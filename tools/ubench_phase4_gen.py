"""Writes tools/ubench_phase4_blocks.inc for tools/ubench_phase4.hip: which place of an issue slot does an opcode take?  One asm statement of 1 024
instructions per opcode, eight independent chains.  Operands: %0-%7 x[8], %8-%15 acc[8] (64-bit), %16 y, %17 z."""
import sys
N = 1024
OPS = {
    "add": "v_add_u32 %{c}, %{c}, %16", "alignbit": "v_alignbit_b32 %{c}, %{c}, %{c}, 7", "mad64": "v_mad_u64_u32 %{a}, vcc, %16, %17, %{a}",
    "bitop3": "v_bitop3_b32 %{c}, %{c}, %16, %17 bitop3:0x96", "add3": "v_add3_u32 %{c}, %{c}, %16, %17",
    "bfe_i32": "v_bfe_i32 %{c}, %{c}, 3, 29", "bfe_u32": "v_bfe_u32 %{c}, %{c}, 3, 29", "lshl_add_u64": "v_lshl_add_u64 %{a}, %{a}, 1, %{a}",
    "lshrrev_b64": "v_lshrrev_b64 %{a}, 29, %{a}", "cndmask": "v_cndmask_b32 %{c}, %{c}, %16, vcc", "cmp": "v_cmp_gt_u32 vcc, %{c}, %16",
    "add_co": "v_add_co_u32 %{c}, vcc, %{c}, %16", "addc_co": "v_addc_co_u32 %{c}, vcc, %{c}, %16, vcc", "and_or": "v_and_or_b32 %{c}, %{c}, %16, %17",
    "perm": "v_perm_b32 %{c}, %{c}, %16, %17", "mul_lo": "v_mul_lo_u32 %{c}, %{c}, %16", "mul_hi": "v_mul_hi_u32 %{c}, %{c}, %16",
    "mad_u32_u24": "v_mad_u32_u24 %{c}, %{c}, %16, %17", "mul_u32_u24": "v_mul_u32_u24 %{c}, %{c}, %16", "mov": "v_mov_b32 %{c}, %16",
    "lshl_add_u32": "v_lshl_add_u32 %{c}, %{c}, 3, %16", "lshl_or": "v_lshl_or_b32 %{c}, %{c}, 3, %16", "xor": "v_xor_b32 %{c}, %{c}, %16",
    "add_lit": "v_add_u32 %{c}, 0x5a827999, %{c}", "sub": "v_sub_u32 %{c}, %{c}, %16", "lshlrev": "v_lshlrev_b32 %{c}, 3, %{c}",
    "and_e64": "v_and_b32_e64 %{c}, %{c}, %16", "fma_f32": "v_fma_f32 %{c}, %{c}, %16, %17", "mul_f32": "v_mul_f32 %{c}, %{c}, %16",
}
with open(sys.argv[1], "w") as f:
    for k, op in OPS.items():
        f.write('#define BLK_%s "%s"\n' % (k, "\\n\\t".join(op.format(c=u % 8, a=8 + u % 8) for u in range(N))))
    f.write("#define ALL_OPS(X) " + " ".join("X(%s)" % k for k in OPS if k not in ("add", "alignbit")) + "\n")
