#!/usr/bin/env python3
"""Prints the asm text of blend_walk_asm (csrc/blend.hip): the survivor walk of one chunk with the records rolling through
two register sets.  The kernel source carries the output verbatim; this script only documents how it was produced.
Registers: v40/v41 LDS addresses, then temporaries; set A = v42..v51, set B = v52..v61; v62/v63 temporaries."""


def ev(g0, c0, o0):
    return [f"v_sub_f32 v40, v{g0}, %[fpx]", f"v_sub_f32 v41, v{g0 + 1}, %[fpy]",
            f"v_mul_f32 v62, v{c0}, v40", f"v_fma_f32 v62, v{c0 + 1}, v41, v62",
            f"v_mul_f32 v63, v{c0 + 2}, v41", f"v_fma_f32 v63, v63, v41, v{o0}",
            "v_fma_f32 v62, v40, v62, v63", "v_exp_f32 v63, v62",
            f"v_cmpx_le_f32 vcc, v62, v{o0}", "v_min_f32 v63, 0x3f7d70a4, v63",
            "v_cmpx_lt_f32 vcc, 0x3b808081, v63", "v_mul_f32 v40, %[T], v63",
            f"v_fma_f32 %[Cr], v40, v{o0 + 1}, %[Cr]", f"v_fma_f32 %[Cg], v40, v{o0 + 2}, %[Cg]",
            f"v_fma_f32 %[Cb], v40, v{o0 + 3}, %[Cb]", "v_fma_f32 %[T], -%[T], v63, %[T]", "s_mov_b64 exec, -1"]


def load(addr, g0, c0, o0):
    return ["s_ff1_i32_b64 %[idx], %[m]", "s_bitset0_b64 %[m], %[idx]", f"v_lshl_add_u32 v{addr}, %[idx], 4, %[base]",
            f"ds_read_b64 v[{g0}:{g0 + 1}], v{addr}", f"ds_read_b128 v[{c0}:{c0 + 3}], v{addr} offset:4096",
            f"ds_read_b128 v[{o0}:{o0 + 3}], v{addr} offset:8192"]


A, B = (42, 44, 48), (52, 54, 58)
lines = ["s_waitcnt lgkmcnt(0)"] + load(40, *A)
lines += ["1:", "s_cmp_eq_u64 %[m], 0", "s_cbranch_scc1 3f"] + load(41, *B) + ["s_waitcnt lgkmcnt(3)"] + ev(*A)
lines += ["s_cmp_eq_u64 %[m], 0", "s_cbranch_scc1 4f"] + load(40, *A) + ["s_waitcnt lgkmcnt(3)"] + ev(*B) + ["s_branch 1b"]
lines += ["3:", "s_waitcnt lgkmcnt(0)"] + ev(*A) + ["s_branch 5f"]
lines += ["4:", "s_waitcnt lgkmcnt(0)"] + ev(*B) + ["5:"]
print("\n".join('        "%s\\n\\t"' % l for l in lines))
