#!/usr/bin/env python3
"""Prints the asm text of blend_walk_asm (csrc/blend.hip): the survivor walk of one chunk with the records rolling through
two register sets and a guarded / unguarded evaluation chosen per survivor from the mask %[fast].  The kernel source carries
the output verbatim; this script only documents how it was produced.
Registers: v40/v41 LDS addresses, then temporaries; set A = v42..v51, set B = v52..v61; v62/v63 temporaries."""
import itertools

label = itertools.count(10)


def quad(g0, c0, o0):
    return [f"v_sub_f32 v40, v{g0}, %[fpx]", f"v_sub_f32 v41, v{g0 + 1}, %[fpy]",
            f"v_mul_f32 v62, v{c0}, v40", f"v_fma_f32 v62, v{c0 + 1}, v41, v62",
            f"v_mul_f32 v63, v{c0 + 2}, v41", f"v_fma_f32 v63, v63, v41, v{o0}",
            "v_fma_f32 v62, v40, v62, v63", "v_exp_f32 v63, v62"]


def update(o0):
    return ["v_mul_f32 v40, %[T], v63", f"v_fma_f32 %[Cr], v40, v{o0 + 1}, %[Cr]", f"v_fma_f32 %[Cg], v40, v{o0 + 2}, %[Cg]",
            f"v_fma_f32 %[Cb], v40, v{o0 + 3}, %[Cb]", "v_fma_f32 %[T], -%[T], v63, %[T]", "s_mov_b64 exec, -1"]


def ev(idx, g0, c0, o0):
    lf, lj = next(label), next(label)
    full = [f"v_cmpx_le_f32 vcc, v62, v{o0}", "v_min_f32 v63, 0x3f7d70a4, v63", "v_cmpx_lt_f32 vcc, 0x3b808081, v63"] + update(o0)
    fast = ["s_nop 0", "v_cmpx_lt_f32 vcc, 0x3b808081, v63"] + update(o0)
    return quad(g0, c0, o0) + [f"s_bitcmp1_b64 %[fast], {idx}", f"s_cbranch_scc1 {lf}f"] + full + [f"s_branch {lj}f", f"{lf}:"] + fast + [f"{lj}:"]


def load(idx, addr, g0, c0, o0):
    return [f"s_ff1_i32_b64 {idx}, %[m]", f"s_bitset0_b64 %[m], {idx}", f"v_lshl_add_u32 v{addr}, {idx}, 4, %[base]",
            f"ds_read_b64 v[{g0}:{g0 + 1}], v{addr}", f"ds_read_b128 v[{c0}:{c0 + 3}], v{addr} offset:4096",
            f"ds_read_b128 v[{o0}:{o0 + 3}], v{addr} offset:8192"]


A, B = (42, 44, 48), (52, 54, 58)
IA, IB = "%[ia]", "%[ib]"
lines = ["s_waitcnt lgkmcnt(0)"] + load(IA, 40, *A)
lines += ["1:", "s_cmp_eq_u64 %[m], 0", "s_cbranch_scc1 3f"] + load(IB, 41, *B) + ["s_waitcnt lgkmcnt(3)"] + ev(IA, *A)
lines += ["s_cmp_eq_u64 %[m], 0", "s_cbranch_scc1 4f"] + load(IA, 40, *A) + ["s_waitcnt lgkmcnt(3)"] + ev(IB, *B) + ["s_branch 1b"]
lines += ["3:", "s_waitcnt lgkmcnt(0)"] + ev(IA, *A) + ["s_branch 5f"]
lines += ["4:", "s_waitcnt lgkmcnt(0)"] + ev(IB, *B) + ["5:"]
print("\n".join('        "%s\\n\\t"' % l for l in lines))
