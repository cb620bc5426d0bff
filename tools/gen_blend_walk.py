#!/usr/bin/env python3
"""Prints the asm text of the blend kernel's survivor walk (blend_walk2_asm in csrc/blend.hip).  The kernel source carries the
output verbatim; this script documents how it was produced.

Two quadrants (a 16x8 half-tile) per wave, a lane = two pixels 8 columns apart.  Records roll through two register sets; one
set of LDS reads per record serves both quadrants, and so do dy, C*dy and t1 = fma(C dy, dy, L) of the quadratic; which
quadrants a record is evaluated on comes from the masks %[ma] / %[mb], guarded or not from %[fa] / %[fb].
Registers: v40/v41 LDS addresses, then temporaries; set A = v42..v51, set B = v52..v61; v39/v62/v63 temporaries."""
import itertools

label = itertools.count(10)
MAXA, MINA = "0x3f7d70a4", "0x3b808081"  # 0.99f, 1/255


def update(o0, T, Cr, Cg, Cb):
    return [f"v_mul_f32 v40, {T}, v63", f"v_fma_f32 {Cr}, v40, v{o0 + 1}, {Cr}", f"v_fma_f32 {Cg}, v40, v{o0 + 2}, {Cg}",
            f"v_fma_f32 {Cb}, v40, v{o0 + 3}, {Cb}", f"v_fma_f32 {T}, -{T}, v63, {T}", "s_mov_b64 exec, -1"]


OUT_OF_LINE = []  # guarded evaluations: the rarer case (~20 %) leaves the straight-line path and jumps back


def guarded_or_not(fastmask, idx, o0, st):
    lg, lj = next(label), next(label)
    full = [f"v_cmpx_le_f32 vcc, v62, v{o0}", f"v_min_f32 v63, {MAXA}, v63", f"v_cmpx_lt_f32 vcc, {MINA}, v63"] + update(o0, *st)
    # (the s_bitcmp1 + s_cbranch below sit between v_exp_f32 and its first consumer: the one wait state a transcendental's
    # result needs on gfx940+; the guarded path starts with a v_cmpx that does not read it)
    fast = [f"v_cmpx_lt_f32 vcc, {MINA}, v63"] + update(o0, *st)
    OUT_OF_LINE.extend([f"{lg}:"] + full + [f"s_branch {lj}b"])
    return [f"s_bitcmp1_b64 {fastmask}, {idx}", f"s_cbranch_scc0 {lg}f"] + fast + [f"{lj}:"]


def load(idx, addr, g0, c0, o0):
    # %[p1] / %[p2]: byte offsets of the record's second and third LDS plane (immediates: BATCH * 16, BATCH * 32)
    return [f"s_ff1_i32_b64 {idx}, %[m]", f"s_bitset0_b64 %[m], {idx}", f"v_lshl_add_u32 v{addr}, {idx}, 4, %[base]",
            f"ds_read_b64 v[{g0}:{g0 + 1}], v{addr}", f"ds_read_b128 v[{c0}:{c0 + 3}], v{addr} offset:%[p1]",
            f"ds_read_b128 v[{o0}:{o0 + 3}], v{addr} offset:%[p2]"]


def ev_half(idx, g0, c0, o0):
    # shared by both quadrants: v41 = dy, v39 = t1 = fma(C dy, dy, L)
    out = [f"v_sub_f32 v41, v{g0 + 1}, %[fpy]", f"v_mul_f32 v39, v{c0 + 2}, v41", f"v_fma_f32 v39, v39, v41, v{o0}"]
    for q, (hit, fast, fpx) in enumerate((("%[ma]", "%[fa]", "%[fpxa]"), ("%[mb]", "%[fb]", "%[fpxb]"))):
        st = tuple(f"%[{n}{'ab'[q]}]" for n in ("T", "Cr", "Cg", "Cb"))
        skip = next(label)
        out += [f"s_bitcmp1_b64 {hit}, {idx}", f"s_cbranch_scc0 {skip}f",
                f"v_sub_f32 v40, v{g0}, {fpx}", f"v_mul_f32 v62, v{c0}, v40", f"v_fma_f32 v62, v{c0 + 1}, v41, v62",
                "v_fma_f32 v62, v40, v62, v39", "v_exp_f32 v63, v62"]
        out += guarded_or_not(fast, idx, o0, st) + [f"{skip}:"]
    return out


def walk(ev):
    A, B = (42, 44, 48), (52, 54, 58)
    IA, IB = "%[ia]", "%[ib]"
    lines = ["s_waitcnt lgkmcnt(0)"] + load(IA, 40, *A)
    lines += ["1:", "s_cmp_eq_u64 %[m], 0", "s_cbranch_scc1 3f"] + load(IB, 41, *B) + ["s_waitcnt lgkmcnt(3)"] + ev(IA, *A)
    lines += ["s_cmp_eq_u64 %[m], 0", "s_cbranch_scc1 4f"] + load(IA, 40, *A) + ["s_waitcnt lgkmcnt(3)"] + ev(IB, *B) + ["s_branch 1b"]
    lines += ["3:", "s_waitcnt lgkmcnt(0)"] + ev(IA, *A) + ["s_branch 5f"]
    lines += ["4:", "s_waitcnt lgkmcnt(0)"] + ev(IB, *B) + ["s_branch 5f"] + OUT_OF_LINE + ["5:"]
    return lines


lines = walk(ev_half)
print("\n".join('        "%s\\n\\t"' % l for l in lines))
