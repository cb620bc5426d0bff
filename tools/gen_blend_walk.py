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


def two_quadrants():
    return walk(ev_half)


# ---------------------------------------------------------------------------------------------------------------------------
# The PIPELINED one-quadrant walk (blend_walk1p_asm): python tools/gen_blend_walk.py pipelined
#
# For launches that cannot fill the machine (a multi-GPU rank's shard: ~4 waves per SIMD) a wave's walk runs at its own latency
# (tools/walk_latency.hip: 77 ns per record, ~7 cycles per instruction of one serial stream), so the stream is software-pipelined
# by one record: alpha of record k+1 (8 instructions, two independent sub-chains, the v_exp_f32 last) is computed BEFORE record k
# is applied (v_cmpx + 5 masked updates).  The transcendental's latency and the quadratic's dependent chain then sit under the
# previous record's update, and nothing waits on an instruction issued just before it.  Same operations on the same values in the
# same per-pixel order as the plain kernel: bit-identical.  Three register sets of 12 (record 10 + p + alpha) rotate: while record k
# is applied (needs its q2 = {L, r, g, b}, p, alpha) record k+1 is evaluated and record k+2's LDS reads are in flight.
# Registers: sets at v54, v66, v78 (g0: mean 2, +2: A B C pthr, +6: L r g b, +10: p, +11: alpha); v90 address, v91 dy, v92 t1,
# v93 dx, v94 u, v95 w.  96 VGPRs: the kernel variant that uses it runs at <= 5 waves per SIMD by construction.
def pipelined():
    lab = itertools.count(40)
    SETS = (54, 66, 78)
    IDX = ("%[ia]", "%[ib]", "%[ic]")
    T, Cr, Cg, Cb = "%[T]", "%[Cr]", "%[Cg]", "%[Cb]"
    ool = []

    def pop_load(s):
        g = SETS[s]
        return [f"s_ff1_i32_b64 {IDX[s]}, %[m]", f"s_bitset0_b64 %[m], {IDX[s]}", f"v_lshl_add_u32 v90, {IDX[s]}, 4, %[base]",
                f"ds_read_b64 v[{g}:{g + 1}], v90", f"ds_read_b128 v[{g + 2}:{g + 5}], v90 offset:%[p1]",
                f"ds_read_b128 v[{g + 6}:{g + 9}], v90 offset:%[p2]"]

    def s1(s):  # alpha of the record in set s: two sub-chains interleaved, full EXEC
        g = SETS[s]
        return [f"v_sub_f32 v91, v{g + 1}, %[fpy]", f"v_sub_f32 v93, v{g}, %[fpx]", f"v_mul_f32 v92, v{g + 4}, v91",
                f"v_mul_f32 v94, v{g + 2}, v93", f"v_fma_f32 v92, v92, v91, v{g + 6}", f"v_fma_f32 v94, v{g + 3}, v91, v94",
                f"v_fma_f32 v{g + 10}, v93, v94, v92", f"v_exp_f32 v{g + 11}, v{g + 10}"]

    def s2(s):  # apply the record in set s (its alpha was computed a block ago)
        g = SETS[s]
        upd = [f"v_cmpx_lt_f32 vcc, {MINA}, v{g + 11}", f"v_mul_f32 v95, {T}, v{g + 11}", f"v_fma_f32 {Cr}, v95, v{g + 7}, {Cr}",
               f"v_fma_f32 {Cg}, v95, v{g + 8}, {Cg}", f"v_fma_f32 {Cb}, v95, v{g + 9}, {Cb}", f"v_fma_f32 {T}, -{T}, v{g + 11}, {T}",
               "s_mov_b64 exec, -1"]
        lg, lj = next(lab), next(lab)
        ool.extend([f"{lg}:", f"v_cmpx_le_f32 vcc, v{g + 10}, v{g + 6}", f"v_min_f32 v{g + 11}, {MAXA}, v{g + 11}"] + upd + [f"s_branch {lj}b"])
        return [f"s_bitcmp1_b64 %[fa], {IDX[s]}", f"s_cbranch_scc0 {lg}f"] + upd + [f"{lj}:"]

    L = ["s_waitcnt lgkmcnt(0)"] + pop_load(0)
    L += ["s_cmp_eq_u64 %[m], 0", "s_cbranch_scc1 30f"]                       # a single survivor
    L += pop_load(1) + ["s_waitcnt lgkmcnt(3)"] + s1(0)
    # block s: record k in set s has its alpha; record k+1 (set s+1) is loaded or loading
    for s in range(3):
        n, nn = (s + 1) % 3, (s + 2) % 3
        L += [f"{10 + s}:", "s_cmp_eq_u64 %[m], 0", f"s_cbranch_scc1 {20 + s}f"]
        L += pop_load(nn) + ["s_waitcnt lgkmcnt(3)"] + s1(n) + s2(s)
        if s == 2:
            L += ["s_branch 10b"]
    for s in range(3):  # drain: no record k+2
        n = (s + 1) % 3
        L += [f"{20 + s}:", "s_waitcnt lgkmcnt(0)"] + s1(n) + s2(s) + s2(n) + ["s_branch 39f"]
    L += ["30:", "s_waitcnt lgkmcnt(0)"] + s1(0) + ["s_nop 0"] + s2(0) + ["s_branch 39f"]  # s_nop: the wait state between v_exp_f32 and its reader
    L += ool + ["39:"]
    return L


if __name__ == "__main__":
    import sys

    lines = pipelined() if len(sys.argv) > 1 and sys.argv[1] == "pipelined" else two_quadrants()
    print("\n".join('        "%s\\n\\t"' % l for l in lines))
