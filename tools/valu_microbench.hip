// valu_microbench.hip — measured VALU issue rates on gfx950, per instruction kind and waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o tools/valu_microbench
// Prints cycles per wave64 instruction per SIMD (from wall time and s_memtime clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *cyc)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // v_fma_f32, 8 independent chains x2
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) {  // v_mul_f32
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 2) {  // v_exp_f32
            REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                               "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 3) {  // v_cmp + v_cndmask pairs
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %1, %0, vcc\n v_cmp_gt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %2, %1, vcc\n"
                               "v_cmp_gt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %3, %2, vcc\n v_cmp_gt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %0, %3, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
        } else if (KIND == 4) {  // v_pk_fma_f32 (2 fp32 FMAs per lane)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
            a0 = p0.x + p0.y; a2 = p1.x + p1.y; a4 = p2.x + p2.y; a6 = p3.x + p3.y;
        } else if (KIND == 5) {  // v_sub_f32 with an SGPR operand (the blend's dx = mean - px pattern)
            REP16(asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                               "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 6) {  // dependent chain of v_fma_f32 (latency)
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               : "+v"(a0) : "v"(b), "v"(c));)
        } else if (KIND == 8 || KIND == 9) {  // v_fma_f32 with EXEC = lanes 0..31 only (8) / even lanes only (9): does an empty 32-lane half skip its pass?
            const int l = threadIdx.x & 63;
            if (KIND == 8 ? (l < 32) : ((l & 1) == 0)) {
                REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
            }
        } else if (KIND == 10) {  // v_exp_f32 with EXEC = lanes 0..31 only
            if ((threadIdx.x & 63) < 32) {
                REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                                   "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
            }
        } else if (KIND == 11) {  // v_cmp alone (result to an SGPR pair, never read by a VALU)
            unsigned long long m0, m1, m2, m3;
            REP16(asm volatile("v_cmp_gt_f32 %0, %4, %8\n v_cmp_gt_f32 %1, %5, %8\n v_cmp_gt_f32 %2, %6, %8\n v_cmp_gt_f32 %3, %7, %8\n"
                               "v_cmp_gt_f32 %0, %5, %8\n v_cmp_gt_f32 %1, %6, %8\n v_cmp_gt_f32 %2, %7, %8\n v_cmp_gt_f32 %3, %4, %8\n"
                               : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));)
            a4 += (float)(unsigned)(m0 ^ m1 ^ m2 ^ m3);
        } else if (KIND == 12) {  // v_cndmask alone (mask in a fixed SGPR pair)
            const unsigned long long msk = 0x5555AAAA5555AAAAull ^ (unsigned long long)i;
            REP16(asm volatile("v_cndmask_b32 %0, %1, %0, %4\n v_cndmask_b32 %1, %2, %1, %4\n v_cndmask_b32 %2, %3, %2, %4\n v_cndmask_b32 %3, %0, %3, %4\n"
                               "v_cndmask_b32 %0, %1, %0, %4\n v_cndmask_b32 %1, %2, %1, %4\n v_cndmask_b32 %2, %3, %2, %4\n v_cndmask_b32 %3, %0, %3, %4\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(msk));)
        } else if (KIND == 13) {  // v_cmpx + 3 v_fma under the new EXEC + restore (the masked-update form of the blend)
            REP16(asm volatile("v_cmpx_gt_f32 %0, %4\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n s_mov_b64 exec, -1\n"
                               "v_cmpx_gt_f32 %1, %4\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n s_mov_b64 exec, -1\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
        } else if (KIND == 14) {  // v_readlane_b32 (VGPR lane -> SGPR)
            unsigned s0, s1, s2, s3;
            REP16(asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 17\n v_readlane_b32 %2, %6, 40\n v_readlane_b32 %3, %7, 63\n"
                               "v_readlane_b32 %0, %5, 5\n v_readlane_b32 %1, %6, 19\n v_readlane_b32 %2, %7, 42\n v_readlane_b32 %3, %4, 61\n"
                               : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
            a4 += (float)(s0 ^ s1 ^ s2 ^ s3);
        } else if (KIND == 7) {  // ds_read_b128 broadcast (all lanes same address)
            __shared__ float4 buf[256];
            if (i == 0) buf[threadIdx.x] = make_float4(a0, a1, a2, a3);
            const float4 *p = &buf[(i * 7) & 255];
            float4 v;
            REP16(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");)
            a0 += v.x;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char *name, int blocks_per_cu, int instr_per_iter)
{
    const int cus = 256, iters = 2000;
    float *out;
    unsigned long long *cyc, h = 0;
    hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<cus * blocks_per_cu, 256>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<cus * blocks_per_cu, 256>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double instr_per_wave = (double)iters * instr_per_iter;
    const double waves_per_simd = blocks_per_cu;  // 4 waves per block, 4 SIMDs per CU
    // s_memtime ticks at 100 MHz on gfx9 (constant clock); report both wall-derived and tick-derived numbers
    const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-28s waves/SIMD=%d  %8.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz, %.2f @2.0GHz)  memtime ticks=%llu\n",
           name, blocks_per_cu, ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4, ns_per_instr_per_simd * 2.0, h);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {2, 8}) {
        run<0>("v_fma_f32 x8 indep", w, 128); run<8>("v_fma_f32 EXEC=lanes 0-31", w, 128); run<9>("v_fma_f32 EXEC=even lanes", w, 128);
        run<2>("v_exp_f32", w, 128); run<10>("v_exp_f32 EXEC=lanes 0-31", w, 128);
        run<3>("v_cmp+v_cndmask pairs", w, 128); run<11>("v_cmp alone (to SGPR)", w, 128); run<12>("v_cndmask alone", w, 128);
        run<13>("v_cmpx + 3 fma + s_mov exec", w, 128 + 32); run<14>("v_readlane_b32", w, 128);
    }
    if (getenv("GSR_MICROBENCH_FULL"))
    for (int w : {1, 2, 4, 8}) {
        if (w == 1) { run<0>("v_fma_f32 x8 indep", 1, 128); run<1>("v_mul_f32", 1, 128); run<2>("v_exp_f32", 1, 128); run<3>("v_cmp+v_cndmask", 1, 128); run<4>("v_pk_fma_f32", 1, 128); run<5>("v_sub_f32", 1, 128); run<6>("v_fma_f32 dependent", 1, 128); run<7>("ds_read_b128 bcast+wait", 1, 16); }
        if (w == 2) { run<0>("v_fma_f32 x8 indep", 2, 128); run<1>("v_mul_f32", 2, 128); run<2>("v_exp_f32", 2, 128); run<3>("v_cmp+v_cndmask", 2, 128); run<4>("v_pk_fma_f32", 2, 128); run<6>("v_fma_f32 dependent", 2, 128); run<7>("ds_read_b128 bcast+wait", 2, 16); }
        if (w == 4) { run<0>("v_fma_f32 x8 indep", 4, 128); run<1>("v_mul_f32", 4, 128); run<2>("v_exp_f32", 4, 128); run<3>("v_cmp+v_cndmask", 4, 128); run<4>("v_pk_fma_f32", 4, 128); run<6>("v_fma_f32 dependent", 4, 128); run<7>("ds_read_b128 bcast+wait", 4, 16); }
        if (w == 8) { run<0>("v_fma_f32 x8 indep", 8, 128); run<1>("v_mul_f32", 8, 128); run<2>("v_exp_f32", 8, 128); run<3>("v_cmp+v_cndmask", 8, 128); run<4>("v_pk_fma_f32", 8, 128); run<6>("v_fma_f32 dependent", 8, 128); run<7>("ds_read_b128 bcast+wait", 8, 16); }
    }
    return 0;
}
