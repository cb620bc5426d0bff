// scratch test: blend_one (C) vs blend_one_x (EXEC-masked asm) on random inputs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#ifndef NOPS
#define NOPS ""
#endif
#define GSR_MAX_ALPHA 0.99f
#define GSR_MIN_ALPHA (1.0f / 255.0f)
__device__ __forceinline__ void blend_one(const float2 g, const float4 c, const float4 o, float fpx, float fpy, float &T, float &Cr, float &Cg, float &Cb)
{
    const float dx = g.x - fpx, dy = g.y - fpy;
    const float p = fmaf(dx, fmaf(c.y, dy, c.x * dx), fmaf(c.z * dy, dy, o.x));
    float alpha = fminf(__builtin_amdgcn_exp2f(p), GSR_MAX_ALPHA);
    const bool valid = (alpha > GSR_MIN_ALPHA) & (p <= o.x);
    alpha = valid ? alpha : 0.0f;
    const float w = alpha * T;
    Cr = fmaf(w, o.y, Cr); Cg = fmaf(w, o.z, Cg); Cb = fmaf(w, o.w, Cb);
    T = T - w;
}
__device__ __forceinline__ void blend_one_x(const float2 g, const float4 c, const float4 o, float fpx, float fpy, float &T, float &Cr, float &Cg, float &Cb)
{
    float dx, dy, t0, t1;
    asm volatile(
        "v_sub_f32 %4, %8, %16\n\t" "v_sub_f32 %5, %9, %17\n\t" "v_mul_f32 %6, %10, %4\n\t" "v_fma_f32 %6, %11, %5, %6\n\t"
        "v_mul_f32 %7, %12, %5\n\t" "v_fma_f32 %7, %7, %5, %13\n\t" "v_fma_f32 %6, %4, %6, %7\n\t" "v_exp_f32 %7, %6\n\t" "v_cmpx_le_f32 vcc, %6, %13\n\t"
        "v_min_f32 %7, 0x3f7d70a4, %7\n\t" "v_cmpx_lt_f32 vcc, 0x3b808081, %7\n\t"
        "v_mul_f32 %7, %7, %0\n\t" "v_fma_f32 %1, %7, %14, %1\n\t" "v_fma_f32 %2, %7, %15, %2\n\t" "v_fma_f32 %3, %7, %18, %3\n\t"
        "v_sub_f32 %0, %0, %7\n\t" "s_mov_b64 exec, -1"
        : "+v"(T), "+v"(Cr), "+v"(Cg), "+v"(Cb), "=&v"(dx), "=&v"(dy), "=&v"(t0), "=&v"(t1)
        : "v"(g.x), "v"(g.y), "v"(c.x), "v"(c.y), "v"(c.z), "v"(o.x), "v"(o.y), "v"(o.z), "v"(fpx), "v"(fpy), "v"(o.w), "v"(c.w)
        : "vcc");
}
// V1: p and alpha in C, EXEC-masked update in asm.  V2: p and alpha in asm, select + update in C.
__device__ __forceinline__ void blend_v1(const float2 g, const float4 c, const float4 o, float fpx, float fpy, float &T, float &Cr, float &Cg, float &Cb)
{
    const float dx = g.x - fpx, dy = g.y - fpy;
    const float p = fmaf(dx, fmaf(c.y, dy, c.x * dx), fmaf(c.z * dy, dy, o.x));
    float alpha = fminf(__builtin_amdgcn_exp2f(p), GSR_MAX_ALPHA);
    float w;
    asm volatile("v_cmpx_lt_f32 vcc, 0x3b808081, %5\n\t" "v_cmpx_le_f32 vcc, %6, %7\n\t"
        "v_mul_f32 %4, %5, %0\n\t" "v_fma_f32 %1, %4, %8, %1\n\t" "v_fma_f32 %2, %4, %9, %2\n\t" "v_fma_f32 %3, %4, %10, %3\n\t"
        "v_sub_f32 %0, %0, %4\n\t" "s_mov_b64 exec, -1"
        : "+v"(T), "+v"(Cr), "+v"(Cg), "+v"(Cb), "=&v"(w) : "v"(alpha), "v"(p), "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w) : "vcc");
}
__device__ __forceinline__ void blend_v2(const float2 g, const float4 c, const float4 o, float fpx, float fpy, float &T, float &Cr, float &Cg, float &Cb)
{
    float dx, dy, p, alpha;
    asm volatile("v_sub_f32 %0, %4, %10\n\t" "v_sub_f32 %1, %5, %11\n\t" "v_mul_f32 %2, %6, %0\n\t" "v_fma_f32 %2, %7, %1, %2\n\t"
        "v_mul_f32 %3, %8, %1\n\t" "v_fma_f32 %3, %3, %1, %9\n\t" "v_fma_f32 %2, %0, %2, %3\n\t" "v_exp_f32 %3, %2\n\t"
        "v_min_f32 %3, 0x3f7d70a4, %3"
        : "=&v"(dx), "=&v"(dy), "=&v"(p), "=&v"(alpha) : "v"(g.x), "v"(g.y), "v"(c.x), "v"(c.y), "v"(c.z), "v"(o.x), "v"(fpx), "v"(fpy));
    const bool valid = (alpha > GSR_MIN_ALPHA) & (p <= o.x);
    alpha = valid ? alpha : 0.0f;
    const float w = alpha * T;
    Cr = fmaf(w, o.y, Cr); Cg = fmaf(w, o.z, Cg); Cb = fmaf(w, o.w, Cb);
    T = T - w;
}
__global__ void kv(const float *rec, int n, float *out)
{
    const float fpx = (float)(threadIdx.x & 7), fpy = (float)(threadIdx.x >> 3);
    float T0 = 1, a0 = 0, b0 = 0, c0 = 0, T1 = 1, a1 = 0, b1 = 0, c1 = 0, T2 = 1, a2 = 0, b2 = 0, c2 = 0;
    for (int i = 0; i < n; ++i) {
        const float *r = rec + 12 * i;
        const float2 g = make_float2(r[0], r[1]);
        const float4 c = make_float4(r[4], r[5], r[6], r[7]), o = make_float4(r[8], r[9], r[10], r[11]);
        blend_one(g, c, o, fpx, fpy, T0, a0, b0, c0);
        blend_v1(g, c, o, fpx, fpy, T1, a1, b1, c1);
        blend_v2(g, c, o, fpx, fpy, T2, a2, b2, c2);
    }
    out[threadIdx.x * 3] = T0; out[threadIdx.x * 3 + 1] = T1; out[threadIdx.x * 3 + 2] = T2;
}
__global__ void dbg(const float *rec, float *out, unsigned long long *masks)
{
    const float fpx = (float)(threadIdx.x & 7), fpy = (float)(threadIdx.x >> 3);
    const float *r = rec;
    const float2 g = make_float2(r[0], r[1]);
    const float4 c = make_float4(r[4], r[5], r[6], r[7]), o = make_float4(r[8], r[9], r[10], r[11]);
    const float dx = g.x - fpx, dy = g.y - fpy;
    const float p = fmaf(dx, fmaf(c.y, dy, c.x * dx), fmaf(c.z * dy, dy, o.x));
    float alpha = fminf(__builtin_amdgcn_exp2f(p), GSR_MAX_ALPHA);
    unsigned long long e1, e2, v1, v2;
    asm volatile("v_cmpx_lt_f32 vcc, 0x3b808081, %4\n\t s_mov_b64 %0, exec\n\t s_mov_b64 %2, vcc\n\t v_cmpx_le_f32 vcc, %5, %6\n\t s_mov_b64 %1, exec\n\t s_mov_b64 %3, vcc\n\t s_mov_b64 exec, -1"
                 : "=s"(e1), "=s"(e2), "=s"(v1), "=s"(v2) : "v"(alpha), "v"(p), "v"(o.x) : "vcc");
    out[threadIdx.x * 2] = p; out[threadIdx.x * 2 + 1] = alpha;
    if (threadIdx.x == 0) { masks[0] = e1; masks[1] = e2; masks[2] = v1; masks[3] = v2; masks[4] = __ballot(alpha > GSR_MIN_ALPHA); masks[5] = __ballot(p <= o.x); }
}
__global__ void k(const float *rec, int n, float *outA, float *outB)
{
    const float fpx = (float)(threadIdx.x & 7), fpy = (float)(threadIdx.x >> 3);
    float T = 1, Cr = 0, Cg = 0, Cb = 0, T2 = 1, Cr2 = 0, Cg2 = 0, Cb2 = 0;
    for (int i = 0; i < n; ++i) {
        const float *r = rec + 12 * i;
        const float2 g = make_float2(r[0], r[1]);
        const float4 c = make_float4(r[4], r[5], r[6], r[7]), o = make_float4(r[8], r[9], r[10], r[11]);
        blend_one(g, c, o, fpx, fpy, T, Cr, Cg, Cb);
        blend_one_x(g, c, o, fpx, fpy, T2, Cr2, Cg2, Cb2);
        if (i < 4) { outA[(i * 64 + threadIdx.x) * 4 + 0] = T; outA[(i * 64 + threadIdx.x) * 4 + 1] = Cr; outB[(i * 64 + threadIdx.x) * 4 + 0] = T2; outB[(i * 64 + threadIdx.x) * 4 + 1] = Cr2; }
    }
    float *a = outA + 4 * 64 * 4 + threadIdx.x * 4, *b = outB + 4 * 64 * 4 + threadIdx.x * 4;
    a[0] = T; a[1] = Cr; a[2] = Cg; a[3] = Cb; b[0] = T2; b[1] = Cr2; b[2] = Cg2; b[3] = Cb2;
}
int main()
{
    const int n = 40;
    float h[12 * n];
    srand(1);
    for (int i = 0; i < n; ++i) {
        float *r = h + 12 * i;
        r[0] = 8.0f * rand() / RAND_MAX; r[1] = 8.0f * rand() / RAND_MAX; r[2] = r[3] = 0;
        r[4] = -0.05f - 0.2f * rand() / RAND_MAX; r[5] = 0.05f * rand() / RAND_MAX; r[6] = -0.05f - 0.2f * rand() / RAND_MAX; r[7] = 0;
        r[8] = log2f(0.1f + 0.85f * rand() / RAND_MAX); r[9] = 1.0f * rand() / RAND_MAX; r[10] = 1.0f * rand() / RAND_MAX; r[11] = 1.0f * rand() / RAND_MAX;
    }
    float *d, *a, *b;
    const int outn = (4 * 64 * 4 + 64 * 4);
    hipMalloc(&d, sizeof h); hipMalloc(&a, outn * 4); hipMalloc(&b, outn * 4);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    {
        float *o; unsigned long long *m; hipMalloc(&o, 64 * 8); hipMalloc(&m, 48);
        dbg<<<1, 64>>>(d, o, m);
        float ho[128]; unsigned long long hm[6];
        hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost); hipMemcpy(hm, m, 48, hipMemcpyDeviceToHost);
        printf("exec after cmpx1 %016llx  vcc %016llx   ballot(alpha > MIN) %016llx\n", hm[0], hm[2], hm[4]);
        printf("exec after cmpx2 %016llx  vcc %016llx   ballot(p <= L)      %016llx\n", hm[1], hm[3], hm[5]);
        printf("L = %f; lane 0: p %f alpha %f; lane 42: p %f alpha %f; lane 21: p %f alpha %f\n", h[8], ho[0], ho[1], ho[84], ho[85], ho[42], ho[43]);
    }
    {
        float *o; hipMalloc(&o, 64 * 12);
        kv<<<1, 64>>>(d, n, o);
        float ho[192]; hipMemcpy(ho, o, 768, hipMemcpyDeviceToHost);
        for (int v = 1; v <= 2; ++v) { for (int l = 0; l < 64; ++l) printf("%c", fabsf(ho[3 * l] - ho[3 * l + v]) > 1e-6f ? 'X' : '.'); printf("  <- V%d vs C\n", v); }
    }
    k<<<1, 64>>>(d, n, a, b);
    float ha[outn], hb[outn];
    hipMemcpy(ha, a, outn * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, outn * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; l += 21) printf("entry %d lane %2d: C version T %.6f Cr %.6f | asm T %.6f Cr %.6f\n", i, l, ha[(i * 64 + l) * 4], ha[(i * 64 + l) * 4 + 1], hb[(i * 64 + l) * 4], hb[(i * 64 + l) * 4 + 1]);
    double md = 0; for (int i = 4 * 64 * 4; i < outn; ++i) md = fmax(md, fabs((double)ha[i] - hb[i]));
    printf("final max abs diff over lanes: %g\n", md);
    for (int l = 0; l < 64; ++l) printf("%c", fabsf(ha[4 * 64 * 4 + l * 4] - hb[4 * 64 * 4 + l * 4]) > 1e-6f ? 'X' : '.');
    printf("  <- lanes whose final T differs\n");
    for (int l = 0; l < 64; ++l) printf("%c", hb[4 * 64 * 4 + l * 4] == 1.0f ? '1' : '.');
    printf("  <- lanes whose asm T is still exactly 1\n");
    return 0;
}
