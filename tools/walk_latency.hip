// walk_latency.hip — how long does ONE wave take per record in the blend's survivor walk (csrc/blend.hip, blend_walk2_asm)?
// The kernel's throughput is waves-per-SIMD / per-wave latency until the VALU saturates, so this is the number that
// matters.  W waves per SIMD on the whole chip (W = 1: every wave alone on its SIMD), 64 synthetic records in LDS,
// every chunk walks all 64 with the given hit / fast masks; time from the 100 MHz wall clock.
// Build: python tools/gen_blend_walk.py > /tmp/walk_asm.inc && hipcc --offload-arch=gfx950 -O3 -I/tmp tools/walk_latency.hip -o tools/walk_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int PLANE = 2048;

__device__ __forceinline__ void walk(unsigned long long m, unsigned long long ma, unsigned long long mb, unsigned long long fa, unsigned long long fb,
                                     unsigned lds_chunk, float fpxa, float fpxb, float fpy, float &Ta, float &Cra, float &Cga, float &Cba, float &Tb,
                                     float &Crb, float &Cgb, float &Cbb)
{
    int ia, ib;
    asm volatile(
#include "walk_asm.inc"
        : [Ta] "+v"(Ta), [Cra] "+v"(Cra), [Cga] "+v"(Cga), [Cba] "+v"(Cba), [Tb] "+v"(Tb), [Crb] "+v"(Crb), [Cgb] "+v"(Cgb),
          [Cbb] "+v"(Cbb), [m] "+s"(m), [ia] "=&s"(ia), [ib] "=&s"(ib)
        : [base] "v"(lds_chunk), [fpxa] "v"(fpxa), [fpxb] "v"(fpxb), [fpy] "v"(fpy), [ma] "s"(ma), [mb] "s"(mb), [fa] "s"(fa), [fb] "s"(fb),
          [p1] "i"(PLANE), [p2] "i"(2 * PLANE)
        : "vcc", "scc", "memory", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
          "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

// the pipelined one-quadrant walk (blend_walk1p_asm): python tools/gen_blend_walk.py pipelined > /tmp/walk1p_asm.inc
__device__ __forceinline__ void walk1p(unsigned long long m, unsigned long long fa, unsigned lds_chunk, float fpx, float fpy, float &T, float &Cr,
                                       float &Cg, float &Cb)
{
    int ia, ib, ic;
    asm volatile(
#include "walk1p_asm.inc"
        : [T] "+v"(T), [Cr] "+v"(Cr), [Cg] "+v"(Cg), [Cb] "+v"(Cb), [m] "+s"(m), [ia] "=&s"(ia), [ib] "=&s"(ib), [ic] "=&s"(ic)
        : [base] "v"(lds_chunk), [fpx] "v"(fpx), [fpy] "v"(fpy), [fa] "s"(fa), [p1] "i"(PLANE), [p2] "i"(2 * PLANE)
        : "vcc", "scc", "memory", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68",
          "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86",
          "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
}

__global__ __launch_bounds__(256) void kp(float *out, int chunks, unsigned long long ma, unsigned long long fa, unsigned long long *ticks)
{
    __shared__ float4 srec[3][128];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 128) {
        const int e = threadIdx.x;
        srec[0][e] = make_float4(4.0f + 0.01f * e, 3.5f, 0.0f, 0.0f);
        srec[1][e] = make_float4(-0.002f, 0.0005f, -0.003f, -3.0e38f);
        srec[2][e] = make_float4(-4.0f, 0.3f, 0.5f, 0.7f);
    }
    __syncthreads();
    const unsigned lds = (unsigned)(size_t)&srec[0][0];
    const float fpx = (float)(lane & 7), fpy = (float)(lane >> 3);
    float T = 1, Cr = 0, Cg = 0, Cb = 0;
    const unsigned long long t0 = wall_clock64();
    for (int c = 0; c < chunks; ++c) {
        walk1p(ma, fa, lds, fpx, fpy, T, Cr, Cg, Cb);
        T = T * 0.5f + 0.5f;
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = T + Cr + Cg + Cb;
}

__global__ __launch_bounds__(256) void k(float *out, int chunks, unsigned long long ma, unsigned long long mb, unsigned long long fa,
                                         unsigned long long fb, unsigned long long *ticks)
{
    __shared__ float4 srec[3][128];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 128) {
        const int e = threadIdx.x;
        srec[0][e] = make_float4(4.0f + 0.01f * e, 3.5f, 0.0f, 0.0f);           // mean inside the half-tile
        srec[1][e] = make_float4(-0.002f, 0.0005f, -0.003f, -3.0e38f);        // A, B, C (log2 domain): broad footprint
        srec[2][e] = make_float4(-4.0f, 0.3f, 0.5f, 0.7f);                     // log2(opacity) = -4: alpha ~ 0.06 > 1/255
    }
    __syncthreads();
    const unsigned lds = (unsigned)(size_t)&srec[0][0];
    const float fpxa = (float)(lane & 7), fpxb = fpxa + 8.0f, fpy = (float)(lane >> 3);
    float Ta = 1, Cra = 0, Cga = 0, Cba = 0, Tb = 1, Crb = 0, Cgb = 0, Cbb = 0;
    const unsigned long long t0 = wall_clock64();
    for (int c = 0; c < chunks; ++c) {
        walk(ma | mb, ma, mb, fa, fb, lds, fpxa, fpxb, fpy, Ta, Cra, Cga, Cba, Tb, Crb, Cgb, Cbb);
        Ta = Ta * 0.5f + 0.5f; Tb = Tb * 0.5f + 0.5f;  // keep T from underflowing; negligible next to 64 records
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    if (threadIdx.x == 0) { ticks[2 + 2 * blockIdx.x] = t0; ticks[3 + 2 * blockIdx.x] = t1; }  // every workgroup's start / end
    out[blockIdx.x * blockDim.x + threadIdx.x] = Ta + Cra + Cga + Cba + Tb + Crb + Cgb + Cbb;
}

int main()
{
    float *out; unsigned long long *ticks, h;
    hipMalloc(&out, 1 << 22); hipMalloc(&ticks, 8 * (2 + 2 * 4096));
    std::vector<unsigned long long> hb(2 + 2 * 4096);
    const int chunks = 2000;
    const unsigned long long ALL = ~0ull, ALT = 0x5555555555555555ull;
    struct { const char *name; unsigned long long ma, mb, fa, fb; } pat[] = {
        {"A only, unguarded          ", ALL, 0, ALL, 0},
        {"A only, guarded            ", ALL, 0, 0, 0},
        {"A and B, unguarded         ", ALL, ALL, ALL, ALL},
        {"A and B, guarded           ", ALL, ALL, 0, 0},
        {"A / B alternating, unguarded", ALT, ~ALT, ALL, ALL},
        {"A always, B every other, mixed guards", ALL, ALT, ALT, ~ALT},
    };
    for (int waves : {1, 2, 4, 8}) {  // waves per SIMD: 256 * waves workgroups of 4 waves, spread over the 256 CUs by the dispatcher
        for (auto &p : pat) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256 * waves), dim3(256), 0, 0, out, chunks, p.ma, p.mb, p.fa, p.fb, ticks);
            hipDeviceSynchronize();
            hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
            // residency check: if all 256 * waves workgroups really run together, the grid's span equals one workgroup's time
            hipMemcpy(hb.data(), ticks, 8 * (2 + 2 * 256 * waves), hipMemcpyDeviceToHost);
            unsigned long long lo = ~0ull, hi = 0;
            for (int b = 0; b < 256 * waves; ++b) { lo = std::min(lo, hb[2 + 2 * b]); hi = std::max(hi, hb[3 + 2 * b]); }
            printf("%d wave(s)/SIMD  %-38s %6.1f ns per record per wave   (grid span / workgroup 0's time = %.2f)\n", waves, p.name,
                   (double)h * 10.0 / ((double)chunks * 64.0), (double)(hi - lo) / (double)h);
        }
    }
    for (int waves : {1, 2, 4}) {
        struct { const char *name; unsigned long long ma, fa; } pp[] = {{"pipelined, A only, unguarded", ALL, ALL}, {"pipelined, A only, guarded  ", ALL, 0},
                                                                      {"pipelined, every other record", ALT, ALL}};
        for (auto &p : pp) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kp, dim3(256 * waves), dim3(256), 0, 0, out, chunks, p.ma, p.fa, ticks);
            hipDeviceSynchronize();
            hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
            const int per = p.ma == ALL ? 64 : 32;
            printf("%d wave(s)/SIMD  %-38s %6.1f ns per record per wave\n", waves, p.name, (double)h * 10.0 / ((double)chunks * per));
        }
    }
    return 0;
}
