// probe_kernels.hip — synthetic HBM-bound kernels for tools/overlap_probe.py: which property lets a kernel run UNDER the blend?
// All move the same bytes (n float4 read + written); they differ in wave lifetime, VGPR footprint and dependent round trips.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probe_kernels.hip -o tools/libprobe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

// v0: short-lived waves, one float4 per thread
__global__ __launch_bounds__(256) void k_short(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
// v1: long-lived waves: ITER float4 per thread, one after the other (each load waits for the previous store's issue only)
template <int ITER>
__global__ __launch_bounds__(256) void k_long(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n)
{
    size_t i = ((size_t)blockIdx.x * ITER) * 256 + threadIdx.x;
    for (int k = 0; k < ITER; ++k, i += 256)
        if (i < n) {
            float4 v = s[i];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // serialise: one round trip per iteration
            d[i] = v;
        }
}
// v2: short-lived, but 16 float4 per thread all in flight (64 VGPRs of payload): a fat wave
__global__ __launch_bounds__(256) void k_fat(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n)
{
    const size_t base = ((size_t)blockIdx.x * 16) * 256 + threadIdx.x;
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = base + (size_t)k * 256 < n ? s[base + (size_t)k * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 16; ++k) if (base + (size_t)k * 256 < n) d[base + (size_t)k * 256] = v[k];
}
// v3: two dependent round trips (load an index, then gather), one float4 per thread
__global__ __launch_bounds__(256) void k_dep(const float4 *__restrict__ s, float4 *__restrict__ d, const uint32_t *__restrict__ idx, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[idx[i]];
}
// v4: one round trip, then ~600 VALU instructions of arithmetic, then the store (a compute tail like the preprocess has)
__global__ __launch_bounds__(256) void k_alu(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 v = s[i];
#pragma unroll 1
    for (int k = 0; k < 150; ++k) { v.x = fmaf(v.x, 1.0001f, v.y); v.y = fmaf(v.y, 0.9999f, v.z); v.z = fmaf(v.z, 1.0002f, v.w); v.w = fmaf(v.w, 0.9998f, v.x); }
    d[i] = v;
}

extern "C" int probe_launch(int variant, const void *s, void *d, const void *idx, size_t n, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const float4 *S = (const float4 *)s; float4 *D = (float4 *)d;
    const unsigned g1 = (unsigned)((n + 255) / 256);
    switch (variant) {
    case 0: hipLaunchKernelGGL(k_short, dim3(g1), dim3(256), 0, st, S, D, n); break;
    case 1: hipLaunchKernelGGL(k_long<32>, dim3((g1 + 31) / 32), dim3(256), 0, st, S, D, n); break;
    case 2: hipLaunchKernelGGL(k_fat, dim3((g1 + 15) / 16), dim3(256), 0, st, S, D, n); break;
    case 3: hipLaunchKernelGGL(k_dep, dim3(g1), dim3(256), 0, st, S, D, (const uint32_t *)idx, n); break;
    case 4: hipLaunchKernelGGL(k_alu, dim3(g1), dim3(256), 0, st, S, D, n); break;
    case 5: hipLaunchKernelGGL(k_long<4>, dim3((g1 + 3) / 4), dim3(256), 0, st, S, D, n); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
