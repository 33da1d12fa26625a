// Probe: (a) does a wave64 VALU op with only 32 active lanes cost one pass?  (b) 64-bit vs 32-bit-pair formulations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "isaac_core.h"
using namespace hr;
struct LdsMem { u64 *col; __device__ u64 ld(int i) const { return col[i * 64]; } __device__ uint32_t off(int i) const { return (uint32_t)i * (uint32_t)(64 * 8); }
    __device__ u64 ldo(uint32_t o) const { return *reinterpret_cast<const u64 *>(reinterpret_cast<const unsigned char *>(col) + o); } __device__ void st(int i, u64 v) { col[i * 64] = v; } };
struct Sink { u64 acc; __device__ void put(int, u64 v) { acc ^= v; } };

// 1 wave of 64 lanes  vs  2 waves with 32 active lanes each (same 64 LDS columns)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void kfull(int items, u64 *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    u64 *mem = (u64 *)smem;
    const IsaacWarm warm = isaac_warm();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = 64 / WAVES;
    if (lane >= per) return;
    LdsMem m{mem + wave * per + lane};
    Sink s{0};
    for (int it = blockIdx.x; it < items; it += gridDim.x) isaac_seed_round(m, warm, 8700304ULL, (u64)it, (u64)threadIdx.x, 7ULL, s);
    out[blockIdx.x * 64 + wave * per + lane] = s.acc;
}

// pure VALU chains, no memory: 64-bit ops as written vs 32-bit pairs
__global__ __launch_bounds__(64) void kalu64(int iters, u64 *out) {
    u64 a = threadIdx.x + 1, b = a * 3, c = a * 5, d = a * 7, e = a * 11, f = a * 13, g = a * 17, h = a * 19;
    for (int i = 0; i < iters; i++) { HR_ISAAC_MIX(a, b, c, d, e, f, g, h) }
    out[threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
struct U2 { unsigned lo, hi; };
__device__ __forceinline__ U2 sub2(U2 a, U2 b) { U2 r; r.lo = a.lo - b.lo; r.hi = a.hi - b.hi - (a.lo < b.lo ? 1u : 0u); return r; }
__device__ __forceinline__ U2 add2(U2 a, U2 b) { U2 r; r.lo = a.lo + b.lo; r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u); return r; }
template <int K> __device__ __forceinline__ U2 shr2(U2 a) { U2 r; r.lo = __builtin_amdgcn_alignbit(a.hi, a.lo, K); r.hi = a.hi >> K; return r; }
template <int K> __device__ __forceinline__ U2 shl2(U2 a) { U2 r; r.hi = __builtin_amdgcn_alignbit(a.hi, a.lo, 32 - K); r.lo = a.lo << K; return r; }
__device__ __forceinline__ U2 xor2(U2 a, U2 b) { U2 r; r.lo = a.lo ^ b.lo; r.hi = a.hi ^ b.hi; return r; }
__global__ __launch_bounds__(64) void kalu32(int iters, u64 *out) {
    U2 a{threadIdx.x + 1, 1}, b{a.lo * 3, 2}, c{a.lo * 5, 3}, d{a.lo * 7, 4}, e{a.lo * 11, 5}, f{a.lo * 13, 6}, g{a.lo * 17, 7}, h{a.lo * 19, 8};
    for (int i = 0; i < iters; i++) {
        a = sub2(a, e); f = xor2(f, shr2<9>(h));  h = add2(h, a);
        b = sub2(b, f); g = xor2(g, shl2<9>(a));  a = add2(a, b);
        c = sub2(c, g); h = xor2(h, shr2<23>(b)); b = add2(b, c);
        d = sub2(d, h); a = xor2(a, shl2<15>(c)); c = add2(c, d);
        e = sub2(e, a); b = xor2(b, shr2<14>(d)); d = add2(d, e);
        f = sub2(f, b); c = xor2(c, shl2<20>(e)); e = add2(e, f);
        g = sub2(g, c); d = xor2(d, shr2<17>(f)); f = add2(f, g);
        h = sub2(h, d); e = xor2(e, shl2<14>(g)); g = add2(g, h);
    }
    out[threadIdx.x] = ((u64)(a.hi ^ b.hi ^ c.hi ^ d.hi ^ e.hi ^ f.hi ^ g.hi ^ h.hi) << 32) | (a.lo ^ b.lo ^ c.lo ^ d.lo ^ e.lo ^ f.lo ^ g.lo ^ h.lo);
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    u64 *d; hipMalloc(&d, 256 * 64 * 8);
    int items = 129600;
    hipFuncSetAttribute((const void *)kfull<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)kfull<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)kfull<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    float t1 = timeit([&] { hipLaunchKernelGGL(kfull<1>, dim3(256), dim3(64), 131072, 0, items, d); });
    float t2 = timeit([&] { hipLaunchKernelGGL(kfull<2>, dim3(256), dim3(128), 131072, 0, items, d); });
    float t4 = timeit([&] { hipLaunchKernelGGL(kfull<4>, dim3(256), dim3(256), 131072, 0, items, d); });
    printf("seed round, %d items: 1x64 lanes %.2f ms | 2 waves x32 lanes %.2f ms | 4 waves x16 lanes %.2f ms\n", items, t1, t2, t4);
    int iters = 100000;
    float a64 = timeit([&] { hipLaunchKernelGGL(kalu64, dim3(256), dim3(64), 0, 0, iters, d); });
    float a32 = timeit([&] { hipLaunchKernelGGL(kalu32, dim3(256), dim3(64), 0, 0, iters, d); });
    printf("mix chain x%d, one wave/CU: u64 ops %.2f ms (%.1f cyc/mix)  32-bit pairs %.2f ms (%.1f cyc/mix)\n", iters, a64, a64 * 1e-3 * 2.1e9 / iters, a32, a32 * 1e-3 * 2.1e9 / iters);
    return 0;
}
