// Phase timing probe for the ISAAC-64 seed kernel (run on the GPU box): full / init only / round only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "isaac_core.h"
using namespace hr;
struct LdsMem { u64 *col; __device__ u64 ld(int i) const { return col[i * 64]; } __device__ uint32_t off(int i) const { return (uint32_t)i * (uint32_t)(64 * 8); }
    __device__ u64 ldo(uint32_t o) const { return *reinterpret_cast<const u64 *>(reinterpret_cast<const unsigned char *>(col) + o); } __device__ void st(int i, u64 v) { col[i * 64] = v; } };
struct Sink { u64 acc; __device__ void put(int, u64 v) { acc ^= v; } };
template <int MODE>
__global__ __launch_bounds__(64) void k(int items, u64 *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    u64 *mem = (u64 *)smem;
    const IsaacWarm warm = isaac_warm();
    LdsMem m{mem + threadIdx.x};
    Sink s{0};
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        if (MODE == 0) isaac_seed_round(m, warm, 8700304ULL, (u64)it, (u64)threadIdx.x, 7ULL, s);
        if (MODE == 1) {  // init passes only
            u64 a = warm.r[0] + it, b = warm.r[1], c = warm.r[2] + threadIdx.x, d = warm.r[3], e = warm.r[4], f = warm.r[5], g = warm.r[6], h = warm.r[7];
            _Pragma("unroll 1") for (int i = 0; i < 256; i += 8) { HR_ISAAC_MIX(a, b, c, d, e, f, g, h) m.st(i, a); m.st(i+1, b); m.st(i+2, c); m.st(i+3, d); m.st(i+4, e); m.st(i+5, f); m.st(i+6, g); m.st(i+7, h); }
            _Pragma("unroll 1") for (int i = 0; i < 256; i += 8) { a += m.ld(i); b += m.ld(i+1); c += m.ld(i+2); d += m.ld(i+3); e += m.ld(i+4); f += m.ld(i+5); g += m.ld(i+6); h += m.ld(i+7);
                HR_ISAAC_MIX(a, b, c, d, e, f, g, h) m.st(i, a); m.st(i+1, b); m.st(i+2, c); m.st(i+3, d); m.st(i+4, e); m.st(i+5, f); m.st(i+6, g); m.st(i+7, h); }
            s.acc ^= a;
        }
        if (MODE == 2) {  // one round only on whatever is in LDS
            u64 aa = it, bb = 1;
            u64 x = m.ld(0);
            _Pragma("unroll 1") for (int n = 0; n < 256; n += 4) {
                for (int j = 0; j < 4; j++) {
                    u64 mixv = j == 0 ? ~(aa ^ (aa << 21)) : j == 1 ? aa ^ (aa >> 5) : j == 2 ? aa ^ (aa << 12) : aa ^ (aa >> 33);
                    aa = mixv + m.ld((n + j + 128) & 255);
                    u64 y = m.ld((int)((x >> 3) & 255)) + aa + bb;
                    m.st(n + j, y);
                    bb = m.ld((int)((y >> 11) & 255)) + x;
                    x = m.ld((n + j + 1) & 255);
                }
            }
            s.acc ^= bb;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = s.acc;
}
template <int MODE> float run(int items, u64 *d) {
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64), 131072, 0, items, d);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64), 131072, 0, items, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    u64 *d; hipMalloc(&d, 256 * 64 * 8);
    int items = 129600;
    float f = run<0>(items, d), i = run<1>(items, d), r = run<2>(items, d);
    double per = 256.0 / items * 1e-3 * 2.1e9;
    printf("items %d: full %.2f ms (%.0f cyc/item @2.1GHz)  init %.2f ms (%.0f)  naive-round %.2f ms (%.0f)\n", items, f, f * per, i, i * per, r, r * per);
    return 0;
}
