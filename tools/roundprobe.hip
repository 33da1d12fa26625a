// Uncontended phase costs of the producer / consumer seed kernel's consumer wave: isaac_init_back<16> and isaac_round on
// the [256][40] u64 half-LDS layout, 2 waves x 40 lanes per CU, nothing else on the chip.
// hipcc --offload-arch=gfx950 -O3 -I hanamaru-renderer_amd/csrc tools/roundprobe.hip -o tools/bin/roundprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "isaac_core.h"
using namespace hr;
static const int LANES = 40;
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(const __attribute__((address_space(3))) void *)p; }
__device__ __forceinline__ u64 lds_load64(uint32_t a) { return *(const __attribute__((address_space(3))) u64 *)(size_t)a; }
struct Mem {
    u64 *col;
    __device__ __forceinline__ u64 ld(int i) const { return col[i * LANES]; }
    __device__ __forceinline__ uint32_t off(int i) const { return lds_addr(col) + (uint32_t)i * (uint32_t)(LANES * 8); }
    __device__ __forceinline__ u64 ldo(uint32_t o) const { return lds_load64(o); }
    __device__ __forceinline__ void st(int i, u64 v) { col[i * LANES] = v; }
};
struct Sink { u64 acc; __device__ void put(int, u64 v) { acc ^= v; } };
template <int MODE>
__global__ __launch_bounds__(128) void k(int reps, u64 *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= LANES) return;
    Mem m{reinterpret_cast<u64 *>(smem) + (size_t)wave * 256 * LANES + lane};
    for (int i = 0; i < 256; i++) m.st(i, (u64)(i * 0x9e3779b97f4a7c13ULL + lane * 77 + blockIdx.x));
    Sink s{0};
    u64 st16[16];
    for (int q = 0; q < 16; q++) st16[q] = (u64)q * 0x12345677ULL + lane;
    for (int r = 0; r < reps; r++) {
        if (MODE & 1) { st16[0] += r; isaac_init_back<16>(m, st16); }
        if (MODE & 2) isaac_round(m, s);
    }
    out[blockIdx.x * 128 + threadIdx.x] = s.acc + m.ld(5);
}
template <int MODE> float run(int reps, u64 *d) {
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(128), 163840, 0, reps, d);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(128), 163840, 0, reps, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    u64 *d; hipMalloc(&d, 256 * 128 * 8);
    int reps = 1620;
    float b = run<1>(reps, d), r = run<2>(reps, d), br = run<3>(reps, d);
    printf("reps %d per workgroup: init_back<16> %.2f ms (%.2f us each)  round %.2f ms (%.2f us each, %.1f ns/step)  both %.2f ms\n", reps, b, b / reps * 1e3, r,
           r / reps * 1e3, r / reps * 1e6 / 256, br);
    return 0;
}
