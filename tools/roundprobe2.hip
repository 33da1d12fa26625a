// Generators per lane: the consumer wave's round (isaac_round_n) with NG generators per lane on LANES = 40 / NG lanes, same
// [256][40] u64 half-LDS layout and the same 80 generator states per CU as the seed kernel; nothing else on the chip.
// hipcc --offload-arch=gfx950 -O3 -I hanamaru-renderer_amd/csrc tools/roundprobe2.hip -o tools/bin/roundprobe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include "isaac_core.h"
using namespace hr;
static const int COLS = 40;
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(const __attribute__((address_space(3))) void *)p; }
__device__ __forceinline__ u64 lds_load64(uint32_t a) { return *(const __attribute__((address_space(3))) u64 *)(size_t)a; }
struct Mem {
    u64 *col;
    __device__ __forceinline__ u64 ld(int i) const { return col[i * COLS]; }
    __device__ __forceinline__ uint32_t off(int i) const { return lds_addr(col) + (uint32_t)i * (uint32_t)(COLS * 8); }
    __device__ __forceinline__ u64 ldo(uint32_t o) const { return lds_load64(o); }
    __device__ __forceinline__ void st(int i, u64 v) { col[i * COLS] = v; }
};
struct Sink { u64 acc; __device__ void put(int, u64 v) { acc ^= v; } };
template <int NG, int MODE>
__global__ __launch_bounds__(128) void k(int reps, u64 *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LANES = COLS / NG;
    if (lane >= LANES) return;
    Mem m[NG];
    Sink s[NG];
    u64 st16[NG][16];
    for (int g = 0; g < NG; g++) {
        m[g].col = reinterpret_cast<u64 *>(smem) + (size_t)wave * 256 * COLS + lane + g * LANES;
        for (int i = 0; i < 256; i++) m[g].st(i, (u64)(i * 0x9e3779b97f4a7c13ULL + (lane + g * LANES) * 77 + blockIdx.x));
        s[g].acc = 0;
        for (int q = 0; q < 16; q++) st16[g][q] = (u64)q * 0x12345677ULL + lane + g * LANES;
    }
    for (int r = 0; r < reps; r++) {
        if (MODE & 1) for (int g = 0; g < NG; g++) { st16[g][0] += r; isaac_init_back<16>(m[g], st16[g]); }
        if (MODE & 2) isaac_round_n<28, NG>(m, s);
    }
    u64 a = 0;
    for (int g = 0; g < NG; g++) a += s[g].acc + m[g].ld(5);
    out[blockIdx.x * 128 + threadIdx.x] = a;
}
template <int NG, int MODE> float run(int reps, u64 *d) {
    hipFuncSetAttribute((const void *)k<NG, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<NG, MODE>), dim3(256), dim3(128), 163840, 0, reps, d);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NG, MODE>), dim3(256), dim3(128), 163840, 0, reps, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
template <int NG> void report(int reps, u64 *d) {
    float b = run<NG, 1>(reps, d), r = run<NG, 2>(reps, d), br = run<NG, 3>(reps, d);
    printf("NG %d (%2d lanes x 2 waves): init_back<16> %.2f us  round %.2f us (%.1f ns/step)  both %.2f us   per group of 80 states\n", NG, COLS / NG,
           b / reps * 1e3, r / reps * 1e3, r / reps * 1e6 / 256, br / reps * 1e3);
}
int main() {
    u64 *d; hipMalloc(&d, 256 * 128 * 8);
    int reps = 800;
    report<1>(reps, d); report<2>(reps, d); report<4>(reps, d); report<5>(reps, d);
    return 0;
}
