// The consumer wave's round under different LDS layouts: row stride (u64 columns per row) x active lanes.  A gather mem[idx][lane]
// with a random idx per lane is bank-conflict-free only when the row stride is a multiple of 256 B (32 columns): with 40 columns
// per row (the seed kernel's 80 KiB half) lanes c and c + 8k collide whenever their row indices differ by the right amount.
// The first rows compare the two formulations of the round in isaac_core.h: the production one (look-ahead operands one step deeper,
// two waits per step, 20-step loop bodies: isaac_round<28>) and the one it replaced (isaac_round_n<28, 1>: three waits per step).
// hipcc --offload-arch=gfx950 -O3 -I hanamaru-renderer_amd/csrc tools/roundprobe3.hip -o tools/bin/roundprobe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include "isaac_core.h"
using namespace hr;
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(const __attribute__((address_space(3))) void *)p; }
__device__ __forceinline__ u64 lds_load64(uint32_t a) { return *(const __attribute__((address_space(3))) u64 *)(size_t)a; }
template <int COLS>
struct Mem {
    u64 *col;
    __device__ __forceinline__ u64 ld(int i) const { return col[i * COLS]; }
    __device__ __forceinline__ uint32_t off(int i) const { return lds_addr(col) + (uint32_t)i * (uint32_t)(COLS * 8); }
    __device__ __forceinline__ u64 ldo(uint32_t o) const { return lds_load64(o); }
    __device__ __forceinline__ void st(int i, u64 v) { col[i * COLS] = v; }
};
struct Sink { u64 acc; __device__ void put(int, u64 v) { acc ^= v; } };
template <int COLS, int LANES, bool OLD = false>
__global__ __launch_bounds__(128) void k(int reps, u64 *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= LANES) return;
    Mem<COLS> m;
    Sink s;
    m.col = reinterpret_cast<u64 *>(smem) + (size_t)wave * 256 * COLS + lane;
    for (int i = 0; i < 256; i++) m.st(i, (u64)(i * 0x9e3779b97f4a7c13ULL + lane * 77 + blockIdx.x) * 0xff51afd7ed558ccdULL);
    s.acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (OLD) { for (int r = 0; r < reps; r++) isaac_round_n<28, 1>(&m, &s); } else { for (int r = 0; r < reps; r++) isaac_round<28>(m, s); }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 128 + threadIdx.x] = s.acc + m.ld(5);
    if (lane == 0 && wave == 0) out[256 * 128 + blockIdx.x] = t1 - t0;
}
template <int COLS, int LANES, bool OLD = false> void report(int waves, int reps, u64 *d) {
    const int lds = waves * 256 * COLS * 8;
    hipFuncSetAttribute((const void *)k<COLS, LANES, OLD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<COLS, LANES, OLD>), dim3(256), dim3(64 * waves), lds, 0, reps, d);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<COLS, LANES, OLD>), dim3(256), dim3(64 * waves), lds, 0, reps, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    static u64 h[256];
    hipMemcpy(h, d + 256 * 128, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < 256; i++) cyc += (double)h[i];
    cyc /= 256.0 * reps * 256.0;
    printf("%srow stride %2d columns, %2d lanes, %d wave(s) per CU: %.1f ns/step, %.1f shader cycles/step (s_memtime)  -> %.2f GHz\n", OLD ? "three-wait round: " : "", COLS, LANES, waves, ms / reps * 1e6 / 256, cyc, cyc / (ms / reps * 1e6 / 256));
}
int main() {
    u64 *d; hipMalloc(&d, (256 * 128 + 256) * 8);
    int reps = 800;
    report<40, 40>(2, reps, d);
    report<40, 40, true>(2, reps, d);
    report<40, 32>(2, reps, d);
    report<32, 32>(2, reps, d);
    report<32, 16>(2, reps, d);
    report<16, 16>(2, reps, d);
    report<64, 40>(1, reps, d);
    report<64, 64>(1, reps, d);
    report<40, 40>(1, reps, d);
    return 0;
}
