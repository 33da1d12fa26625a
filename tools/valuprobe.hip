// How long a wave64 VALU instruction occupies a SIMD on gfx950: W waves per SIMD, each a long stream of independent instructions (8 chains,
// 512 per loop body), every CU busy.  Result (profiles/r05_valuprobe.txt, cycles per instruction per SIMD at 2.4 GHz): v_fma_f32 5.3 for a lone wave
// (issue-limited), 2.9 with two to four waves, 2.5 with eight; v_pk_fma_f32 4.5 - 5.1 (two flops per lane: the same flop rate); v_lshl_add_u64 5.0 - 5.6
// (the seed kernel's 64-bit integer mixing runs at half the fp32 rate); a 40-lane wave's instruction costs what a full one costs, a 16-lane one
// slightly MORE (3.4).  Confirms tools/simdprobe.hip (profiles/r02_simdprobe.txt: 2.4 cycles with four waves) with hand-written streams.
// hipcc --offload-arch=gfx950 -O3 tools/valuprobe.hip -o tools/bin/valuprobe
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void stream(int iters, int lanes, float *out) {
    const int lane = threadIdx.x & 63;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    unsigned long long u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    const float m = 0.999f, c = 1e-3f;
    const f2 pm = {m, m}, pc = {c, c};
    if (lane < lanes) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 64; k++) {
                if (KIND == 0) {
                    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
                } else if (KIND == 1) {
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                                 "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
                } else {
                    asm volatile("v_lshl_add_u64 %0, %0, 1, %4\n v_lshl_add_u64 %1, %1, 1, %4\n v_lshl_add_u64 %2, %2, 1, %4\n v_lshl_add_u64 %3, %3, 1, %4\n"
                                 "v_lshl_add_u64 %0, %0, 1, %4\n v_lshl_add_u64 %1, %1, 1, %4\n v_lshl_add_u64 %2, %2, 1, %4\n v_lshl_add_u64 %3, %3, 1, %4"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0 | 1ull));
                }
            }
        }
    }
    const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 + u1 + u2 + u3);
    if (r == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
static void run(const char *name, int waves_per_simd, int lanes, float *out) {
    const int iters = 400, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    stream<KIND><<<blocks, 256>>>(20, lanes, out);
    hipEventRecord(e0);
    stream<KIND><<<blocks, 256>>>(iters, lanes, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = (double)iters * 64 * 8;                       // VALU instructions of one wave
    const double simd_cycles = ms * 1e-3 * 2.4e9;
    printf("%-16s waves/SIMD %d  lanes %2d : %7.3f ms  %5.2f cycles per instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, lanes, ms, simd_cycles / (per_wave * waves_per_simd));
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    float *out;
    hipMalloc(&out, 1 << 24);
    for (int w : {1, 2, 4, 5, 8}) {
        run<0>("v_fma_f32", w, 64, out);
        run<1>("v_pk_fma_f32", w, 64, out);
        run<2>("v_lshl_add_u64", w, 64, out);
    }
    run<0>("v_fma_f32", 4, 40, out);
    run<0>("v_fma_f32", 4, 16, out);
    run<2>("v_lshl_add_u64", 1, 40, out);
    return 0;
}
