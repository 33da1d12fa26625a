// LDS dependent-gather latency probe: each active lane chases indices through its own column of a [256][COLS] table
// (the access pattern of the ISAAC-64 round's serial chain).  hipcc --offload-arch=gfx950 -O3 tools/ldsprobe.hip -o ldsprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
template <typename T, int COLS>
__global__ void chase(int lanes, int iters, u64 *out) {
    extern __shared__ unsigned char smem[];
    T *tab = reinterpret_cast<T *>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = wave * lanes + lane;
    const bool act = lane < lanes;
    if (act) for (int i = 0; i < 256; i++) tab[i * COLS + col] = (T)((i * 167 + col * 31 + 13) & 255);
    __syncthreads();
    if (!act) return;
    u64 t0 = wall_clock64();
    unsigned idx = col & 255;
    for (int k = 0; k < iters; k++) idx = (unsigned)tab[idx * COLS + col] & 255u;
    u64 t1 = wall_clock64();
    if (lane == 0) out[blockIdx.x * 8 + wave] = (t1 - t0);
    if (lane == 1) out[4096 + blockIdx.x * 8 + wave] = idx;
}
template <typename T, int COLS>
void run(const char *name, int waves, int lanes) {
    u64 *d; hipMalloc(&d, 8192 * 8); hipMemset(d, 0, 8192 * 8);
    const int iters = 20000;
    size_t lds = (size_t)256 * COLS * sizeof(T);
    hipFuncSetAttribute((const void *)chase<T, COLS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    chase<T, COLS><<<256, 64 * waves, lds>>>(lanes, iters, d);
    hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
    if (e1 != hipSuccess || e2 != hipSuccess) printf("launch: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    std::vector<u64> h(8192); hipMemcpy(h.data(), d, 8192 * 8, hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int b = 0; b < 256; b++) for (int w = 0; w < waves; w++) { s += (double)h[b * 8 + w]; n++; }
    printf("%-28s waves %d lanes %2d: %.1f clock64 ticks per dependent gather (x%.1f = shader cycles at 2.4 GHz / 100 MHz)\n", name, waves, lanes, s / n / iters, 24.0);
    hipFree(d);
}
int main() {
    run<u64, 80>("u64 [256][80]", 2, 40);
    run<u64, 80>("u64 [256][80]", 1, 40);
    run<u64, 80>("u64 [256][80]", 4, 20);
    run<unsigned, 80>("u32 [256][80]", 2, 40);
    run<unsigned, 160>("u32 [256][160]", 4, 40);
    run<u64, 64>("u64 [256][64]", 1, 64);
    run<unsigned, 64>("u32 [256][64]", 1, 64);
    run<unsigned, 64>("u32 [256][64]", 1, 16);
    return 0;
}
