// The consumer wave's round, production form (isaac_round<28>) against an experimental paired-operand form (isaac_round_pair28 below:
// the two look-ahead operands of a step in ONE ds_read2st64_b64, 17 instead of 18 issue slots per step): cycles per step, and the two
// forms' outputs compared word for word.  Result (profiles/r03_roundprobe4.txt): bit-identical, 107.6 vs 107.8 cycles per step — the
// step is bound by its serial chain (gather latency + add + store + index), not by the issue slot that was saved; not adopted.
// hipcc --offload-arch=gfx950 -O3 -I hanamaru-renderer_amd/csrc tools/roundprobe4.hip -o tools/bin/roundprobe4
#include <hip/hip_runtime.h>
#include <cstdio>
#include "isaac_core.h"
using namespace hr;
namespace hr {
// The same round with the two look-ahead operands of a step fetched by ONE LDS instruction.  x_k = mem[k] and the +128 operand of
// step k, mem[(k + 128) & 255], sit exactly 128 rows apart — 80 x 512 bytes in the [row][40 columns] layout — so one
// ds_read2st64_b64 (two 8-byte reads at base + offset0 x 512 and base + offset1 x 512) fetches both, provided the base register
// points at a row congruent to k modulo 8 (8 rows = 5 x 512 bytes).  Eight base pointers, one per residue, advanced once per
// 40-step loop body, give every step its pair with immediate offsets only: 17 instead of 18 issue slots per step (the compiler
// forms the instruction from the two adjacent loads).  Both operands are now four steps ahead (the +128 operand was three): one
// more register pair in the pipeline.  Mem: as for isaac_round, plus COLS, rowp (pointer to a row entry of this column),
// rowp row(int i), u64 at(rowp p, int rows) = p[rows].
#if defined(__HIP_DEVICE_COMPILE__)
#define HR_OPAQUEP(p) asm volatile("" : "+v"(p))
#else
#define HR_OPAQUEP(p)
#endif
template <class Mem, class Tail>
HD void isaac_round_pair28(Mem &mem, Tail &tail) {
    typedef typename Mem::rowp rowp;
    u64 aa, x0, x1, x2, x3, m1, m2, m3, g1, T, g2, xprev;
    uint32_t p1;
    x0 = mem.ld(0); x1 = mem.ld(1); x2 = mem.ld(2); x3 = mem.ld(3);
    g1 = mem.ld((int)((x0 >> 3) & 255));
    aa = ~(u64)0 + mem.ld(128);
    m1 = mem.ld(129); m2 = mem.ld(130); m3 = mem.ld(131);
    T = g1 + aa + 1; g2 = 0; xprev = 0;
    p1 = mem.off((int)((x1 >> 3) & 255));
    // b<r>: row R0 + r, R0 (a multiple of 8) = the first row the current body fetches (first half) or that row - 128 (second half)
    rowp b0 = mem.row(0), b1 = mem.row(1), b2 = mem.row(2), b3 = mem.row(3), b4 = mem.row(4), b5 = mem.row(5), b6 = mem.row(6), b7 = mem.row(7);
#define PB(j) ((((j) & 7) == 0) ? b0 : (((j) & 7) == 1) ? b1 : (((j) & 7) == 2) ? b2 : (((j) & 7) == 3) ? b3 : (((j) & 7) == 4) ? b4 : (((j) & 7) == 5) ? b5 : (((j) & 7) == 6) ? b6 : b7)
    // step N (its store goes to row N + R); the pair it fetches: x for step N + 4 at PB(J) + (J / 8) * 8 + XO rows, its +128 operand at + MO
#define PSTEP(N, R, MIXEXPR_NEXT, J, XO, MO, TAILPREV)                                              \
    {                                                                                               \
        HR_OPAQUE32(p1);                                                                            \
        HR_SCHED_FENCE();                                                                           \
        u64 y = T + g2;                                                                             \
        mem.st((N) + (R), y);                                                                       \
        g1 = mem.ldo(p1);                                                                           \
        u64 g2n = mem.ld((int)((y >> 11) & 255));                                                   \
        HR_SCHED_FENCE();                                                                           \
        if (TAILPREV) tail.put((N) - 1, g2 + xprev);                                                \
        g2 = g2n;                                                                                   \
        p1 = mem.off((int)((x2 >> 3) & 255));                                                       \
        u64 x4 = mem.at(PB(J), ((J) >> 3) * 8 + (XO));                                              \
        u64 m4 = mem.at(PB(J), ((J) >> 3) * 8 + (MO));                                              \
        const u64 A_ = aa;                                                                          \
        aa = (MIXEXPR_NEXT) + m1;                                                                   \
        T = g1 + (aa + x0);                                                                         \
        HR_OPAQUE64(T);                                                                             \
        xprev = x0; x0 = x1; x1 = x2; x2 = x3; x3 = x4; m1 = m2; m2 = m3; m3 = m4;                  \
    }
#define PSTEP4(n, R, J, XO, MO, TP0, TP)                                   \
    PSTEP((n), R, A_ ^ (A_ >> 5), (J), XO, MO, TP0)                        \
    PSTEP((n) + 1, R, A_ ^ (A_ << 12), (J) + 1, XO, MO, TP)                \
    PSTEP((n) + 2, R, A_ ^ (A_ >> 33), (J) + 2, XO, MO, TP)                \
    PSTEP((n) + 3, R, ~(A_ ^ (A_ << 21)), (J) + 3, XO, MO, TP)
#define PSTEP20(n, R, J, XO, MO, TP) PSTEP4(n, R, J, XO, MO, TP, TP) PSTEP4((n) + 4, R, (J) + 4, XO, MO, TP, TP) PSTEP4((n) + 8, R, (J) + 8, XO, MO, TP, TP) \
    PSTEP4((n) + 12, R, (J) + 12, XO, MO, TP, TP) PSTEP4((n) + 16, R, (J) + 16, XO, MO, TP, TP)
#define PB_ADVANCE(ROWS) { b0 += (ROWS) * Mem::COLS; b1 += (ROWS) * Mem::COLS; b2 += (ROWS) * Mem::COLS; b3 += (ROWS) * Mem::COLS; b4 += (ROWS) * Mem::COLS; b5 += (ROWS) * Mem::COLS; b6 += (ROWS) * Mem::COLS; b7 += (ROWS) * Mem::COLS; }
#define PB_KEEP() { HR_OPAQUEP(b0); HR_OPAQUEP(b1); HR_OPAQUEP(b2); HR_OPAQUEP(b3); HR_OPAQUEP(b4); HR_OPAQUEP(b5); HR_OPAQUEP(b6); HR_OPAQUEP(b7); }
    // steps 0..3 fetch rows 4..7 (R0 = 0)
    PB_KEEP();
    PSTEP4(0, 0, 4, 0, 128, false, false)
    // steps 4..123 fetch rows 8..127: three bodies of 40 (the x pipeline has period 5, the operand pipeline and the mix period 4:
    // no register copies inside a body)
    PB_ADVANCE(8);
    HR_NOUNROLL
    for (int n = 4; n < 124; n += 40) {
        PB_KEEP();
        PSTEP20(n, 0, 0, 0, 128, false) PSTEP20(n + 20, 0, 20, 0, 128, false)
        PB_ADVANCE(40);
    }
    // steps 124..203 fetch rows 128..207: the operand of row k is now row k - 128, so the bases move DOWN 128 rows and the roles of the
    // two offsets swap (both stay >= 0)
    PB_ADVANCE(-128);
    HR_NOUNROLL
    for (int n = 124; n < 204; n += 40) {
        PB_KEEP();
        PSTEP20(n, 0, 0, 128, 0, false) PSTEP20(n + 20, 0, 20, 128, 0, false)
        PB_ADVANCE(40);
    }
    // steps 204..255, straight-line, fetch rows 208..255 (the last four fetches are never used: they repeat valid rows); the store rows
    // are written as (row - 128) + tb with an opaque tb = 128 (constant rows beyond 65,535 bytes would each get an address register)
    int tb = 128;
    HR_OPAQUE32(tb);
    PB_KEEP();
#define PTAILR (tb - 128)
    PSTEP4(204, PTAILR, 0, 128, 0, false, false)
    PSTEP20(208, PTAILR, 4, 128, 0, false)
    PSTEP4(228, PTAILR, 24, 128, 0, false, true)
    PSTEP20(232, PTAILR, 28, 128, 0, true)
    PSTEP4(252, PTAILR, 40, 128, 0, true, true)
    tail.put(255, g2 + xprev);
#undef PTAILR
#undef PSTEP
#undef PSTEP4
#undef PSTEP20
#undef PB
#undef PB_ADVANCE
#undef PB_KEEP
}
}  // namespace hr
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(const __attribute__((address_space(3))) void *)p; }
__device__ __forceinline__ u64 lds_load64(uint32_t a) { return *(const __attribute__((address_space(3))) u64 *)(size_t)a; }
typedef __attribute__((address_space(3))) u64 *lds_rowp;
template <int C>
struct Mem {
    static const int COLS = C;
    typedef lds_rowp rowp;
    u64 *col;
    __device__ __forceinline__ rowp row(int i) const { return (rowp)col + i * COLS; }
    __device__ __forceinline__ u64 at(rowp p, int rows) const { return p[rows * COLS]; }
    __device__ __forceinline__ u64 ld(int i) const { return col[i * COLS]; }
    __device__ __forceinline__ uint32_t off(int i) const { return lds_addr(col) + (uint32_t)i * (uint32_t)(COLS * 8); }
    __device__ __forceinline__ u64 ldo(uint32_t o) const { return lds_load64(o); }
    __device__ __forceinline__ void st(int i, u64 v) { col[i * COLS] = v; }
};
struct Sink { u64 acc; __device__ void put(int step, u64 v) { acc = (acc ^ v) * 0x9e3779b97f4a7c13ULL + (u64)step; } };
template <int COLS, int LANES, int VAR>
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
    for (int r = 0; r < reps; r++) { if (VAR == 0) isaac_round<28>(m, s); else isaac_round_pair28(m, s); }
    const unsigned long long t1 = __builtin_readcyclecounter();
    u64 h = s.acc;
    for (int i = 0; i < 256; i++) h = (h ^ m.ld(i)) * 0xff51afd7ed558ccdULL;
    out[blockIdx.x * 128 + threadIdx.x] = h;
    if (lane == 0 && wave == 0) out[256 * 128 + blockIdx.x] = t1 - t0;
}
template <int COLS, int LANES, int VAR> void report(int waves, int reps, u64 *d, u64 *hashes) {
    const int lds = waves * 256 * COLS * 8;
    hipFuncSetAttribute((const void *)k<COLS, LANES, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipMemset(d, 0, (256 * 128 + 256) * 8);
    hipLaunchKernelGGL((k<COLS, LANES, VAR>), dim3(256), dim3(64 * waves), lds, 0, reps, d);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<COLS, LANES, VAR>), dim3(256), dim3(64 * waves), lds, 0, reps, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    static u64 h[256 * 128 + 256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < 256; i++) cyc += (double)h[256 * 128 + i];
    cyc /= 256.0 * reps * 256.0;
    for (int i = 0; i < 256 * 128; i++) hashes[i] = h[i];
    printf("%s, %d wave(s) per CU: %.1f ns/step, %.1f shader cycles/step\n", VAR ? "paired operands (isaac_round_pair28)" : "production round (isaac_round<28>)  ", waves, ms / reps * 1e6 / 256, cyc);
}
int main() {
    u64 *d; hipMalloc(&d, (256 * 128 + 256) * 8);
    static u64 ha[256 * 128], hb[256 * 128];
    int reps = 800;
    report<40, 40, 0>(2, reps, d, ha);
    report<40, 40, 1>(2, reps, d, hb);
    size_t bad = 0;
    for (int i = 0; i < 256 * 128; i++) bad += ha[i] != hb[i];
    printf("state + tail hashes of %d generators after %d rounds: %zu differ\n", 256 * 80, reps, bad);
    report<40, 40, 0>(1, reps, d, ha);
    report<40, 40, 1>(1, reps, d, hb);
    return bad != 0;
}
