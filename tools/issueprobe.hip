// What does ONE wave pay per instruction?  The seed kernel's consumer wave is alone on its SIMD for its critical path, so its
// speed is set by how fast a single wave can issue.  Each test is a loop of 64 copies of one instruction pattern (independent
// unless noted), 40 active lanes, one wave per CU on 2 SIMDs (2 waves per workgroup like the consumers); cycles per
// instruction from s_memtime around the loop.
// hipcc --offload-arch=gfx950 -O3 tools/issueprobe.hip -o tools/bin/issueprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
typedef unsigned long long u64;

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define TEST_BEGIN(ID)                                                             \
    if (test == ID) {                                                                \
        u64 t0 = __builtin_readcyclecounter();                                       \
        for (int it = 0; it < iters; it++) {
#define TEST_END(N)                                                                  \
        }                                                                            \
        u64 t1 = __builtin_readcyclecounter();                                       \
        cyc = (double)(t1 - t0) / ((double)iters * (N));                            \
    }

__global__ __launch_bounds__(128) void k(int test, int iters, int lanes, float *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= lanes) return;
    unsigned a0 = lane * 8 + wave * 81920, a1 = a0 + 320, a2 = a0 + 640, a3 = a0 + 960;
    u64 r0 = lane, r1 = lane + 1, r2 = lane + 2, r3 = lane + 3, r4 = 5, r5 = 6, r6 = 7, r7 = 8;
    unsigned w0 = lane, w1 = lane * 3, w2 = 7, w3 = 9;
    double cyc = 0;
    // ---- VALU
    TEST_BEGIN(0) asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(w0) : "v"(w1)); TEST_END(64)                      // dependent 32-bit
    TEST_BEGIN(1) asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n") : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(lane)); TEST_END(64)   // 4 independent chains
    TEST_BEGIN(2) asm volatile(REP64("v_lshl_add_u64 %0, %0, 0, %1\n") : "+v"(r0) : "v"(r1)); TEST_END(64)              // dependent 64-bit add
    TEST_BEGIN(3) asm volatile(REP16("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(r4)); TEST_END(64)
    TEST_BEGIN(4) asm volatile(REP16("v_lshlrev_b64 %0, 5, %0\n v_lshlrev_b64 %1, 5, %1\n v_lshlrev_b64 %2, 5, %2\n v_lshlrev_b64 %3, 5, %3\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)); TEST_END(64)
    TEST_BEGIN(5) asm volatile(REP16("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n") : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(lane)); TEST_END(64)
    TEST_BEGIN(6) asm volatile(REP16("v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8\n") : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(lane)); TEST_END(64)
    // ---- LDS, independent, one waitcnt per 16
    TEST_BEGIN(10) asm volatile(REP4(REP4("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7\n") "s_waitcnt lgkmcnt(0)\n") : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory"); TEST_END(64)
    TEST_BEGIN(11) asm volatile(REP4(REP4("ds_write_b64 %4, %0\n ds_write_b64 %5, %1\n ds_write_b64 %6, %2\n ds_write_b64 %7, %3\n") "s_waitcnt lgkmcnt(0)\n") : : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory"); TEST_END(64)
    TEST_BEGIN(12) asm volatile(REP4(REP4("ds_read2_b64 %0, %4 offset1:40\n ds_read2_b64 %1, %5 offset1:40\n ds_read2_b64 %2, %6 offset1:40\n ds_read2_b64 %3, %7 offset1:40\n") "s_waitcnt lgkmcnt(0)\n") : "=v"(*(__uint128_t *)&r0), "=v"(*(__uint128_t *)&r2), "=v"(*(__uint128_t *)&r4), "=v"(*(__uint128_t *)&r6) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory"); TEST_END(64)
    TEST_BEGIN(13) asm volatile(REP4(REP4("ds_read_b128 %0, %4\n ds_read_b128 %1, %5\n ds_read_b128 %2, %6\n ds_read_b128 %3, %7\n") "s_waitcnt lgkmcnt(0)\n") : "=v"(*(__uint128_t *)&r0), "=v"(*(__uint128_t *)&r2), "=v"(*(__uint128_t *)&r4), "=v"(*(__uint128_t *)&r6) : "v"(a0 * 2), "v"(a1 * 2), "v"(a2 * 2), "v"(a3 * 2) : "memory"); TEST_END(64)
    TEST_BEGIN(14) asm volatile(REP4(REP4("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n") "s_waitcnt lgkmcnt(0)\n") : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory"); TEST_END(64)
    // ---- mixtures: 1 LDS read + k VALU
    TEST_BEGIN(20) asm volatile(REP16("ds_read_b64 %0, %4\n v_xor_b32 %2, %2, %5\n v_xor_b32 %3, %3, %5\n v_add_u32 %1, %1, %5\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(a0), "v"(lane) : "memory"); TEST_END(64)
    TEST_BEGIN(21) asm volatile(REP16("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %6\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r0), "=v"(r1), "+v"(w2), "+v"(w3) : "v"(a0), "v"(a1), "v"(lane) : "memory"); TEST_END(64)
    // ---- s_waitcnt that does not wait, s_nop
    TEST_BEGIN(30) asm volatile(REP64("s_waitcnt lgkmcnt(15)\n")); TEST_END(64)
    TEST_BEGIN(31) asm volatile(REP16("v_xor_b32 %0, %0, %2\n s_waitcnt lgkmcnt(15)\n v_xor_b32 %1, %1, %2\n s_waitcnt lgkmcnt(15)\n") : "+v"(w0), "+v"(w1) : "v"(lane)); TEST_END(64)
    TEST_BEGIN(32) asm volatile(REP64("s_nop 0\n")); TEST_END(64)
    // ---- dependent LDS round trip: read -> address of the next read (latency)
    TEST_BEGIN(40) asm volatile(REP64("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a0) : : "memory"); TEST_END(64)
    TEST_BEGIN(41) asm volatile(REP64("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)\n v_lshl_add_u64 %0, %0, 0, %0\n") : "+v"(r0) : "v"(a0) : "memory"); TEST_END(128)
    // ---- sub with carry pair (the mix's "a -= e")
    TEST_BEGIN(50) asm volatile(REP16("v_sub_co_u32 %0, vcc, %0, %4\n v_subb_co_u32 %1, vcc, %1, %5, vcc\n v_sub_co_u32 %2, vcc, %2, %4\n v_subb_co_u32 %3, vcc, %3, %5, vcc\n") : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(lane), "v"(lane) : "vcc"); TEST_END(64)
    if (lane == 0 && wave == 0 && blockIdx.x == 0) out[test] = (float)cyc;
    if (w0 + w1 + w2 + w3 + (unsigned)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7) + a0 == 0x12345) out[99] = 1.0f;
}
int main() {
    float *d; hipMalloc(&d, 400);
    hipMemset(d, 0, 400);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    struct T { int id; const char *name; };
    std::vector<T> tests = {{0, "v_add_u32 dependent"}, {1, "v_add_u32 4 chains"}, {2, "v_lshl_add_u64 dependent"}, {3, "v_lshl_add_u64 4 chains"}, {4, "v_lshlrev_b64 4 chains"},
                            {5, "v_xor_b32 4 chains"}, {6, "v_mad_u32_u24 / v_bfe_u32"}, {10, "ds_read_b64 indep (wait per 16)"}, {11, "ds_write_b64"}, {12, "ds_read2_b64 offset1:40"},
                            {13, "ds_read_b128"}, {14, "ds_read_b32"}, {20, "1 ds_read_b64 + 3 VALU"}, {21, "2 ds_read_b64 + 2 VALU"}, {30, "s_waitcnt (no wait)"},
                            {31, "v_xor + s_waitcnt alternating"}, {32, "s_nop 0"}, {40, "ds_read_b32 dependent round trip"}, {41, "ds_read_b64 + wait + dependent add64 (per instr of 2)"}, {50, "v_sub_co/v_subb pairs"}};
    for (int lanes : {40, 64}) {
        for (int waves : {2, 4}) {
            printf("---- %d lanes, %d waves per CU (cycles per instruction, s_memtime ticks)\n", lanes, waves);
            for (auto &t : tests) {
                hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 163840, 0, t.id, 200, lanes, d);
                hipDeviceSynchronize();
                float v[100];
                hipMemcpy(v, d, 400, hipMemcpyDeviceToHost);
                printf("  %-36s %6.2f\n", t.name, v[t.id]);
            }
        }
    }
    return 0;
}
