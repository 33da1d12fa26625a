// ISAAC-64 seeding kernels: seed_seg_kernel (three-run init, the default), seed_pc_kernel (producer / consumer waves
// with a ring of generator words), seed_isaac64_kernel (fused),
// seed_debug_kernel (raw outputs for the parity tests); the rare paths whose lens rejection loop runs past the hand-off record are
// re-derived by the consumer waves themselves at the end of a launch (seed_fixup_wave) — included by hr_api.hip only (one translation unit: the kernels and the C ABI that launches them).
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.h"
#include "isaac_core.h"

using namespace hr;


// 32-bit LDS address of a generic pointer into shared memory / load from such an address (isaac_round keeps the address of
// its next gather in a register across a scheduling fence)
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(const __attribute__((address_space(3))) void *)p; }
__device__ __forceinline__ u64 lds_load64(uint32_t a) { return *(const __attribute__((address_space(3))) u64 *)(size_t)a; }

// One generator per LDS bank column: mem[i][col], SEED_COLS columns per workgroup.
static const int SEED_COLS = 80;               // 80 x 2 KiB = 160 KiB = the whole LDS of a CU
static const int SEED_WAVES = 2, SEED_LANES = SEED_COLS / SEED_WAVES;   // 2 waves x 40 active lanes
// the priority governor's view of a seed kernel (device_scene.h GovDev): one wave per workgroup stamps the start and the end; the producer
// waves' priorities (bits 0-1 even groups, bits 2-3 odd groups) follow from the level in force when the workgroup starts
__device__ __forceinline__ uint32_t seed_gov_begin(const RenderParams &rp, bool stamp) {
    if (!rp.gov) return rp.pad[1];
    const int32_t lvl = __hip_atomic_load(&rp.gov->level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamp) {
        atomicMin(&rp.gov->t0[0][rp.gov_slot], (unsigned long long)__builtin_amdgcn_s_memrealtime());
        if (blockIdx.x == 0) rp.gov->lvl[0][rp.gov_slot] = (uint32_t)lvl;
    }
    return gov_producer_prio(lvl, rp.pad[1] & 3u);
}
__device__ __forceinline__ void seed_gov_end(const RenderParams &rp, bool stamp) {
    if (rp.gov && stamp) atomicMax(&rp.gov->t1[0][rp.gov_slot], (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
struct LdsMem {
    u64 *col;  // &mem[0][col]
    __device__ __forceinline__ u64 ld(int i) const { return col[i * SEED_COLS]; }
    __device__ __forceinline__ uint32_t off(int i) const { return lds_addr(col) + (uint32_t)i * (uint32_t)(SEED_COLS * 8); }
    __device__ __forceinline__ u64 ldo(uint32_t o) const { return lds_load64(o); }
    __device__ __forceinline__ void st(int i, u64 v) { col[i * SEED_COLS] = v; }
};
// the path's hand-off record in global memory (device_scene.h: [item][quad][lane][4]); four slots = one quad per store
// (the padding lanes of the last group write into spare items behind the last one: no predicate in the hot loop)
struct RecStore {
    float *rec;   // &recs[item * REC_ITEM_FLOATS + lane * 4]
    uint64_t lo_off;   // RenderParams::rec_lo_off (uniform): the records' twin for the draws' residuals (RecordTail<.., LO>)
    __device__ __forceinline__ RecStore(float *recs, uint64_t pid, uint64_t lo = 0) : rec(recs + (size_t)(pid >> 6) * REC_ITEM_FLOATS + (size_t)(pid & 63u) * 4u), lo_off(lo) {}
    __device__ __forceinline__ void st4(int slot, float a, float b, float c, float d) {
        f4 v; v.x = a; v.y = b; v.z = c; v.w = d;
        *reinterpret_cast<f4 *>(rec + (slot >> 2) * 256) = v;
    }
    __device__ __forceinline__ void st4lo(int slot, float a, float b, float c, float d) {
        f4 v; v.x = a; v.y = b; v.z = c; v.w = d;
        *reinterpret_cast<f4 *>(rec + lo_off + (slot >> 2) * 256) = v;
    }
};
static const uint32_t SEED_SPARE_ITEMS = 3;

// ---- fix-up of the paths whose lens rejection loop runs past the hand-off record ----------------------------------------------
// A consumer wave notes such paths (all LENS_FAST first attempts rejected: 4.6e-4 of the paths of a round-lens camera) in its own
// list and, when its share of the launch is done, re-derives them in its own LDS columns with the fused init + round, keeping
// the last ISAAC_TAIL raw outputs in a per-wave scratch window; the reference's rejection loop is replayed over that window and
// the record rewritten rebased to a = 0 (record_from_window).  All bookkeeping is per wave: no atomics, no extra kernel, nothing
// between the seed kernels on their stream.  A path that would need more than ISAAC_TAIL outputs (probability 4e-16), or a list
// that overflows, is counted in rng_overflow and reported by hr_synchronize as HR_ERR_RNG_WINDOW instead of producing a wrong image.
static const uint32_t SEED_OVF_MIN = 2048;                 // entries per consumer wave, at least; hr_api.hip sizes the lists per launch
                                                           // from the paths a wave seeds (RenderParams::ovf_cap)
static const uint32_t SEED_WIN_WORDS = ISAAC_TAIL * 40;   // u64 per consumer wave: [ISAAC_TAIL][40 lanes]
__device__ __forceinline__ uint32_t wave_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ void ovf_note(bool ov, uint64_t pid, uint32_t *list, uint32_t &count, uint32_t cap) {   // whole wave
    const unsigned long long m = __ballot(ov);
    if (!m) return;
    const uint32_t slot = count + wave_rank(m);
    if (ov && slot < cap) list[slot] = (uint32_t)pid;
    count += (uint32_t)__popcll(m);
}
struct GlobalWindow {
    u64 *col;   // &win[0][lane]
    __device__ __forceinline__ void put(int step, u64 v) { col[(255 - step) * 40] = v; }
    __device__ __forceinline__ u64 ld(int k) const { return __hip_atomic_load(col + k * 40, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
template <class Mem, bool LO = false>
__device__ __forceinline__ void seed_fixup_wave(const RenderParams &rp, int lens_shape, Mem m, uint32_t lane40, bool lane_on, const uint32_t *list, uint32_t count,
                                                u64 *win, float *recs, Counters *cnt) {
    if (count > rp.ovf_cap) {
        if (lane40 == 0 && lane_on) atomicAdd(&cnt->rng_overflow, (unsigned long long)(count - rp.ovf_cap));
        count = rp.ovf_cap;
    }
    const IsaacWarm warm = isaac_warm();
    // The list was written by OTHER lanes of this wave (ovf_note: the lane that seeded the path), the last entries only a group ago: their stores
    // must have reached L2 before the L1-bypassing loads below ask for them — a lane that read the entry of an earlier launch instead re-derived
    // a path that did not need it and left the one that did with its unfinished record (round 6: one path in ~10^9 with two outcomes, run to run).
    __builtin_amdgcn_s_waitcnt(0);
    for (uint32_t base = 0; base < count; base += 40u) {
        const bool valid = lane_on && base + lane40 < count;
        const uint32_t pid = __hip_atomic_load(list + (valid ? base + lane40 : base), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t item = pid >> 6, j = pid & 63u;
        const uint32_t tile = item / rp.num_k, k = item - tile * rp.num_k;
        uint32_t px, py, sub;
        tile_lane_pixel(rp, tile, j, px, py, sub);
        u64 s, t;
        path_seed_words(rp.width, rp.height, px, py, sub, s, t);
        if (lane_on) {
            GlobalWindow w{win + lane40};
            isaac_seed_round<ISAAC_TAIL>(m, warm, 8700304ULL, (u64)(rp.sampling_begin + k * rp.stride), s, t, w);
            __builtin_amdgcn_s_waitcnt(0);   // the window stores have left the wave before it is read back (L1-bypassing loads)
            if (valid) {
                RecStore rs(recs, pid, rp.rec_lo_off);
                if (!record_from_window<LO>(w, ISAAC_TAIL, lens_shape, rs)) atomicAdd(&cnt->rng_overflow, 1ULL);
            }
        }
    }
}

static const size_t SEED_LDS_BYTES = (size_t)256 * SEED_COLS * 8;  // 160 KiB: mem[256][80 columns] u64


// recs layout: [item = tile * num_k + k][8 quads][64 lanes][4] f32 (flat path index pid = item * 64 + lane).
// The LDS holds 80 generators, so a workgroup walks the flat path index (item * 64 + j) in strides of 80:
// its two waves (40 active lanes each) run concurrently on two SIMDs — the time of one seeding pass does not
// depend on the lane count (one wave issues at most one instruction every ~4-5 cycles), only on how many
// generator states fit in the CU's LDS.
__global__ __launch_bounds__(64 * SEED_WAVES) void seed_isaac64_kernel(RenderParams rp, int lens_shape, float *__restrict__ recs,
                                                                      uint32_t *__restrict__ ovf, u64 *__restrict__ win, Counters *cnt) {
    extern __shared__ __align__(16) unsigned char smem[];
    u64 *mem = reinterpret_cast<u64 *>(smem);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool lane_on = lane < (uint32_t)SEED_LANES;
    const uint32_t col = wave * SEED_LANES + (lane_on ? lane : 0u);
    uint32_t *ovf_list = ovf + (size_t)(blockIdx.x * 2u + wave) * rp.ovf_cap;
    uint32_t ovf_count = 0;
    // latency-bound waves next to the trace kernel's waves: win issue arbitration (priority is a launch parameter)
    switch (rp.pad[0]) {
        case 0: break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
    const IsaacWarm warm = isaac_warm();
    const uint64_t paths = (uint64_t)rp.tiles_x * rp.tiles_y * rp.num_k * 64u;
    for (uint64_t base = (uint64_t)blockIdx.x * SEED_COLS; base < paths; base += (uint64_t)gridDim.x * SEED_COLS) {
        const uint64_t pid = base + col;
        const bool in_range = pid < paths;
        const uint32_t item = (uint32_t)((in_range ? pid : paths - 1) >> 6), j = (uint32_t)((in_range ? pid : paths - 1) & 63u);
        uint32_t tile = item / rp.num_k, k = item - tile * rp.num_k;
        uint32_t px, py, sub;
        tile_lane_pixel(rp, tile, j, px, py, sub);
        bool valid = in_range && px < rp.width && py < rp.height;
        u64 s, t;
        path_seed_words(rp.width, rp.height, valid ? px : 0u, valid ? py : 0u, sub, s, t);
        LdsMem m{mem + col};
        RecStore rs(recs, pid);
        RecordTail<RecStore> lt(rs, lens_shape);
        if (lane_on) {
            isaac_seed_round<REC_DRAWS>(m, warm, 8700304ULL, (u64)(rp.sampling_begin + k * rp.stride), s, t, lt);
            lt.finish();
        }
        ovf_note(lane_on && valid && lt.overflow(), pid, ovf_list, ovf_count, rp.ovf_cap);
    }
    seed_fixup_wave(rp, lens_shape, LdsMem{mem + col}, lane_on ? lane : 0u, lane_on, ovf_list, ovf_count, win + (size_t)(blockIdx.x * 2u + wave) * SEED_WIN_WORDS, recs, cnt);
}

// ---- producer / consumer seeding (debug option seed_mode = 1) -----------------------------------------------------------------
// One workgroup per CU, four waves, a contiguous range of path groups (80 paths = one LDS fill) per workgroup.
// Waves 2,3 (producers) run the scratch-free init of the paths AHEAD in registers, 64 lanes = one chunk of 64 consecutive
// paths per pass, and scatter the states into a small ring of group buffers in global memory that belongs to this
// workgroup (written and re-read on the same CU within ~40 us: L2 / Infinity Cache traffic, not HBM); waves 0,1
// (consumers) fill their half of the LDS from the ring with straight global_load_lds copies and run the round on 40
// lanes each.  A generator state then sits in LDS only for fill + round (~2/3 of the fused kernel's residency, and LDS
// capacity is what bounds seeding), and the init runs on full waves.
// The ring traffic is what this costs (it slows the trace kernel next door), so the producers stop after SPLIT of the 32
// init blocks and ship those plus the 16 registers the sweep continues from; the consumer does blocks >= SPLIT itself
// while its fill is in flight (isaac_init_front / isaac_init_back) — no mix is computed twice.
// Group buffer, per half: [row = 0 .. 8*SPLIT)[40 columns] u64 = the LDS image of generator words 0 .. 8*SPLIT - 1, then the
// registers as [8 pairs][40 columns][2] u64 (16 bytes per lane, contiguous across the lanes: eight coalesced 16-byte loads for the consumer).  Two barriers per
// group: in iteration `it` the consumers work on group it-1 while the producers complete group it+1 (5 chunks per 4 groups) —
// one group of slack, so that the consumer can fetch the registers of group `it` while it runs the round of group it-1.
template <int SPLIT>   // init blocks (of 8 words) done by the producer; even
struct PcLayout {
    static const int SHIP_ROWS = 8 * SPLIT;
    static const int RING_ROWS = SHIP_ROWS + 16;
    static const size_t HALF_WORDS = (size_t)RING_ROWS * SEED_LANES;    // u64 per half in the ring
    static const size_t GROUP_WORDS = 2 * HALF_WORDS;
    static_assert((SHIP_ROWS * SEED_LANES * 8) % 1024 == 0, "fill copies 1 KiB per wave instruction");
};
static const int SEED_RING_GROUPS = 4;                                   // group g lives in buffer g & 3
static const size_t SEED_RING_WORDS_MAX = SEED_RING_GROUPS * PcLayout<32>::GROUP_WORDS;   // per workgroup, any SPLIT
static const size_t SEED_LDS_HALF_BYTES = (size_t)256 * SEED_LANES * 8;  // 80 KiB

// Ring stores: a lane owns one column, so its words i and i + 1 are a row (320 B) apart.  Lane pairs (l, l ^ 1) swap one
// word each so that the even lane stores row i and the odd lane row i + 1 as 16-byte pieces {column 2k, column 2k + 1}:
// one dwordx4 store instruction then writes whole rows.  The swap is two v_cndmask_b32_dpp per word pair
// (even lane: {own v0, partner's v0}; odd lane: {partner's v1, own v1}), written by hand on the registers the mix leaves its
// result in (two-instruction hazard distances counted from isaac_mix_gfx950.h): the compiler's form of the same thing took 34 issue slots per block, this 18.
// The caller masks lanes without a path off around the whole init (lanes 2m, 2m + 1 are always on or off together).
template <int HEAD>
struct RingState {
    u64 *pair;   // even lane: &row0[col]; odd lane: &row1[col - 1]
    __device__ __forceinline__ RingState(u64 *half_base, uint32_t column, uint32_t lane) {
        u64 *col = half_base + column;
        pair = (lane & 1u) ? col + SEED_LANES - 1 : col;
        regs = half_base + (size_t)PcLayout<HEAD>::SHIP_ROWS * SEED_LANES + (size_t)column * 2u;
    }
    // block i / 8 of the pass-2 sweep: words i .. i + 7 of this lane's column
    __device__ __forceinline__ void st8(int i, u64 A, u64 B, u64 C, u64 D, u64 E, u64 F, u64 G, u64 H) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        const u64 even = 0x5555555555555555ULL, odd = 0xAAAAAAAAAAAAAAAAULL;
        // D = vcc ? src1 : dpp(src0).  The DPP operands are read >= 2 instructions after the mix last wrote them (VALU -> DPP hazard):
        // B, D, F, H first (H's last write is four instructions before the end of the mix), G — written last — at the very end.
#define HR_SWZ(dst, own, other) "v_cndmask_b32_dpp " dst ", " other ", " own ", vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define HR_LO(v) "v"((uint32_t)(v))
#define HR_HI(v) "v"((uint32_t)((v) >> 32))
        uint32_t x0l, x0h, x1l, x1h, x2l, x2h, x3l, x3h, y0l, y0h, y1l, y1h, y2l, y2h, y3l, y3h;
        asm("s_mov_b64 vcc, %24\n\t"
            HR_SWZ("%0", "%8", "%10") HR_SWZ("%1", "%9", "%11")        // q0.x = even ? A : partner's B
            HR_SWZ("%2", "%12", "%14") HR_SWZ("%3", "%13", "%15")      // q1.x = even ? C : partner's D
            HR_SWZ("%4", "%16", "%18") HR_SWZ("%5", "%17", "%19")      // q2.x = even ? E : partner's F
            HR_SWZ("%6", "%20", "%22") HR_SWZ("%7", "%21", "%23")      // q3.x = even ? G : partner's H
            : "=&v"(x0l), "=&v"(x0h), "=&v"(x1l), "=&v"(x1h), "=&v"(x2l), "=&v"(x2h), "=&v"(x3l), "=&v"(x3h)
            : HR_LO(A), HR_HI(A), HR_LO(B), HR_HI(B), HR_LO(C), HR_HI(C), HR_LO(D), HR_HI(D), HR_LO(E), HR_HI(E), HR_LO(F), HR_HI(F), HR_LO(G), HR_HI(G), HR_LO(H), HR_HI(H), "s"(even)
            : "vcc");
        asm("s_mov_b64 vcc, %24\n\t"
            HR_SWZ("%0", "%10", "%8") HR_SWZ("%1", "%11", "%9")        // q0.y = odd ? B : partner's A
            HR_SWZ("%2", "%14", "%12") HR_SWZ("%3", "%15", "%13")      // q1.y = odd ? D : partner's C
            HR_SWZ("%4", "%18", "%16") HR_SWZ("%5", "%19", "%17")      // q2.y = odd ? F : partner's E
            HR_SWZ("%6", "%22", "%20") HR_SWZ("%7", "%23", "%21")      // q3.y = odd ? H : partner's G
            : "=&v"(y0l), "=&v"(y0h), "=&v"(y1l), "=&v"(y1h), "=&v"(y2l), "=&v"(y2h), "=&v"(y3l), "=&v"(y3h)
            : HR_LO(A), HR_HI(A), HR_LO(B), HR_HI(B), HR_LO(C), HR_HI(C), HR_LO(D), HR_HI(D), HR_LO(E), HR_HI(E), HR_LO(F), HR_HI(F), HR_LO(G), HR_HI(G), HR_LO(H), HR_HI(H), "s"(odd)
            : "vcc");
#undef HR_SWZ
#undef HR_LO
#undef HR_HI
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 q0 = {x0l, x0h, y0l, y0h}, q1 = {x1l, x1h, y1l, y1h}, q2 = {x2l, x2h, y2l, y2h}, q3 = {x3l, x3h, y3l, y3h};
        // sc1: written through, not kept in the XCD's L2 (measured against plain / nt / sc0 sc1 stores: the trace kernel next door
        // keeps more of its tree in L2).  Fixed at compile time: a wave-uniform switch cost ~3 scalar issue slots per store.
        u64x2 *dst = reinterpret_cast<u64x2 *>(pair + i * SEED_LANES);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\t"
                     "global_store_dwordx4 %0, %2, off offset:640 sc1\n\t"
                     "global_store_dwordx4 %0, %3, off offset:1280 sc1\n\t"
                     "global_store_dwordx4 %0, %4, off offset:1920 sc1\n\t"
                     "s_nop 1" : : "v"(dst), "v"(q0), "v"(q1), "v"(q2), "v"(q3) : "memory");   // s_nop: the data registers may be rewritten right after (VMEM-store-data hazard, invisible to the compiler here)
        static_assert(SEED_LANES * 8 * 2 == 640, "row pair stride of the ring");
    }
    // registers j, j + 1 (j even) of this lane's column: [j / 2][column][2] behind the shipped rows, no lane swap needed
    __device__ __forceinline__ void end2(int j, u64 v0, u64 v1) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        u64x2 q;
        q.x = v0; q.y = v1;
        *reinterpret_cast<u64x2 *>(regs + (j >> 1) * (2 * SEED_LANES)) = q;
    }
    u64 *regs;   // &half[SHIP_ROWS * 40 + column * 2]
};
struct LdsHalfMem {
    u64 *col;
    __device__ __forceinline__ u64 ld(int i) const { return col[i * SEED_LANES]; }
    __device__ __forceinline__ uint32_t off(int i) const { return lds_addr(col) + (uint32_t)i * (uint32_t)(SEED_LANES * 8); }
    __device__ __forceinline__ u64 ldo(uint32_t o) const { return lds_load64(o); }
    __device__ __forceinline__ void st(int i, u64 v) { col[i * SEED_LANES] = v; }
};
// PROF (option seed_prof): s_memtime stamps around the consumer's phases, summed per wave into Counters::seed_phase
//   0 issue of the register loads + bookkeeping   1 wait for the 16 registers   2 isaac_init_back   3 barrier B (fill landed)
//   4 isaac_round + record head   5 overflow note   6 barrier A (waiting for the producers / the other half)   7 groups
// The two roles are two separate loops that meet at the same two barriers per iteration: the register allocation of one role
// then does not carry the other's live values (the consumer's 32 prefetched registers through the producer's init, ...), and the
// kernel stays at <= 128 VGPRs — one more and the trace kernel next door loses a wave per SIMD.
struct PcRange {
    uint64_t paths, G0, G1, first_path, end_path;
    u64 *ring_wg;
};
template <int SEED_SPLIT, bool PROF>
__device__ __forceinline__ void seed_pc_consumer(const RenderParams &rp, int lens_shape, const PcRange &r, unsigned char *smem, uint32_t lane, uint32_t half,
                                                 float *__restrict__ recs, uint32_t *__restrict__ ovf, u64 *__restrict__ win, Counters *cnt) {
    typedef PcLayout<SEED_SPLIT> L;
    uint32_t *ovf_list = ovf + (size_t)(blockIdx.x * 2u + half) * rp.ovf_cap;
    uint32_t ovf_count = 0;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm = 0;
#define HR_STAMP(i) do { if (PROF) { unsigned long long now_ = __builtin_readcyclecounter(); pc[i] += now_ - tm; tm = now_; } } while (0)
    typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
    u64x2_t st16v[8];   // the 16 registers of the group whose init_back comes next (fetched one round ahead)
    const uint32_t colr = lane < (uint32_t)SEED_LANES ? lane : 0u;
    auto load_regs = [&](uint64_t g) {
        const u64 *regs = r.ring_wg + (g & (SEED_RING_GROUPS - 1)) * L::GROUP_WORDS + half * L::HALF_WORDS + (size_t)L::SHIP_ROWS * SEED_LANES + (size_t)colr * 2u;
#pragma unroll
        for (int q = 0; q < 8; q++) st16v[q] = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t *>(regs + q * (2 * SEED_LANES)));
    };
    unsigned char *lds_half = smem + (size_t)half * SEED_LDS_HALF_BYTES;
    LdsHalfMem m{reinterpret_cast<u64 *>(lds_half) + colr};
    const uint64_t n_groups = r.G1 - r.G0;
    __syncthreads();   // A of iteration 0: group G0 is complete in the ring
    for (uint64_t it = 1; it <= n_groups; it++) {
        // ---- group g enters the LDS: the producer wave of this half copies words < 8*SPLIT from the ring; this wave fetches the 16
        //      registers and finishes init blocks >= SPLIT straight into LDS meanwhile (no mix is computed twice).  Barrier B: complete.
        if (PROF) tm = __builtin_readcyclecounter();
        const uint64_t g = r.G0 + it - 1;
        if (it == 1) load_regs(g);   // later groups: fetched during the previous round
        const uint64_t pid = g * SEED_COLS + half * SEED_LANES + colr;
        const bool in_range = pid < r.paths;
        const uint32_t item = (uint32_t)((in_range ? pid : r.paths - 1) >> 6), j = (uint32_t)((in_range ? pid : r.paths - 1) & 63u);
        uint32_t tile = item / rp.num_k;
        uint32_t px, py, sub;
        tile_lane_pixel(rp, tile, j, px, py, sub);
        const bool valid = in_range && px < rp.width && py < rp.height;
        HR_STAMP(0);
        if (PROF) { __builtin_amdgcn_s_waitcnt(0x0F70); HR_STAMP(1); }   // vmcnt(0)
        u64 st16[16];
#pragma unroll
        for (int q = 0; q < 8; q++) { st16[2 * q] = st16v[q].x; st16[2 * q + 1] = st16v[q].y; }
        if (lane < (uint32_t)SEED_LANES) isaac_init_back<SEED_SPLIT>(m, st16);
        HR_STAMP(2);
        __syncthreads();   // B
        HR_STAMP(3);
        if (it < n_groups) load_regs(r.G0 + it);   // group G0 + it has been complete since barrier A; used after this round
        // ---- the round of group g
        RecStore rs(recs, pid);
        RecordTail<RecStore> lt(rs, lens_shape);
        if (lane < (uint32_t)SEED_LANES) {
            isaac_round<REC_DRAWS>(m, lt);
            lt.finish();
        }
        HR_STAMP(4);
        ovf_note(lane < (uint32_t)SEED_LANES && valid && lt.overflow(), pid, ovf_list, ovf_count, rp.ovf_cap);
        HR_STAMP(5);
        if (PROF) pc[7]++;
        __syncthreads();   // A: group G0 + it is complete in the ring; the LDS and buffer (G0 + it - 1) & 3 are free again
        HR_STAMP(6);
    }
#undef HR_STAMP
    if (PROF && lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&cnt->seed_phase[i], pc[i]);
    const bool lane_on = lane < (uint32_t)SEED_LANES;
    seed_fixup_wave(rp, lens_shape, m, colr, lane_on, ovf_list, ovf_count, win + (size_t)(blockIdx.x * 2u + half) * SEED_WIN_WORDS, recs, cnt);
}
template <int SEED_SPLIT>
__device__ __forceinline__ void seed_pc_producer(const RenderParams &rp, const PcRange &r, unsigned char *smem, uint32_t lane, uint32_t half) {
    typedef PcLayout<SEED_SPLIT> L;
    constexpr int CHUNKS = L::SHIP_ROWS * SEED_LANES * 8 / 1024;   // 1 KiB per wave-instruction
    static_assert(CHUNKS % 2 == 0, "fill is unrolled by two");
    const IsaacWarm warm = isaac_warm();
    uint64_t frontier = r.first_path & ~63ull;                   // first path not yet produced (chunk aligned)
    const uint64_t n_groups = r.G1 - r.G0;
    for (uint64_t it = 0; it <= n_groups; it++) {
        if (it > 0) {
            // the fill of group G0 + it - 1 (complete in the ring since the barrier that ended the last iteration): straight 1 KiB
            // global_load_lds copies, no VGPR round trip; ~60 cycles of issue each, which the consumer's critical path does not pay
            if (!(rp.pad[2] & 8u)) {
                const uint64_t g = r.G0 + it - 1;
                const u64 *src = r.ring_wg + (g & (SEED_RING_GROUPS - 1)) * L::GROUP_WORDS + half * L::HALF_WORDS;
                // generator words 0 .. 8*SPLIT - 1 of every column; the instruction offset advances both addresses, so one address pair
                // serves two 1 KiB copies
                const unsigned char *gsrc = reinterpret_cast<const unsigned char *>(src) + lane * 16u;
                unsigned char *ldst = smem + (size_t)half * SEED_LDS_HALF_BYTES;
#pragma unroll 5
                for (int q = 0; q < CHUNKS; q += 2, gsrc += 2048, ldst += 2048) {
                    const void __attribute__((address_space(1))) *gp = (const void __attribute__((address_space(1))) *)gsrc;
                    void __attribute__((address_space(3))) *lp = (void __attribute__((address_space(3))) *)ldst;
                    __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 2);      // aux 2 = nt: read once, do not keep
                    __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 2);
                }
                __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): landed
            }
            __syncthreads();   // B
        }
        // ---- complete group G0 + it + 1 in the ring (one group of slack: the consumer fetches the registers of group `it` while it
        //      runs the round of group it - 1)
        const uint64_t need = (r.G0 + it + 2) * SEED_COLS;      // paths below `need` (clipped to this workgroup's range) must be in the ring
        uint32_t n = 0;
        while (frontier < need && frontier < r.end_path) {
            if ((n & 1u) == half) {
                const uint64_t pid0 = frontier + lane;
                const bool on = pid0 >= r.first_path && pid0 < r.end_path && !(rp.pad[2] & 4u);   // pad[2]: timing experiments (debug_skip)
                const uint64_t ppid = pid0 >= r.first_path && pid0 < r.end_path ? pid0 : r.end_path - 1;
                const uint32_t item = (uint32_t)(ppid >> 6), j = (uint32_t)(ppid & 63u);
                uint32_t tile = item / rp.num_k, k = item - tile * rp.num_k;
                uint32_t px, py, sub;
                tile_lane_pixel(rp, tile, j, px, py, sub);
                bool pvalid = px < rp.width && py < rp.height;
                u64 s, t;
                path_seed_words(rp.width, rp.height, pvalid ? px : 0u, pvalid ? py : 0u, sub, s, t);
                const uint64_t g = ppid / SEED_COLS;
                const uint32_t c80 = (uint32_t)(ppid - g * SEED_COLS);
                RingState<SEED_SPLIT> out(r.ring_wg + (g & (SEED_RING_GROUPS - 1)) * L::GROUP_WORDS + (c80 >= (uint32_t)SEED_LANES ? L::HALF_WORDS : 0), c80 % SEED_LANES, lane);
                if (on) isaac_init_front<SEED_SPLIT>(out, warm, 8700304ULL, (u64)(rp.sampling_begin + k * rp.stride), s, t);
            }
            frontier += 64;
            n++;
        }
        __builtin_amdgcn_s_waitcnt(0);   // the ring stores are hand-written: the compiler does not wait for them at the barrier by itself
        __syncthreads();   // A
    }
}
template <int SEED_SPLIT, bool PROF = false>   // SPLIT: init blocks done by the producers
__global__ __launch_bounds__(256) void seed_pc_kernel(RenderParams rp, int lens_shape, u64 *__restrict__ ring, float *__restrict__ recs,
                                                      uint32_t *__restrict__ ovf, u64 *__restrict__ win, Counters *cnt) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, half = wave & 1u;
    const bool consumer = wave < 2u;
    const uint32_t pprio = consumer ? 0u : seed_gov_begin(rp, wave == 2u && lane == 0u);   // the producers' priorities (even | odd groups) at the governor's level
    const uint32_t prio = consumer ? rp.pad[0] : pprio & 3u;
    switch (prio) {  // s_setprio takes an immediate
        case 0: break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
    PcRange r;
    r.paths = (uint64_t)rp.tiles_x * rp.tiles_y * rp.num_k * 64u;
    const uint64_t groups = (r.paths + SEED_COLS - 1) / SEED_COLS;
    r.G0 = groups * blockIdx.x / gridDim.x; r.G1 = groups * (blockIdx.x + 1) / gridDim.x;   // this workgroup's groups
    r.first_path = r.G0 * SEED_COLS; r.end_path = r.G1 * SEED_COLS < r.paths ? r.G1 * SEED_COLS : r.paths;
    r.ring_wg = ring + (size_t)blockIdx.x * SEED_RING_WORDS_MAX;
    if (consumer) seed_pc_consumer<SEED_SPLIT, PROF>(rp, lens_shape, r, smem, lane, half, recs, ovf, win, cnt);
    else {
        seed_pc_producer<SEED_SPLIT>(rp, r, smem, lane, half);
        seed_gov_end(rp, wave == 2u && lane == 0u);
    }
}

// ---- three-run seeding (the default; debug option seed_mode = 2) ---------------------------------------------------------------------------------
// Same four waves, same LDS halves, same round — but NO generator words travel through memory.  The init sweep of a group is cut
// into three runs of SEG_NBLK blocks (device_scene.h) that are computed at the same time, in the window in which the half's LDS is
// free, by different LANES: the mix is the same instruction stream whatever block it works on, so one wave can advance 64
// (generator, run) pairs at once, and a consumer wave has 24 lanes to spare:
//     consumer wave of the half   lanes 0-39: generators 0-39, run 3 (blocks 21-31)     lanes 40-63: generators 0-23, run 2 (blocks 11-21)
//     producer wave of the half   lanes 0-39: generators 0-39, run 1 (blocks 0-10)      lanes 40-55: generators 24-39, run 2
// Each lane starts from the 16 registers the sweep holds at its first block.  Those come from the producer wave's AHEAD pass, which
// runs the same sweep two groups early without its stores (registers only: isaac_init_ahead) while the consumer is in its round, and
// leaves three 128-byte states per generator in a small ring (15 KiB per half and group instead of 45 KiB + an LDS-DMA fill).
// The window is 11 blocks long instead of 16, and the ring traffic that slows the trace kernel next door falls by two thirds.
struct SegLayout {
    static const size_t STATE_WORDS = (size_t)16 * SEED_LANES;   // one state of every column of a half: [8 pairs][40 columns][2] u64
    static const size_t HALF_WORDS = 3 * STATE_WORDS;
    static const size_t GROUP_WORDS = 2 * HALF_WORDS;
};
static_assert(SEED_RING_GROUPS * SegLayout::GROUP_WORDS <= SEED_RING_WORDS_MAX, "the ring allocation of the producer / consumer kernel holds it");
struct SegStateOut {
    u64 *base;   // &half[0][0][column][0]
    __device__ __forceinline__ void state(int k, u64 a, u64 b, u64 c, u64 d, u64 e, u64 f, u64 g, u64 h, u64 A, u64 B, u64 C, u64 D, u64 E, u64 F, u64 G, u64 H) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        u64x2 *dst = reinterpret_cast<u64x2 *>(base + (size_t)k * SegLayout::STATE_WORDS);
        const u64x2 q[8] = {{a, b}, {c, d}, {e, f}, {g, h}, {A, B}, {C, D}, {E, F}, {G, H}};
#pragma unroll
        // (the compiler does not know this is a store: a VALU write to the data registers in the next slot would race with the
        // store still reading them 16 lanes at a time — gfx9's VMEM-store-data hazard — hence the s_nop)
        for (int j = 0; j < 8; j++) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst + j * SEED_LANES), "v"(q[j]) : "memory");
    }
};
// what a lane does in the window: which generator (column of the half), which run
struct SegLane {
    uint32_t col, run;
    bool on;
};
__device__ __forceinline__ SegLane seg_lane(bool consumer, uint32_t lane) {
    SegLane l;
    if (lane < (uint32_t)SEED_LANES) { l.col = lane; l.run = consumer ? 2u : 0u; l.on = true; }
    else if (consumer) { l.col = lane - (uint32_t)SEED_LANES; l.run = 1u; l.on = true; }                 // 24 lanes: generators 0-23
    else { l.col = lane - 16u; l.run = 1u; l.on = lane < 56u; if (!l.on) l.col = (uint32_t)SEED_LANES - 1u; }   // 16 lanes: generators 24-39
    return l;
}
struct SegRegs {
    typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
    u64x2_t v[8];
    __device__ __forceinline__ void load(const u64 *ring_wg, uint64_t g, uint32_t half, const SegLane &l) {
        const u64 *regs = ring_wg + (g & (SEED_RING_GROUPS - 1)) * SegLayout::GROUP_WORDS + half * SegLayout::HALF_WORDS + (size_t)l.run * SegLayout::STATE_WORDS + (size_t)l.col * 2u;
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t *>(regs + q * (2 * SEED_LANES)));
    }
    __device__ __forceinline__ void run(unsigned char *lds_half, const SegLane &l) const {
        u64 st16[16];
#pragma unroll
        for (int q = 0; q < 8; q++) { st16[2 * q] = v[q].x; st16[2 * q + 1] = v[q].y; }
        const uint32_t first = l.run == 0u ? 0u : l.run == 1u ? (uint32_t)SEG_B1 : (uint32_t)SEG_B2;
        LdsHalfMem m{reinterpret_cast<u64 *>(lds_half) + l.col + (size_t)first * 8u * SEED_LANES};
        if (l.on) isaac_init_run<SEG_NBLK>(m, st16);
    }
};
template <bool PROF, bool LO>
__device__ __forceinline__ void seed_seg_consumer(const RenderParams &rp, int lens_shape, const PcRange &r, unsigned char *smem, uint32_t lane, uint32_t half,
                                                  float *__restrict__ recs, uint32_t *__restrict__ ovf, u64 *__restrict__ win, Counters *cnt) {
    uint32_t *ovf_list = ovf + (size_t)(blockIdx.x * 2u + half) * rp.ovf_cap;
    uint32_t ovf_count = 0;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm = 0;
#define HR_STAMP(i) do { if (PROF) { unsigned long long now_ = __builtin_readcyclecounter(); pc[i] += now_ - tm; tm = now_; } } while (0)
    const SegLane sl = seg_lane(true, lane);
    SegRegs regs;
    const uint32_t colr = lane < (uint32_t)SEED_LANES ? lane : 0u;
    unsigned char *lds_half = smem + (size_t)half * SEED_LDS_HALF_BYTES;
    LdsHalfMem m{reinterpret_cast<u64 *>(lds_half) + colr};
    const uint64_t n_groups = r.G1 - r.G0;
    __syncthreads();   // A of iteration 0: the states of groups G0 and G0 + 1 are in the ring
    for (uint64_t it = 1; it <= n_groups; it++) {
        if (PROF) tm = __builtin_readcyclecounter();
        const uint64_t g = r.G0 + it - 1;
        if (it == 1) regs.load(r.ring_wg, g, half, sl);   // later groups: fetched during the previous round
        const uint64_t pid = g * SEED_COLS + half * SEED_LANES + colr;
        const bool in_range = pid < r.paths;
        const uint32_t item = (uint32_t)((in_range ? pid : r.paths - 1) >> 6), j = (uint32_t)((in_range ? pid : r.paths - 1) & 63u);
        uint32_t tile = item / rp.num_k;
        uint32_t px, py, sub;
        tile_lane_pixel(rp, tile, j, px, py, sub);
        const bool valid = in_range && px < rp.width && py < rp.height;
        HR_STAMP(0);
        if (PROF) { __builtin_amdgcn_s_waitcnt(0x0F70); HR_STAMP(1); }   // vmcnt(0)
        regs.run(lds_half, sl);                          // the window: this wave's runs of the sweep, straight into LDS
        HR_STAMP(2);
        __syncthreads();   // B: all three runs of every column are in
        HR_STAMP(3);
        if (it < n_groups) regs.load(r.ring_wg, r.G0 + it, half, sl);
        RecStore rs(recs, pid, rp.rec_lo_off);
        RecordTail<RecStore, LO> lt(rs, lens_shape);
        if (lane < (uint32_t)SEED_LANES) {
            isaac_round<REC_DRAWS>(m, lt);
            lt.finish();
        }
        HR_STAMP(4);
        ovf_note(lane < (uint32_t)SEED_LANES && valid && lt.overflow(), pid, ovf_list, ovf_count, rp.ovf_cap);
        HR_STAMP(5);
        if (PROF) pc[7]++;
        __syncthreads();   // A: the LDS is free again; the states of group G0 + it + 1 are in the ring
        HR_STAMP(6);
    }
#undef HR_STAMP
    if (PROF && lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&cnt->seed_phase[i], pc[i]);
    const bool lane_on = lane < (uint32_t)SEED_LANES;
    seed_fixup_wave<LdsHalfMem, LO>(rp, lens_shape, m, colr, lane_on, ovf_list, ovf_count, win + (size_t)(blockIdx.x * 2u + half) * SEED_WIN_WORDS, recs, cnt);
}
__device__ __forceinline__ void seed_seg_producer(const RenderParams &rp, const PcRange &r, unsigned char *smem, uint32_t lane, uint32_t half, const uint32_t pprio) {
    const IsaacWarm warm = isaac_warm();
    const SegLane sl = seg_lane(false, lane);
    SegRegs regs;
    unsigned char *lds_half = smem + (size_t)half * SEED_LDS_HALF_BYTES;
    const uint64_t n_groups = r.G1 - r.G0;
    uint64_t frontier = r.first_path & ~63ull;               // first path whose states are not in the ring yet (chunks of 64 paths)
    // ahead pass: the states of every path below group `upto`, in chunks of 64 consecutive paths — all 64 lanes busy, whatever group /
    // half boundary a chunk straddles; the two producer waves take alternate chunks
    auto ahead = [&](uint64_t upto) {
        const uint64_t need = upto * SEED_COLS;
        for (; frontier < need && frontier < r.end_path; frontier += 64) {
            if (((frontier >> 6) & 1u) != half) continue;
            const uint64_t pid0 = frontier + lane;
            const bool on = pid0 >= r.first_path && pid0 < r.end_path && !(rp.pad[2] & 4u);   // pad[2]: timing experiments (debug_skip)
            const uint64_t ppid = pid0 >= r.first_path && pid0 < r.end_path ? pid0 : r.end_path - 1;
            const uint32_t item = (uint32_t)(ppid >> 6), j = (uint32_t)(ppid & 63u);
            uint32_t tile = item / rp.num_k, k = item - tile * rp.num_k;
            uint32_t px, py, sub;
            tile_lane_pixel(rp, tile, j, px, py, sub);
            bool pvalid = px < rp.width && py < rp.height;
            u64 s, t;
            path_seed_words(rp.width, rp.height, pvalid ? px : 0u, pvalid ? py : 0u, sub, s, t);
            const uint64_t g = ppid / SEED_COLS;
            const uint32_t c80 = (uint32_t)(ppid - g * SEED_COLS), hh = c80 >= (uint32_t)SEED_LANES ? 1u : 0u;
            SegStateOut out{r.ring_wg + (g & (SEED_RING_GROUPS - 1)) * SegLayout::GROUP_WORDS + hh * SegLayout::HALF_WORDS + (size_t)(c80 - hh * (uint32_t)SEED_LANES) * 2u};
            if (on) isaac_init_ahead<SEG_B1, SEG_B2>(out, warm, 8700304ULL, (u64)(rp.sampling_begin + k * rp.stride), s, t);
        }
        __builtin_amdgcn_s_waitcnt(0);   // the state stores are hand-written: the compiler does not wait for them at the barrier by itself
    };
    // iteration 0: the states of groups G0 and G0 + 1 (one group of slack, as in the producer / consumer kernel)
    ahead(r.G0 + 2);
    __syncthreads();   // A
    if (n_groups >= 1) regs.load(r.ring_wg, r.G0, half, sl);
    for (uint64_t it = 1; it <= n_groups; it++) {
        regs.run(lds_half, sl);          // the window
        __syncthreads();   // B
        if (it < n_groups) regs.load(r.ring_wg, r.G0 + it, half, sl);   // for the next window; complete in the ring since the last barrier A
        // the ahead pass's priority may alternate between groups (pprio bits 2-3: the priority of the odd groups): the priority
        // governor balances the two kernels with it, and the balance point usually lies between two whole levels
        if (((pprio >> 2) & 3u) != (pprio & 3u)) {
            const uint32_t pr = (it & 1u) ? (pprio >> 2) & 3u : pprio & 3u;
            if (pr == 0u) __builtin_amdgcn_s_setprio(0); else if (pr == 1u) __builtin_amdgcn_s_setprio(1); else if (pr == 2u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
        }
        ahead(r.G0 + it + 2);
        __syncthreads();   // A
    }
}
// LO: the records' twin with the draws' residuals is written too (precise shading, RenderParams::rec_lo_off)
template <bool PROF = false, bool LO = false>
__global__ __launch_bounds__(256) void seed_seg_kernel(RenderParams rp, int lens_shape, u64 *__restrict__ ring, float *__restrict__ recs,
                                                       uint32_t *__restrict__ ovf, u64 *__restrict__ win, Counters *cnt) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, half = wave & 1u;
    const bool consumer = wave < 2u;
    const uint32_t pprio = consumer ? 0u : seed_gov_begin(rp, wave == 2u && lane == 0u);   // the producers' priorities (even | odd groups) at the governor's level
    const uint32_t prio = consumer ? rp.pad[0] : pprio & 3u;
    switch (prio) {  // s_setprio takes an immediate
        case 0: break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
    PcRange r;
    r.paths = (uint64_t)rp.tiles_x * rp.tiles_y * rp.num_k * 64u;
    const uint64_t groups = (r.paths + SEED_COLS - 1) / SEED_COLS;
    r.G0 = groups * blockIdx.x / gridDim.x; r.G1 = groups * (blockIdx.x + 1) / gridDim.x;
    r.first_path = r.G0 * SEED_COLS; r.end_path = r.G1 * SEED_COLS < r.paths ? r.G1 * SEED_COLS : r.paths;
    r.ring_wg = ring + (size_t)blockIdx.x * SEED_RING_WORDS_MAX;
    if (consumer) seed_seg_consumer<PROF, LO>(rp, lens_shape, r, smem, lane, half, recs, ovf, win, cnt);
    else {
        seed_seg_producer(rp, r, smem, lane, half, pprio);
        seed_gov_end(rp, wave == 2u && lane == 0u);
    }
}

// The two kernels below (seed_mode 3 and 4) are measured experiments, both slower than the three-run kernel (profiles/NOTES.md): they are
// compiled only into builds made with `make EXPERIMENTS=1` (-DHR_EXPERIMENTS); the product library does not carry them.
#if defined(HR_EXPERIMENTS)
// ---- phase-shifted four-run seeding (debug option seed_mode = 3) ----------------------------------------------------------------------------------
// The three-run kernel's window is 11 blocks long because a half has 120 lanes to fill it with (its consumer wave and its producer
// wave).  Here the two halves run HALF A PERIOD APART: while one consumer is in the middle of its round the other half's window is
// served by that half's consumer AND BOTH producer waves — 192 lanes for 40 generators x 4 runs of 8 blocks (PS_NBLK; 160 lane-runs):
//     slot 0 = the half's consumer, slot 1 / 2 = the producer waves; lane-run L = 64 slot + lane < 160: generator L % 40, run L / 40
// The window is 8 blocks instead of 11.  s_barrier is workgroup-wide, so everybody meets at every barrier; per group there are four:
//     W0s  half 0 is free (consumer 0 has finished its round)      W0e  half 0's states are in -> consumer 0 starts its round
//     W1s  half 1 is free                                          W1e  half 1's states are in -> consumer 1 starts its round
// A consumer passes the OTHER half's two barriers in the middle of its round (PsRoundSync: after steps 112 and 156 — the round is
// 112 + 44 + 100 steps, its last 100 carry the record tail, and 44 steps take as long as the 8-block window), the producers run a
// window at each pair and one half of a PAUSED ahead pass in each of the two slots in between (isaac_ahead_part1 / part2: a whole
// ahead pass of a 64-path chunk is longer than a slot).  Consumer 1 idles half a period at the start, consumer 0 at the end; two
// trailing barriers let consumer 1 finish its last round.
static const int PS_NRUN = 4, PS_NBLK = 8, PS_P1BLK = 4;
struct PsLayout {
    static const size_t STATE_WORDS = (size_t)16 * SEED_LANES;
    static const size_t HALF_WORDS = PS_NRUN * STATE_WORDS;
    static const size_t GROUP_WORDS = 2 * HALF_WORDS;
};
static_assert(SEED_RING_GROUPS * PsLayout::GROUP_WORDS <= SEED_RING_WORDS_MAX, "the ring allocation of the producer / consumer kernel holds it");
struct PsStateOut {
    u64 *base;   // &half[0][0][column][0]
    __device__ __forceinline__ void state(int k, u64 a, u64 b, u64 c, u64 d, u64 e, u64 f, u64 g, u64 h, u64 A, u64 B, u64 C, u64 D, u64 E, u64 F, u64 G, u64 H) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        u64x2 *dst = reinterpret_cast<u64x2 *>(base + (size_t)k * PsLayout::STATE_WORDS);
        const u64x2 q[8] = {{a, b}, {c, d}, {e, f}, {g, h}, {A, B}, {C, D}, {E, F}, {G, H}};
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst + j * SEED_LANES), "v"(q[j]) : "memory");   // (see SegStateOut)
    }
};
struct PsLane { uint32_t col, run; bool on; };
__device__ __forceinline__ PsLane ps_lane(uint32_t slot, uint32_t lane) {
    const uint32_t L = slot * 64u + lane;
    PsLane l;
    l.on = L < (uint32_t)(PS_NRUN * SEED_LANES);
    const uint32_t Lc = l.on ? L : 0u;
    l.run = Lc / (uint32_t)SEED_LANES; l.col = Lc - l.run * (uint32_t)SEED_LANES;
    return l;
}
struct PsRegs {
    typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
    u64x2_t v[8];
    __device__ __forceinline__ void load(const u64 *ring_wg, uint64_t g, uint32_t half, const PsLane &l) {
        const u64 *regs = ring_wg + (g & (SEED_RING_GROUPS - 1)) * PsLayout::GROUP_WORDS + half * PsLayout::HALF_WORDS + (size_t)l.run * PsLayout::STATE_WORDS + (size_t)l.col * 2u;
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t *>(regs + q * (2 * SEED_LANES)));
    }
    __device__ __forceinline__ void run(unsigned char *lds_half, const PsLane &l) const {
        u64 st16[16];
#pragma unroll
        for (int q = 0; q < 8; q++) { st16[2 * q] = v[q].x; st16[2 * q + 1] = v[q].y; }
        LdsHalfMem m{reinterpret_cast<u64 *>(lds_half) + l.col + (size_t)(l.run * (uint32_t)PS_NBLK) * 8u * SEED_LANES};
        if (l.on) isaac_init_run<PS_NBLK>(m, st16);
    }
};
template <bool PROF>
struct PsRoundSync {
    static constexpr bool ON = true;
    static constexpr int N1 = 104, K1 = 2, N2 = 148, K2 = 2;   // after steps 111 and 155
    unsigned long long *pc, *tm; mutable int k;
    __device__ __forceinline__ void mid() const {   // a pure rendezvous: nothing of this wave's is handed over
        if (PROF) { unsigned long long now = __builtin_readcyclecounter(); pc[3 + 2 * k] += now - *tm; *tm = now; }
        __builtin_amdgcn_s_barrier();
        if (PROF) { unsigned long long now = __builtin_readcyclecounter(); pc[4 + 2 * k] += now - *tm; *tm = now; k++; }
    }
};
// PROF (option seed_prof = 1 | 2 | 3: consumer 0, consumer 1, producer 0): cycle stamps around every barrier of the chosen wave, summed into
// Counters::seed_phase — consumer: wait Ws, window, wait We, round part 1, wait, part 2, wait, part 3 + records; producer: wait W0s, window 0,
// wait W0e, slot A, wait W1s, window 1, wait W1e, slot B
template <int PROF>
__device__ __forceinline__ void seed_ps_consumer(const RenderParams &rp, int lens_shape, const PcRange &r, unsigned char *smem, uint32_t lane, uint32_t half,
                                                 float *__restrict__ recs, uint32_t *__restrict__ ovf, u64 *__restrict__ win, Counters *cnt) {
    uint32_t *ovf_list = ovf + (size_t)(blockIdx.x * 2u + half) * rp.ovf_cap;
    uint32_t ovf_count = 0;
    const PsLane sl = ps_lane(0u, lane);
    PsRegs regs;
    constexpr bool P = PROF != 0;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm = 0;
#define HR_STAMP(i) do { if (P) { unsigned long long now_ = __builtin_readcyclecounter(); pc[i] += now_ - tm; tm = now_; } } while (0)
    const uint32_t colr = lane < (uint32_t)SEED_LANES ? lane : 0u;
    unsigned char *lds_half = smem + (size_t)half * SEED_LDS_HALF_BYTES;
    LdsHalfMem m{reinterpret_cast<u64 *>(lds_half) + colr};
    const uint64_t n_groups = r.G1 - r.G0;
    __syncthreads();   // the states of groups G0 and G0 + 1 are in the ring
    if (n_groups >= 1) regs.load(r.ring_wg, r.G0, half, sl);
    if (half == 1u && n_groups >= 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }   // W0s(1), W0e(1): half a period behind
    for (uint64_t it = 1; it <= n_groups; it++) {
        const uint64_t g = r.G0 + it - 1;
        const uint64_t pid = g * SEED_COLS + half * SEED_LANES + colr;
        const bool in_range = pid < r.paths;
        const uint32_t item = (uint32_t)((in_range ? pid : r.paths - 1) >> 6), j = (uint32_t)((in_range ? pid : r.paths - 1) & 63u);
        uint32_t tile = item / rp.num_k;
        uint32_t px, py, sub;
        tile_lane_pixel(rp, tile, j, px, py, sub);
        const bool valid = in_range && px < rp.width && py < rp.height;
        if (P) tm = __builtin_readcyclecounter();
        __syncthreads();   // Ws: this half's LDS is free (the round before has read its last word)
        HR_STAMP(0);
        regs.run(lds_half, sl);                          // the window: this wave's runs of the sweep, straight into LDS
        HR_STAMP(1);
        __syncthreads();   // We: all four runs of every column are in
        HR_STAMP(2);
        if (it < n_groups) regs.load(r.ring_wg, r.G0 + it, half, sl);
        RecStore rs(recs, pid);
        RecordTail<RecStore> lt(rs, lens_shape);
        if (lane < (uint32_t)SEED_LANES) {
            isaac_round_deep28(m, lt, PsRoundSync<P>{pc, &tm, 0});   // passes the other half's Ws and We on its way
            lt.finish();
        }   // (s_barrier is a scalar instruction: the wave passes the two rendezvous once, whatever its lane mask is in there)
        ovf_note(lane < (uint32_t)SEED_LANES && valid && lt.overflow(), pid, ovf_list, ovf_count, rp.ovf_cap);
        HR_STAMP(7);
    }
    if (half == 0u && n_groups >= 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }   // W0s(n + 1), W0e(n + 1): consumer 1's last round passes them
#undef HR_STAMP
    if (P && lane == 0 && half + 1u == (uint32_t)PROF)
        for (int i = 0; i < 8; i++) atomicAdd(&cnt->seed_phase[i], pc[i]);
    const bool lane_on = lane < (uint32_t)SEED_LANES;
    seed_fixup_wave(rp, lens_shape, m, colr, lane_on, ovf_list, ovf_count, win + (size_t)(blockIdx.x * 2u + half) * SEED_WIN_WORDS, recs, cnt);
}
template <int PROF>
__device__ __forceinline__ void seed_ps_producer(const RenderParams &rp, const PcRange &r, unsigned char *smem, uint32_t lane, uint32_t half, Counters *cnt) {
    constexpr bool P = PROF == 3;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm = 0;
#define HR_STAMP(i) do { if (P) { unsigned long long now_ = __builtin_readcyclecounter(); pc[i] += now_ - tm; tm = now_; } } while (0)
    const IsaacWarm warm = isaac_warm();
    const PsLane sl = ps_lane(1u + half, lane);
    PsRegs regs;
    const uint64_t n_groups = r.G1 - r.G0;
    uint64_t frontier = r.first_path & ~63ull;               // first path whose states are not in the ring yet (chunks of 64 paths)
    // one 64-path chunk of the ahead pass, in two parts (this wave takes every other chunk)
    AheadRegs job;
    PsStateOut job_out{nullptr};
    bool job_on = false, job_open = false;
    auto part1 = [&](uint64_t upto) -> bool {   // start the next chunk of this wave below group `upto`, if there is one
        const uint64_t need = upto * SEED_COLS;
        for (; frontier < need && frontier < r.end_path; frontier += 64) {
            if (((frontier >> 6) & 1u) != half) continue;
            const uint64_t pid0 = frontier + lane;
            job_on = pid0 >= r.first_path && pid0 < r.end_path && !(rp.pad[2] & 4u);   // pad[2]: timing experiments (debug_skip)
            const uint64_t ppid = pid0 >= r.first_path && pid0 < r.end_path ? pid0 : r.end_path - 1;
            const uint32_t item = (uint32_t)(ppid >> 6), j = (uint32_t)(ppid & 63u);
            uint32_t tile = item / rp.num_k, k = item - tile * rp.num_k;
            uint32_t px, py, sub;
            tile_lane_pixel(rp, tile, j, px, py, sub);
            bool pvalid = px < rp.width && py < rp.height;
            u64 s, t;
            path_seed_words(rp.width, rp.height, pvalid ? px : 0u, pvalid ? py : 0u, sub, s, t);
            const uint64_t g = ppid / SEED_COLS;
            const uint32_t c80 = (uint32_t)(ppid - g * SEED_COLS), hh = c80 >= (uint32_t)SEED_LANES ? 1u : 0u;
            job_out.base = r.ring_wg + (g & (SEED_RING_GROUPS - 1)) * PsLayout::GROUP_WORDS + hh * PsLayout::HALF_WORDS + (size_t)(c80 - hh * (uint32_t)SEED_LANES) * 2u;
            if (job_on) isaac_ahead_part1<PS_P1BLK>(job_out, warm, 8700304ULL, (u64)(rp.sampling_begin + k * rp.stride), s, t, job);
            frontier += 64;
            return true;
        }
        return false;
    };
    auto part2 = [&]() {
        if (job_on) isaac_ahead_part2<PS_NRUN, PS_NBLK, PS_P1BLK>(job_out, job);
        __builtin_amdgcn_s_waitcnt(0);   // the state stores are hand-written: the compiler does not wait for them at the barrier by itself
    };
    // before the first window: the states of groups G0 and G0 + 1, without pausing
    while (part1(r.G0 + 2)) part2();
    __syncthreads();
    if (P) tm = __builtin_readcyclecounter();
    for (uint64_t it = 1; it <= n_groups; it++) {
        const uint64_t g = r.G0 + it - 1;
        for (uint32_t hw = 0; hw < 2u; hw++) {
            regs.load(r.ring_wg, g, hw, sl);                 // complete in the ring for more than a group
            HR_STAMP(hw ? 3 : 7);
            __syncthreads();   // Ws(hw)
            HR_STAMP(hw ? 4 : 0);
            regs.run(smem + (size_t)hw * SEED_LDS_HALF_BYTES, sl);   // the window of half hw
            HR_STAMP(hw ? 5 : 1);
            __syncthreads();   // We(hw)
            HR_STAMP(hw ? 6 : 2);
            // the slot behind the window: one half of a chunk of the ahead pass.  By the end of this group the states of group
            // it + 2 have to be complete; a wave's chunks are 128 paths apart and the target moves by 80 paths a group, so one
            // chunk per group and wave is enough
            if (hw == 0u) job_open = part1(r.G0 + it + 2);
            else if (job_open) { part2(); job_open = false; }
        }
    }
    if (n_groups >= 1) { __syncthreads(); __syncthreads(); }   // W0s(n + 1), W0e(n + 1)
#undef HR_STAMP
    if (P && lane == 0 && half == 0u)
        for (int i = 0; i < 8; i++) atomicAdd(&cnt->seed_phase[i], pc[i]);
}
template <int PROF = 0>
__global__ __launch_bounds__(256) void seed_ps_kernel(RenderParams rp, int lens_shape, u64 *__restrict__ ring, float *__restrict__ recs,
                                                      uint32_t *__restrict__ ovf, u64 *__restrict__ win, Counters *cnt) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, half = wave & 1u;
    const bool consumer = wave < 2u;
    const uint32_t pprio = consumer ? 0u : seed_gov_begin(rp, wave == 2u && lane == 0u);   // the producers' priorities (even | odd groups) at the governor's level
    const uint32_t prio = consumer ? rp.pad[0] : pprio & 3u;
    switch (prio) {  // s_setprio takes an immediate
        case 0: break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
    PcRange r;
    r.paths = (uint64_t)rp.tiles_x * rp.tiles_y * rp.num_k * 64u;
    const uint64_t groups = (r.paths + SEED_COLS - 1) / SEED_COLS;
    r.G0 = groups * blockIdx.x / gridDim.x; r.G1 = groups * (blockIdx.x + 1) / gridDim.x;
    r.first_path = r.G0 * SEED_COLS; r.end_path = r.G1 * SEED_COLS < r.paths ? r.G1 * SEED_COLS : r.paths;
    r.ring_wg = ring + (size_t)blockIdx.x * SEED_RING_WORDS_MAX;
    if (consumer) seed_ps_consumer<PROF>(rp, lens_shape, r, smem, lane, half, recs, ovf, win, cnt);
    else {
        seed_ps_producer<PROF>(rp, r, smem, lane, half, cnt);
        seed_gov_end(rp, wave == 2u && lane == 0u);
    }
}

// ---- five-wave four-run seeding (debug option seed_mode = 4) ---------------------------------------------------------------------------------------
// The three-run kernel's window is 11 blocks because a half has 120 lanes to fill it with.  With a FIFTH wave in the workgroup there are
// 160 per half — consumer 64 + 96 of the three producer waves' 192 — which is 40 generators x 4 runs of 8 blocks (PS_NBLK): the window is
// 8 blocks, both halves still run it at the same time (no window beside a round, which is what sank the phase-shifted kernel), the
// round is the three-run kernel's own code, and there are the same two barriers per group.  What it costs is paid by the trace kernel,
// which has the time since round 3: five waves on four SIMDs put two of them on one SIMD, and a SIMD with two of this kernel's 128-
// register waves holds two trace waves instead of four.  The two waves that share a SIMD must not be consumers (a consumer's round is
// the critical path): the roles are handed out at run time from the SIMD each wave finds itself on (HW_ID), through LDS words that the
// generators overwrite later.
//     window of half h:  slot 0 = its consumer (64 lanes), slot 1 = producer h (64 lanes), slot 2 = lanes 32 h .. 32 h + 31 of producer 2;
//                        lane-run L = 64 slot + lane (slot 2: 128 + lane - 32 h) < 160: generator L % 40, run L / 40
//     ahead pass:        64-path chunks in the order solo, pair A, solo, pair B (solo = the producer with a SIMD to itself): the two
//                        producers that share a SIMD never have a chunk in the same group
struct W5Roles {
    uint32_t consumer_half;   // 0 | 1 for the two consumers, 2 = producer
    uint32_t prod;            // producers: 0, 1, 2 (the window's slots); 2 is the solo producer when there is one
    uint32_t solo_ok;         // a producer has a SIMD to itself (else the chunks go round-robin)
};
__device__ __forceinline__ W5Roles w5_roles(unsigned char *smem, uint32_t wave, uint32_t lane) {
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    volatile uint32_t *box = reinterpret_cast<volatile uint32_t *>(smem);
    if (lane == 0) box[wave] = (hwid >> 4) & 3u;   // SIMD_ID
    __syncthreads();
    uint32_t simd[5], cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < 5; i++) { simd[i] = box[i]; cnt[simd[i]]++; }
    __syncthreads();                                // (the words are generator state from the first window on)
    // consumers: the first two waves that have their SIMD to themselves (any two, if the placement is an unusual one)
    int c0 = -1, c1 = -1;
    for (int i = 0; i < 5; i++)
        if (cnt[simd[i]] == 1) { if (c0 < 0) c0 = i; else if (c1 < 0) c1 = i; }
    if (c1 < 0) { c0 = 0; c1 = 1; }
    // producers in wave order; the one with a SIMD to itself (if any) takes slot 2
    int pr[3], np = 0, solo = -1;
    for (int i = 0; i < 5; i++)
        if (i != c0 && i != c1) { pr[np] = i; if (cnt[simd[i]] == 1) solo = np; np++; }
    if (solo >= 0 && solo != 2) { const int t = pr[2]; pr[2] = pr[solo]; pr[solo] = t; }
    W5Roles r;
    r.consumer_half = (int)wave == c0 ? 0u : (int)wave == c1 ? 1u : 2u;
    r.prod = (int)wave == pr[0] ? 0u : (int)wave == pr[1] ? 1u : 2u;
    r.solo_ok = solo >= 0 ? 1u : 0u;
    return r;
}
struct W5Lane { uint32_t half, col, run; bool on; };
__device__ __forceinline__ W5Lane w5_lane(uint32_t slot, uint32_t half_of_wave, uint32_t lane) {   // slot 0 consumer, 1 producer 0 | 1, 2 producer 2
    W5Lane l;
    l.half = slot == 2u ? lane >> 5 : half_of_wave;
    const uint32_t L = slot == 2u ? 128u + (lane & 31u) : slot * 64u + lane;
    l.on = L < (uint32_t)(PS_NRUN * SEED_LANES);
    l.run = L / (uint32_t)SEED_LANES; l.col = L - l.run * (uint32_t)SEED_LANES;
    return l;
}
struct W5Regs {
    typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
    u64x2_t v[8];
    __device__ __forceinline__ void load(const u64 *ring_wg, uint64_t g, const W5Lane &l) {
        const u64 *regs = ring_wg + (g & (SEED_RING_GROUPS - 1)) * PsLayout::GROUP_WORDS + l.half * PsLayout::HALF_WORDS + (size_t)l.run * PsLayout::STATE_WORDS + (size_t)l.col * 2u;
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t *>(regs + q * (2 * SEED_LANES)));
    }
    __device__ __forceinline__ void run(unsigned char *smem, const W5Lane &l) const {
        u64 st16[16];
#pragma unroll
        for (int q = 0; q < 8; q++) { st16[2 * q] = v[q].x; st16[2 * q + 1] = v[q].y; }
        LdsHalfMem m{reinterpret_cast<u64 *>(smem + (size_t)l.half * SEED_LDS_HALF_BYTES) + l.col + (size_t)(l.run * (uint32_t)PS_NBLK) * 8u * SEED_LANES};
        if (l.on) isaac_init_run<PS_NBLK>(m, st16);
    }
};
template <bool PROF>
__device__ __forceinline__ void seed_w5_consumer(const RenderParams &rp, int lens_shape, const PcRange &r, unsigned char *smem, uint32_t lane, uint32_t half,
                                                 float *__restrict__ recs, uint32_t *__restrict__ ovf, u64 *__restrict__ win, Counters *cnt) {
    uint32_t *ovf_list = ovf + (size_t)(blockIdx.x * 2u + half) * rp.ovf_cap;
    uint32_t ovf_count = 0;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm = 0;
#define HR_STAMP(i) do { if (PROF) { unsigned long long now_ = __builtin_readcyclecounter(); pc[i] += now_ - tm; tm = now_; } } while (0)
    const W5Lane sl = w5_lane(0u, half, lane);
    W5Regs regs;
    const uint32_t colr = lane < (uint32_t)SEED_LANES ? lane : 0u;
    unsigned char *lds_half = smem + (size_t)half * SEED_LDS_HALF_BYTES;
    LdsHalfMem m{reinterpret_cast<u64 *>(lds_half) + colr};
    const uint64_t n_groups = r.G1 - r.G0;
    __syncthreads();   // A of iteration 0: the states of groups G0 and G0 + 1 are in the ring
    for (uint64_t it = 1; it <= n_groups; it++) {
        if (PROF) tm = __builtin_readcyclecounter();
        const uint64_t g = r.G0 + it - 1;
        if (it == 1) regs.load(r.ring_wg, g, sl);   // later groups: fetched during the previous round
        const uint64_t pid = g * SEED_COLS + half * SEED_LANES + colr;
        const bool in_range = pid < r.paths;
        const uint32_t item = (uint32_t)((in_range ? pid : r.paths - 1) >> 6), j = (uint32_t)((in_range ? pid : r.paths - 1) & 63u);
        uint32_t tile = item / rp.num_k;
        uint32_t px, py, sub;
        tile_lane_pixel(rp, tile, j, px, py, sub);
        const bool valid = in_range && px < rp.width && py < rp.height;
        HR_STAMP(0);
        if (PROF) { __builtin_amdgcn_s_waitcnt(0x0F70); HR_STAMP(1); }   // vmcnt(0)
        regs.run(smem, sl);                              // the window: this wave's runs of the sweep, straight into LDS
        HR_STAMP(2);
        __syncthreads();   // B: all four runs of every column are in
        HR_STAMP(3);
        if (it < n_groups) regs.load(r.ring_wg, r.G0 + it, sl);
        RecStore rs(recs, pid);
        RecordTail<RecStore> lt(rs, lens_shape);
        if (lane < (uint32_t)SEED_LANES) {
            isaac_round<REC_DRAWS>(m, lt);
            lt.finish();
        }
        HR_STAMP(4);
        ovf_note(lane < (uint32_t)SEED_LANES && valid && lt.overflow(), pid, ovf_list, ovf_count, rp.ovf_cap);
        HR_STAMP(5);
        if (PROF) pc[7]++;
        __syncthreads();   // A: the LDS is free again; the states of group G0 + it + 1 are in the ring
        HR_STAMP(6);
    }
#undef HR_STAMP
    if (PROF && lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&cnt->seed_phase[i], pc[i]);
    const bool lane_on = lane < (uint32_t)SEED_LANES;
    seed_fixup_wave(rp, lens_shape, m, colr, lane_on, ovf_list, ovf_count, win + (size_t)(blockIdx.x * 2u + half) * SEED_WIN_WORDS, recs, cnt);
}
__device__ __forceinline__ void seed_w5_producer(const RenderParams &rp, const PcRange &r, unsigned char *smem, uint32_t lane, const W5Roles &role, const uint32_t pprio) {
    const IsaacWarm warm = isaac_warm();
    const W5Lane sl = w5_lane(role.prod == 2u ? 2u : 1u, role.prod & 1u, lane);
    W5Regs regs;
    const uint64_t n_groups = r.G1 - r.G0;
    uint64_t frontier = r.first_path & ~63ull;               // first path whose states are not in the ring yet (chunks of 64 paths)
    // whose chunk is it?  solo, pair A, solo, pair B (producer 2 is the solo one); without a solo producer: round-robin
    auto mine = [&](uint64_t chunk) -> bool {
        if (role.solo_ok) { const uint32_t ph = (uint32_t)(chunk & 3u); return ph == 1u ? role.prod == 0u : ph == 3u ? role.prod == 1u : role.prod == 2u; }
        return (uint32_t)(chunk % 3u) == role.prod;
    };
    auto ahead = [&](uint64_t upto) {
        const uint64_t need = upto * SEED_COLS;
        for (; frontier < need && frontier < r.end_path; frontier += 64) {
            if (!mine(frontier >> 6)) continue;
            const uint64_t pid0 = frontier + lane;
            const bool on = pid0 >= r.first_path && pid0 < r.end_path && !(rp.pad[2] & 4u);   // pad[2]: timing experiments (debug_skip)
            const uint64_t ppid = pid0 >= r.first_path && pid0 < r.end_path ? pid0 : r.end_path - 1;
            const uint32_t item = (uint32_t)(ppid >> 6), j = (uint32_t)(ppid & 63u);
            uint32_t tile = item / rp.num_k, k = item - tile * rp.num_k;
            uint32_t px, py, sub;
            tile_lane_pixel(rp, tile, j, px, py, sub);
            bool pvalid = px < rp.width && py < rp.height;
            u64 s, t;
            path_seed_words(rp.width, rp.height, pvalid ? px : 0u, pvalid ? py : 0u, sub, s, t);
            const uint64_t g = ppid / SEED_COLS;
            const uint32_t c80 = (uint32_t)(ppid - g * SEED_COLS), hh = c80 >= (uint32_t)SEED_LANES ? 1u : 0u;
            PsStateOut out{r.ring_wg + (g & (SEED_RING_GROUPS - 1)) * PsLayout::GROUP_WORDS + hh * PsLayout::HALF_WORDS + (size_t)(c80 - hh * (uint32_t)SEED_LANES) * 2u};
            if (on) {
                AheadRegs job;
                isaac_ahead_part1<PS_P1BLK>(out, warm, 8700304ULL, (u64)(rp.sampling_begin + k * rp.stride), s, t, job);
                isaac_ahead_part2<PS_NRUN, PS_NBLK, PS_P1BLK>(out, job);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);   // the state stores are hand-written: the compiler does not wait for them at the barrier by itself
    };
    ahead(r.G0 + 2);
    __syncthreads();   // A
    if (n_groups >= 1) regs.load(r.ring_wg, r.G0, sl);
    for (uint64_t it = 1; it <= n_groups; it++) {
        regs.run(smem, sl);          // the window
        __syncthreads();   // B
        if (it < n_groups) regs.load(r.ring_wg, r.G0 + it, sl);   // for the next window; complete in the ring since the last barrier A
        if (((pprio >> 2) & 3u) != (pprio & 3u)) {
            const uint32_t pr = (it & 1u) ? (pprio >> 2) & 3u : pprio & 3u;
            if (pr == 0u) __builtin_amdgcn_s_setprio(0); else if (pr == 1u) __builtin_amdgcn_s_setprio(1); else if (pr == 2u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
        }
        ahead(r.G0 + it + 2);
        __syncthreads();   // A
    }
}
template <bool PROF = false>
__global__ __launch_bounds__(320) void seed_w5_kernel(RenderParams rp, int lens_shape, u64 *__restrict__ ring, float *__restrict__ recs,
                                                      uint32_t *__restrict__ ovf, u64 *__restrict__ win, Counters *cnt) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const W5Roles role = w5_roles(smem, wave, lane);
    const bool consumer = role.consumer_half < 2u;
    const uint32_t pprio = consumer ? 0u : seed_gov_begin(rp, role.prod == 0u && lane == 0u);
    const uint32_t prio = consumer ? rp.pad[0] : pprio & 3u;
    switch (prio) {  // s_setprio takes an immediate
        case 0: break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
    PcRange r;
    r.paths = (uint64_t)rp.tiles_x * rp.tiles_y * rp.num_k * 64u;
    const uint64_t groups = (r.paths + SEED_COLS - 1) / SEED_COLS;
    r.G0 = groups * blockIdx.x / gridDim.x; r.G1 = groups * (blockIdx.x + 1) / gridDim.x;
    r.first_path = r.G0 * SEED_COLS; r.end_path = r.G1 * SEED_COLS < r.paths ? r.G1 * SEED_COLS : r.paths;
    r.ring_wg = ring + (size_t)blockIdx.x * SEED_RING_WORDS_MAX;
    if (consumer) seed_w5_consumer<PROF>(rp, lens_shape, r, smem, lane, role.consumer_half, recs, ovf, win, cnt);
    else {
        seed_w5_producer(rp, r, smem, lane, role, pprio);
        seed_gov_end(rp, role.prod == 0u && lane == 0u);
    }
}

#endif   // HR_EXPERIMENTS

// raw generator outputs for the parity tests: out[p * window + k] = k-th next_u64 of pixel-major path p
struct RawTail {
    u64 *out; int window;
    __device__ __forceinline__ void put(int step, u64 v) { int k = 255 - step; if (k < window) out[k] = v; }
};
struct LdsMem64 {
    u64 *col;
    __device__ __forceinline__ u64 ld(int i) const { return col[i * 64]; }
    __device__ __forceinline__ uint32_t off(int i) const { return lds_addr(col) + (uint32_t)i * (uint32_t)(64 * 8); }
    __device__ __forceinline__ u64 ldo(uint32_t o) const { return lds_load64(o); }
    __device__ __forceinline__ void st(int i, u64 v) { col[i * 64] = v; }
};
__global__ __launch_bounds__(64) void seed_debug_kernel(uint32_t W, uint32_t H, uint32_t sampling, uint32_t first_path, uint32_t num_paths,
                                                        int window, u64 *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    u64 *mem = reinterpret_cast<u64 *>(smem);
    const uint32_t lane = threadIdx.x;
    const IsaacWarm warm = isaac_warm();
    uint32_t idx = blockIdx.x * 64 + lane;
    bool valid = idx < num_paths;
    uint32_t p = first_path + (valid ? idx : 0u);
    uint32_t pix = p >> 2, sub = p & 3u;
    u64 s, t;
    path_seed_words(W, H, pix % W, pix / W, sub, s, t);
    LdsMem64 m{mem + lane};
    u64 dummy[ISAAC_TAIL];
    RawTail rt{valid ? out + (size_t)idx * window : dummy, window};
    isaac_seed_round<ISAAC_TAIL>(m, warm, 8700304ULL, (u64)sampling, s, t, rt);
}
