// hr_api.hip — HIP kernels (gfx950) and the C ABI of include/hanamaru_hip.h.
//
// Kernels
//   seed_isaac64_kernel   one lane = one path's ISAAC-64 generator, state in an LDS column: 80 columns x 2 KiB =
//                         all 160 KiB of a CU's LDS, one workgroup (2 waves x 40 lanes) per CU; decides the lens
//                         rejection loop in f64 and stores the last 64 raw outputs of every path (`tails`).
//   trace_kernel          the path-tracing megakernel: persistent waves pull 4x4-pixel tiles from a global counter,
//                         one lane per path, stackless threaded-BVH traversal in box / leaf phases, finished lanes
//                         are refilled with ballot / mbcnt prefix ranks; no LDS, so it co-resides with the seed
//                         kernel of the NEXT batch (own stream).
//   seed_pc_kernel        the default seeding (seed_mode = 1): producer waves run the init in registers, consumer waves the
//                         LDS-bound round; seed_isaac64_kernel is the fused form (seed_mode = 0).
//   debug_render_kernel   DebugRenderer modes (renderer.rs:101-146).
//   tonemap_gamma_kernel, bilateral_quantise_kernel   the post chain.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bvh_build.h"
#include "device_scene.h"
#include "flatten.h"
#include "gpu_bvh.h"
#include "hanamaru_hip.h"
#include "isaac_core.h"
#include "post_core.h"
#include "pt_core.h"

using namespace hr;

// ------------------------------------------------------------------------------------------ helpers

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return fail(HR_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------------ kernels

// 32-bit LDS address of a generic pointer into shared memory / load from such an address (isaac_round keeps the address of
// its next gather in a register across a scheduling fence)
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(const __attribute__((address_space(3))) void *)p; }
__device__ __forceinline__ u64 lds_load64(uint32_t a) { return *(const __attribute__((address_space(3))) u64 *)(size_t)a; }

// One generator per LDS bank column: mem[i][col], SEED_COLS columns per workgroup.
static const int SEED_COLS = 80;               // 80 x 2 KiB = 160 KiB = the whole LDS of a CU
static const int SEED_WAVES = 2, SEED_LANES = SEED_COLS / SEED_WAVES;   // 2 waves x 40 active lanes
struct LdsMem {
    u64 *col;  // &mem[0][col]
    __device__ __forceinline__ u64 ld(int i) const { return col[i * SEED_COLS]; }
    __device__ __forceinline__ uint32_t off(int i) const { return lds_addr(col) + (uint32_t)i * (uint32_t)(SEED_COLS * 8); }
    __device__ __forceinline__ u64 ldo(uint32_t o) const { return lds_load64(o); }
    __device__ __forceinline__ void st(int i, u64 v) { col[i * SEED_COLS] = v; }
};
// global-memory tail of one path: [k][64 lanes] u64 inside the item's slab
// (the padding lanes of the last group write into two spare slabs behind the last item: no predicate in the hot loop)
struct GlobalTail {
    u64 *col;  // &tail[item][0][lane]
    __device__ __forceinline__ u64 ld(int k) const { return col[k * 64]; }
    __device__ __forceinline__ void st(int k, u64 v) { col[k * 64] = v; }
};

static const size_t SEED_LDS_BYTES = (size_t)256 * SEED_COLS * 8;  // 160 KiB: mem[256][80 columns] u64

__device__ __forceinline__ void tile_lane_pixel(const RenderParams &rp, uint32_t tile, uint32_t j, uint32_t &px, uint32_t &py, uint32_t &sub) {
    uint32_t tx = tile % rp.tiles_x, ty = tile / rp.tiles_x;
    uint32_t pix = j >> 2;
    sub = j & 3u;
    px = tx * 4u + (pix & 3u);
    py = ty * 4u + (pix >> 2);
}

// tails layout: [item = tile * num_k + k][ISAAC_TAIL][64 lanes] u64;  lens layout: [item][64 lanes] u32.
// The LDS holds 80 generators, so a workgroup walks the flat path index (item * 64 + j) in strides of 80:
// its two waves (40 active lanes each) run concurrently on two SIMDs — the time of one seeding pass does not
// depend on the lane count (one wave issues at most one instruction every ~4-5 cycles), only on how many
// generator states fit in the CU's LDS.
__global__ __launch_bounds__(64 * SEED_WAVES) void seed_isaac64_kernel(RenderParams rp, int lens_shape, u64 *__restrict__ tails,
                                                                      uint32_t *__restrict__ lens, Counters *cnt) {
    extern __shared__ __align__(16) unsigned char smem[];
    u64 *mem = reinterpret_cast<u64 *>(smem);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane >= (uint32_t)SEED_LANES) return;
    const uint32_t col = wave * SEED_LANES + lane;
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
        GlobalTail gt{tails + (size_t)(pid >> 6) * ISAAC_TAIL * 64 + (pid & 63u)};
        RawLensTail<GlobalTail> lt(gt, lens_shape);
        isaac_seed_round(m, warm, 8700304ULL, (u64)(rp.sampling_begin + k * rp.stride), s, t, lt);
        lt.lens_slow();
        bool ok = lt.in_window();
        if (in_range) lens[(size_t)item * 64 + j] = ok ? (uint32_t)lt.accepted : 0u;
        if (valid && !ok) atomicAdd(&cnt->rng_overflow, 1ULL);
    }
}

// ---- producer / consumer seeding (option seed_mode = 1, the default) -----------------------------------------------------------
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
// Group buffer, per half: [row = 0 .. RING_ROWS)[40 columns] u64; rows 0 .. 8*SPLIT are the LDS image of generator words
// 0 .. 8*SPLIT - 1, the last 16 rows hold the registers.  One __syncthreads per group: in
// iteration `it` the consumers work on group it-1 while the producers complete group `it` (5 chunks per 4 groups).
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
// one dwordx4 store instruction then writes whole rows.
template <int HEAD>
struct RingState {
    u64 *pair;   // even lane: &row0[col]; odd lane: &row1[col - 1]
    bool on, odd;
    __device__ __forceinline__ RingState(u64 *col, bool on_, uint32_t lane) : on(on_), odd(lane & 1u) { pair = odd ? col + SEED_LANES - 1 : col; }
    static __device__ __forceinline__ u64 swap_pair(u64 v) {   // value of lane ^ 1
        uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
        hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, 0xB1, 0xf, 0xf, true);
        return ((u64)hi << 32) | lo;
    }
    __device__ __forceinline__ void row2(int row, u64 v0, u64 v1) {   // rows `row` (v0) and `row + 1` (v1) of this lane's column
        u64 got = swap_pair(odd ? v0 : v1);            // even receives the partner's v0, odd the partner's v1
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        u64x2 q;
        q.x = odd ? got : v0;
        q.y = odd ? v1 : got;
        if (on) *reinterpret_cast<u64x2 *>(pair + row * SEED_LANES) = q;
    }
    __device__ __forceinline__ void st2(int i, u64 v0, u64 v1) { row2(i, v0, v1); }
    __device__ __forceinline__ void end2(int j, u64 v0, u64 v1) { row2(PcLayout<HEAD>::SHIP_ROWS + j, v0, v1); }
};
struct LdsHalfMem {
    u64 *col;
    __device__ __forceinline__ u64 ld(int i) const { return col[i * SEED_LANES]; }
    __device__ __forceinline__ uint32_t off(int i) const { return lds_addr(col) + (uint32_t)i * (uint32_t)(SEED_LANES * 8); }
    __device__ __forceinline__ u64 ldo(uint32_t o) const { return lds_load64(o); }
    __device__ __forceinline__ void st(int i, u64 v) { col[i * SEED_LANES] = v; }
};
template <int SEED_SPLIT>   // = SPLIT: init blocks done by the producers
__global__ __launch_bounds__(256) void seed_pc_kernel(RenderParams rp, int lens_shape, u64 *__restrict__ ring, u64 *__restrict__ tails,
                                                      uint32_t *__restrict__ lens, Counters *cnt) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, half = wave & 1u;
    const bool consumer = wave < 2u;
    const uint32_t prio = consumer ? rp.pad[0] : rp.pad[1];
    switch (prio) {  // s_setprio takes an immediate
        case 0: break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
    const uint64_t paths = (uint64_t)rp.tiles_x * rp.tiles_y * rp.num_k * 64u;
    const uint64_t groups = (paths + SEED_COLS - 1) / SEED_COLS;
    const uint64_t G0 = groups * blockIdx.x / gridDim.x, G1 = groups * (blockIdx.x + 1) / gridDim.x;   // this workgroup's groups
    const uint64_t first_path = G0 * SEED_COLS, end_path = G1 * SEED_COLS < paths ? G1 * SEED_COLS : paths;
    u64 *ring_wg = ring + (size_t)blockIdx.x * SEED_RING_WORDS_MAX;
    const IsaacWarm warm = isaac_warm();
    typedef PcLayout<SEED_SPLIT> L;
    constexpr int SEED_SHIP_ROWS = L::SHIP_ROWS;
    constexpr size_t SEED_HALF_WORDS = L::HALF_WORDS, SEED_GROUP_WORDS = L::GROUP_WORDS;
    constexpr int CHUNKS = SEED_SHIP_ROWS * SEED_LANES * 8 / 1024;   // 1 KiB per wave-instruction
    uint64_t frontier = first_path & ~63ull;                   // first path not yet produced (chunk aligned)
    for (uint64_t it = 0; it <= G1 - G0; it++) {
        // ---- producers: complete group G0 + it
        const uint64_t need = it < G1 - G0 ? (G0 + it + 1) * SEED_COLS : 0;      // paths below `need` must be in the ring
        uint32_t n = 0;
        while (frontier < need && frontier < end_path) {
            if (!consumer && (n & 1u) == half) {
                const uint64_t pid0 = frontier + lane;
                const bool on = pid0 >= first_path && pid0 < end_path && !(rp.pad[2] & 4u);   // pad[2]: timing experiments (debug_skip)
                const uint64_t pid = pid0 >= first_path && pid0 < end_path ? pid0 : end_path - 1;
                const uint32_t item = (uint32_t)(pid >> 6), j = (uint32_t)(pid & 63u);
                uint32_t tile = item / rp.num_k, k = item - tile * rp.num_k;
                uint32_t px, py, sub;
                tile_lane_pixel(rp, tile, j, px, py, sub);
                bool valid = px < rp.width && py < rp.height;
                u64 s, t;
                path_seed_words(rp.width, rp.height, valid ? px : 0u, valid ? py : 0u, sub, s, t);
                const uint64_t g = pid / SEED_COLS;
                const uint32_t c80 = (uint32_t)(pid - g * SEED_COLS);
                RingState<SEED_SPLIT> out(ring_wg + (g & (SEED_RING_GROUPS - 1)) * SEED_GROUP_WORDS + (c80 >= (uint32_t)SEED_LANES ? SEED_HALF_WORDS : 0) + (c80 % SEED_LANES),
                              on, lane);
                isaac_init_front<SEED_SPLIT>(out, warm, 8700304ULL, (u64)(rp.sampling_begin + k * rp.stride), s, t);
            }
            frontier += 64;
            n++;
        }
        // ---- consumers: group G0 + it - 1
        if (consumer && it > 0) {
            const uint64_t g = G0 + it - 1;
            const u64 *src = ring_wg + (g & (SEED_RING_GROUPS - 1)) * SEED_GROUP_WORDS + half * SEED_HALF_WORDS;
            unsigned char *lds_half = smem + (size_t)half * SEED_LDS_HALF_BYTES;
            const uint32_t colr = lane < (uint32_t)SEED_LANES ? lane : 0u;
            u64 st16[16];
#pragma unroll
            for (int q = 0; q < 16; q++) st16[q] = __builtin_nontemporal_load(src + (size_t)(SEED_SHIP_ROWS + q) * SEED_LANES + colr);
            if (!(rp.pad[2] & 8u)) {
                const unsigned char *srcb = reinterpret_cast<const unsigned char *>(src);
                // generator words 0 .. 8*SPLIT - 1 of every column; the instruction offset advances both addresses, so one
                // address pair serves two 1 KiB copies
                static_assert(CHUNKS % 2 == 0, "fill is unrolled by two");
                const unsigned char *gsrc = srcb + lane * 16u;
                unsigned char *ldst = lds_half;
#pragma unroll 5
                for (int q = 0; q < CHUNKS; q += 2, gsrc += 2048, ldst += 2048) {
                    const void __attribute__((address_space(1))) *gp = (const void __attribute__((address_space(1))) *)gsrc;
                    void __attribute__((address_space(3))) *lp = (void __attribute__((address_space(3))) *)ldst;
                    __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 2 /* nt */);
                    __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 2);
                }
            }
            const uint64_t pid = g * SEED_COLS + half * SEED_LANES + colr;
            const bool in_range = pid < paths;
            const uint32_t item = (uint32_t)((in_range ? pid : paths - 1) >> 6), j = (uint32_t)((in_range ? pid : paths - 1) & 63u);
            uint32_t tile = item / rp.num_k;
            uint32_t px, py, sub;
            tile_lane_pixel(rp, tile, j, px, py, sub);
            bool valid = in_range && px < rp.width && py < rp.height;
            LdsHalfMem m{reinterpret_cast<u64 *>(lds_half) + colr};
            if (lane < (uint32_t)SEED_LANES) isaac_init_back<SEED_SPLIT>(m, st16);   // while the fill is in flight
            __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): this wave's half has landed (no other wave touches it)
            if (lane < (uint32_t)SEED_LANES) {
                GlobalTail gt{tails + (size_t)(pid >> 6) * ISAAC_TAIL * 64 + (pid & 63u)};
                RawLensTail<GlobalTail> lt(gt, lens_shape);
                isaac_round(m, lt);
                lt.lens_slow();
                bool ok = lt.in_window();
                if (in_range) lens[(size_t)item * 64 + j] = ok ? (uint32_t)lt.accepted : 0u;
                if (valid && !ok) atomicAdd(&cnt->rng_overflow, 1ULL);
            }
        }
        __syncthreads();   // group G0 + it is complete in the ring; the LDS and buffer (G0 + it - 1) & 3 are free again
    }
}

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
    isaac_seed_round(m, warm, 8700304ULL, (u64)sampling, s, t, rt);
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Persistent waves: a workgroup is 4 independent waves (single-wave workgroups cap residency at ~8 waves per CU);
// every wave pulls 4x4-pixel tiles from a global counter until none are left.  A tile = 64 paths per sampling of
// the batch; finished lanes are refilled from the tile's path queue, and when that runs dry the wave pulls the
// next tile while its slow lanes are still working, so lanes only starve at the very end of a launch
// (measured before: with one tile per wave the mean box-phase pass had 19.6 of 64 lanes active).
// No barriers, no LDS.
static const int TRACE_WAVES = 4;
static const int NODE_UNROLL = 2;   // box tests per pass of the box-phase loop (amortises the ballot / branch overhead)

template <bool CNT, int MINW>
__global__ __launch_bounds__(64 * TRACE_WAVES, MINW) void trace_kernel(Scene sc, RenderParams rp, const u64 *__restrict__ tails,
                                                                       const uint32_t *__restrict__ lens, float *__restrict__ accum,
                                                                       Counters *cnt, uint32_t *tile_counter) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tiles = rp.tiles_x * rp.tiles_y;
    LaneCounters lc = {0, 0, 0, 0, 0};
    uint32_t npaths = 0;
    uint32_t ph[7] = {0, 0, 0, 0, 0, 0, 0};  // wave-uniform phase statistics (counters build only)
    unsigned long long pc[4] = {0, 0, 0, 0}, tmark = 0;   // wave-cycles per phase (counters build only)
#define HR_PHASE_BEGIN() do { if (CNT) tmark = __builtin_readcyclecounter(); } while (0)
#define HR_PHASE_END(i) do { if (CNT) pc[i] += __builtin_readcyclecounter() - tmark; } while (0)
    const uint32_t total = 64u * rp.num_k;   // paths per tile in this launch: slot q = k * 64 + j
    const size_t tile_stride = (size_t)rp.num_k * ISAAC_TAIL * 64;
    uint32_t cur_tile = 0, next = total;      // wave-uniform: the tile being handed out and its queue head
    bool exhausted = false;
    Path p;
    p.q = PATH_IDLE;
    p.tile = 0;
    p.ts.cur = NODE_END; p.ts.leaf = 0; p.ts.leaf2 = 0;
    const uint32_t adv_den = rp.adv_den ? rp.adv_den : 2u;
    const uint32_t leaf_den = rp.leaf_den ? rp.leaf_den : 2u;

    for (;;) {
        // ---- A: lanes whose ray is complete: shade / NEE / next ray (or the path ends)
        if (CNT) {
            uint32_t n = (uint32_t)__popcll(__ballot(p.q != PATH_IDLE && trace_done(p.ts)));
            ph[6]++;
            if (n) { ph[0]++; ph[1] += n; }
        }
        HR_PHASE_BEGIN();
        if (p.q != PATH_IDLE && trace_done(p.ts)) {
            if (path_advance<CNT>(sc, p, tails + (size_t)p.tile * tile_stride, &lc)) {
                // The kernel uses no LDS at all (the seed kernel next to it owns all 160 KiB), so a finished
                // path adds its radiance straight into the accumulator.  A tile belongs to exactly one wave of
                // one launch, so only lanes of this wave ever touch these addresses: workgroup-scope atomics
                // (executed in the XCD's L2) are sufficient.
                uint32_t pix = (p.q & 63u) >> 2;
                uint32_t px = (p.tile % rp.tiles_x) * 4u + (pix & 3u), py = (p.tile / rp.tiles_x) * 4u + (pix >> 2);
                float *dst = accum + ((size_t)py * rp.width + px) * 3;
                __hip_atomic_fetch_add(dst + 0, p.accum.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(dst + 1, p.accum.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(dst + 2, p.accum.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                p.q = PATH_IDLE;
            }
        }
        HR_PHASE_END(0);
        // ---- B: refill idle lanes (ballot + prefix rank = live-lane compaction); pull a new tile when the queue is dry
        HR_PHASE_BEGIN();
        unsigned long long idle = __ballot(p.q == PATH_IDLE);
        if (idle) {
            if (next >= total && !exhausted) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(tile_counter, 1u);
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                if (t >= tiles) exhausted = true;
                else { cur_tile = t; next = 0; }
            }
            if (next < total) {
                uint32_t q = next + lane_rank(idle);
                if (p.q == PATH_IDLE && q < total) {
                    uint32_t k = q >> 6, j = q & 63u, px, py, sub;
                    tile_lane_pixel(rp, cur_tile, j, px, py, sub);
                    if (px < rp.width && py < rp.height) {
                        p.q = q;
                        p.tile = cur_tile;
                        p.draw_base = (uint32_t)(k * ISAAC_TAIL * 64 + j);
                        p.lens_a = lens[((size_t)cur_tile * rp.num_k + k) * 64 + j];
                        path_start(sc, rp, p, px, py, sub, tails + (size_t)cur_tile * tile_stride);
                        npaths++;
                    }
                }
                next += (uint32_t)__popcll(idle);
            }
        }
        HR_PHASE_END(1);
        const bool active = p.q != PATH_IDLE;
        const uint32_t n_active = (uint32_t)__popcll(__ballot(active));
        if (!n_active) {
            if (exhausted) break;
            continue;
        }
        // ---- C: traversal as two well-filled phases.  Box phase: lanes walk nodes until 1/leaf_den of the
        //         traversing lanes have parked a leaf; leaf phase: those lanes test their primitives together.
        //         The whole of C is left as soon as 1/adv_den of the live lanes wait for phase A.
        for (;;) {
            const bool trav = active && !trace_done(p.ts);
            const uint32_t n_trav = (uint32_t)__popcll(__ballot(trav));
            if (!n_trav || (n_active - n_trav) * adv_den >= n_active) break;
            // lanes allowed to be still walking when the leaf phase starts
            const uint32_t park = (n_trav + leaf_den - 1u) / leaf_den;
            const uint32_t walk_max = n_trav - park;
            HR_PHASE_BEGIN();
            for (;;) {
                // a lane may keep walking with ONE leaf parked (trace_node<SPEC>); it stops at the second
                const bool go = trav && p.ts.leaf2 == 0 && p.ts.cur != NODE_END;
                const uint32_t n_go = (uint32_t)__popcll(__ballot(go));
                if (n_go <= walk_max) break;
                if (CNT) { ph[2]++; ph[3] += n_go; }
                if (go) {
                    trace_node<CNT, true>(sc, p.ray, p.ts, &lc);
                    if (NODE_UNROLL > 1 && p.ts.leaf2 == 0 && p.ts.cur != NODE_END) trace_node<CNT, true>(sc, p.ray, p.ts, &lc);
                }
            }
            HR_PHASE_END(2);
            HR_PHASE_BEGIN();
            if (CNT) {
                uint32_t n = (uint32_t)__popcll(__ballot(trav && p.ts.leaf != 0));
                if (n) { ph[4]++; ph[5] += n; }
            }
            if (trav && p.ts.leaf != 0) {
                trace_leaf<CNT>(sc, p.ray, p.ts, &lc);   // clears ts.leaf
                p.ts.leaf = p.ts.leaf2;
                p.ts.leaf2 = 0;
                shadow_early_out(p);
            }
            HR_PHASE_END(3);
        }
    }
#undef HR_PHASE_BEGIN
#undef HR_PHASE_END
    if (CNT) {
        // wave reduction, one atomic per counter per wave
        unsigned long long v[6] = {npaths, lc.rays, lc.node_tests, lc.tri_tests, lc.sphere_tests, lc.cuboid_tests};
        for (int i = 0; i < 6; i++) {
            unsigned long long x = v[i];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            v[i] = x;
        }
        if (lane == 0) {
            atomicAdd(&cnt->paths, v[0]); atomicAdd(&cnt->rays, v[1]); atomicAdd(&cnt->node_tests, v[2]);
            atomicAdd(&cnt->tri_tests, v[3]); atomicAdd(&cnt->sphere_tests, v[4]); atomicAdd(&cnt->cuboid_tests, v[5]);
            atomicAdd(&cnt->shade_calls, (unsigned long long)ph[0]); atomicAdd(&cnt->shade_lanes, (unsigned long long)ph[1]);
            atomicAdd(&cnt->box_passes, (unsigned long long)ph[2]); atomicAdd(&cnt->box_lanes, (unsigned long long)ph[3]);
            atomicAdd(&cnt->leaf_calls, (unsigned long long)ph[4]); atomicAdd(&cnt->leaf_lanes, (unsigned long long)ph[5]);
            atomicAdd(&cnt->outer_iters, (unsigned long long)ph[6]);
            for (int i = 0; i < 4; i++) atomicAdd(&cnt->phase_cycles[i], pc[i]);
        }
    }
}

__global__ void intersect_debug_kernel(Scene sc, uint32_t n, const float *__restrict__ rays, float *__restrict__ out, int32_t *__restrict__ out_elem) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r;
    ray_set(r, v3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), v3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]));
    TraceState ts;
    trace_begin(ts, T_INF);
    LaneCounters lc;
    while (ts.cur != NODE_END) trace_step<false>(sc, r, ts, &lc);
    float *o = out + (size_t)i * 8;
    int32_t elem = -1;
    if (ts.prim >= 0) {
        Surf s;
        hit_surface(sc, r, ts, true, s);
        elem = s.elem;
        o[0] = 1.0f; o[1] = ts.t; o[2] = s.pos.x; o[3] = s.pos.y; o[4] = s.pos.z; o[5] = s.n.x; o[6] = s.n.y; o[7] = s.n.z;
    } else {
        o[0] = 0.0f; o[1] = ts.t;
        for (int k = 2; k < 8; k++) o[k] = 0.0f;
    }
    out_elem[i] = elem;
}

// DebugRenderer (renderer.rs:101-146): one thread per pixel, 2x2 sub-samples, pinhole rays
__global__ void debug_render_kernel(Scene sc, RenderParams rp, int mode, float *__restrict__ accum) {
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= rp.width || y >= rp.height) return;
    LaneCounters lc;
    V3f sum = v3(0, 0, 0);
    for (uint32_t sub = 0; sub < 4; sub++) sum = sum + debug_pixel<false>(sc, rp, x, y, sub, mode, &lc);
    float *o = accum + ((size_t)y * rp.width + x) * 3;
    o[0] += sum.x; o[1] += sum.y; o[2] += sum.z;
}

__global__ void tonemap_gamma_kernel(const float *__restrict__ acc, float *__restrict__ out, uint32_t n, float scale) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tonemap_gamma(acc[i * 3], acc[i * 3 + 1], acc[i * 3 + 2], scale, &out[i * 3]);
}
__global__ void bilateral_quantise_kernel(const float *__restrict__ img, uint8_t *__restrict__ out, uint32_t W, uint32_t H) {
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    bilateral_quantise(img, W, H, x, y, &out[((size_t)y * W + x) * 3]);
}

// ------------------------------------------------------------------------------------------ context

struct EventPair { hipEvent_t a, b; };

struct hr_ctx {
    int device = 0;
    int num_cus = 256;
    // streams: trace + post on `stream` (own or the caller's), the seed kernel of the NEXT batch on `seed_stream`,
    hipStream_t stream = nullptr, own_stream = nullptr, seed_stream = nullptr;
    // scene
    std::vector<void *> scene_allocs;
    Scene dsc{};
    bool have_scene = false;
    uint64_t st_nodes = 0, st_tris = 0, st_spheres = 0, st_cuboids = 0;
    // target
    uint32_t W = 0, H = 0;
    float *accum_own = nullptr, *accum = nullptr;
    float *post_tmp = nullptr;
    uint8_t *d_rgb8 = nullptr;
    // seed -> trace hand-off, double buffered (slot = batch & 1)
    u64 *tails[2] = {nullptr, nullptr};
    uint32_t *lens[2] = {nullptr, nullptr};
    size_t draws_cap = 0;                    // items (tile x sampling) per buffer
    u64 *ring = nullptr;                     // producer / consumer seeding: ring of group buffers, <= 640 KiB per CU
    int seed_split = 16;                      // option seed_split: init blocks done by the producer waves (8, 12, 16, 20, 24)
    uint32_t init_prio = 1;                  // s_setprio of the producer waves
    hipEvent_t seed_done[2] = {nullptr, nullptr}, trace_done[2] = {nullptr, nullptr};
    bool seed_pending[2] = {false, false}, trace_pending[2] = {false, false};
    uint64_t batch_counter = 0;
    Counters *d_counters = nullptr;
    uint32_t *d_tile_counter = nullptr;      // [2]: next tile of the trace launch in each slot
    // options (hr_set_option)
    bool counters = false;
    uint32_t batch = 4;                      // samplings per launch
    uint32_t adv_den = 2, leaf_den = 2;      // trace-kernel phase thresholds
    int min_waves = 5;                       // occupancy variant of the trace kernel
    int max_leaf = 4;                        // BVH leaf size (next upload)
    double split_ratio = 0.0;                // early split clipping (0 = off)
    int bvh_builder = 0;                     // 0 = host binned SAH (bvh_build.cpp), 1 = device LBVH (gpu_bvh.h); next upload
    double bvh_build_ms = 0;                 // device builder: key + sort + hierarchy + fit + emit + gather kernels
    uint64_t max_tail_bytes = 20ull << 30;   // cap of each raw-draw buffer
    int seed_mode = 1;                       // 1 = producer / consumer seed kernel, 0 = fused seed kernel
    uint32_t seed_prio = 3;                  // s_setprio of the seed / round kernel's waves
    int debug_skip = 0;                      // timing experiments only (garbage image): 2 = skip the seed kernel, 4 = no ring stores, 8 = no ring fills, 16 = skip the trace kernel
    // timing (HIP events around every launch, summed when the streams are drained)
    std::vector<EventPair> seed_events, trace_events, post_events;
    double seed_ms = 0, trace_ms = 0, post_ms = 0;
    uint64_t seed_launches = 0, trace_launches = 0;
    uint64_t paths_rendered = 0;
};

static void free_scene(hr_ctx *c) {
    for (void *p : c->scene_allocs) (void)hipFree(p);
    c->scene_allocs.clear();
    c->have_scene = false;
}
template <class T>
static int upload(hr_ctx *c, const std::vector<T> &v, const T **out) {
    void *d = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIP_TRY(hipMalloc(&d, bytes));
    c->scene_allocs.push_back(d);
    if (!v.empty()) HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T *>(d);
    return HR_OK;
}
static int drain_events(hr_ctx *c) {
    auto sum = [](std::vector<EventPair> &ev, double &acc) -> hipError_t {
        for (auto &e : ev) {
            float ms = 0;
            hipError_t r = hipEventElapsedTime(&ms, e.a, e.b);
            if (r != hipSuccess) return r;
            acc += ms;
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        ev.clear();
        return hipSuccess;
    };
    HIP_TRY(sum(c->seed_events, c->seed_ms));
    HIP_TRY(sum(c->trace_events, c->trace_ms));
    HIP_TRY(sum(c->post_events, c->post_ms));
    return HR_OK;
}
static int sync_all(hr_ctx *c) {
    HIP_TRY(hipStreamSynchronize(c->seed_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->trace_pending[0] = c->trace_pending[1] = false;
    c->seed_pending[0] = c->seed_pending[1] = false;
    return drain_events(c);
}

// ------------------------------------------------------------------------------------------ C ABI

static int create_resources(hr_ctx *c);

extern "C" {

const char *hr_last_error(void) { return g_err.c_str(); }
int hr_abi_version(void) { return HR_ABI_VERSION; }

int hr_create(int device_id, hr_ctx **out) {
    if (!out) return fail(HR_ERR_INVALID, "hr_create: out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(HR_ERR_INVALID, "hr_create: device %d not in [0,%d)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    hr_ctx *c = new hr_ctx;
    c->device = device_id;
    int rc = create_resources(c);
    if (rc) { (void)hr_destroy(c); return rc; }
    *out = c;
    return HR_OK;
}

static int create_resources(hr_ctx *c) {
    const int device_id = c->device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->seed_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipEventCreateWithFlags(&c->seed_done[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->trace_done[i], hipEventDisableTiming));
    }
    HIP_TRY(hipMalloc((void **)&c->d_counters, sizeof(Counters)));
    HIP_TRY(hipMemset(c->d_counters, 0, sizeof(Counters)));
    HIP_TRY(hipMalloc((void **)&c->d_tile_counter, 2 * sizeof(uint32_t)));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_isaac64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_debug_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 8));
    return HR_OK;
}

int hr_destroy(hr_ctx *c) {
    if (!c) return HR_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    free_scene(c);
    for (auto *ev : {&c->seed_events, &c->trace_events, &c->post_events})
        for (auto &e : *ev) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (c->accum_own) (void)hipFree(c->accum_own);
    for (int i = 0; i < 2; i++) {
        if (c->tails[i]) (void)hipFree(c->tails[i]);
        if (c->lens[i]) (void)hipFree(c->lens[i]);
        if (c->seed_done[i]) (void)hipEventDestroy(c->seed_done[i]);
        if (c->trace_done[i]) (void)hipEventDestroy(c->trace_done[i]);
    }
    if (c->ring) (void)hipFree(c->ring);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_tile_counter) (void)hipFree(c->d_tile_counter);
    if (c->post_tmp) (void)hipFree(c->post_tmp);
    if (c->d_rgb8) (void)hipFree(c->d_rgb8);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->seed_stream) (void)hipStreamDestroy(c->seed_stream);
    delete c;
    return HR_OK;
}

// Device LBVH (option bvh_builder = 1): the primitive arrays were uploaded in input order; build the tree over them,
// re-store the primitives in leaf order and point the scene at the results.  Scratch is freed before returning.
static int build_bvh_on_device(hr_ctx *c, const HostScene &hs) {
    using namespace lbvh;
    Scene &d = c->dsc;
    const int n = (int)(d.num_tris + d.num_spheres + d.num_cuboids);
    const int N = 2 * n - 1;
    Prims p{};
    p.tris = d.tris; p.num_tris = d.num_tris; p.spheres = d.spheres; p.num_spheres = d.num_spheres; p.cuboids = d.cuboids; p.num_cuboids = d.num_cuboids;
    for (int a = 0; a < 3; a++) {
        double ext = hs.scene_max[a] - hs.scene_min[a];
        p.smin[a] = (float)hs.scene_min[a];
        p.sinv[a] = ext > 0 ? (float)(1.0 / ext) : 0.0f;
    }
    std::vector<void *> scratch;
    auto cleanup = [&]() { for (void *q : scratch) (void)hipFree(q); };
    auto alloc = [&](size_t bytes, bool keep) -> void * {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
        (keep ? c->scene_allocs : scratch).push_back(q);
        return q;
    };
#define LBVH_ALLOC(var, type, count, keep)                                                                        \
    type *var = (type *)alloc(sizeof(type) * (size_t)(count), keep);                                                \
    if (!var) { cleanup(); return fail(HR_ERR_DEVICE, "hr_upload_scene: out of device memory in the BVH build"); }
    LBVH_ALLOC(keys_in, mkey_t, n, false)
    LBVH_ALLOC(keys, mkey_t, n, false)
    Work w{};
    LBVH_ALLOC(parent, uint32_t, N, false) LBVH_ALLOC(left, uint32_t, n, false) LBVH_ALLOC(right, uint32_t, n, false)
    LBVH_ALLOC(first, uint32_t, n, false) LBVH_ALLOC(last, uint32_t, n, false) LBVH_ALLOC(flags, uint32_t, n, false)
    LBVH_ALLOC(bmin, float, 3 * (size_t)N, false) LBVH_ALLOC(bmax, float, 3 * (size_t)N, false)
    LBVH_ALLOC(word, uint32_t, N, false) LBVH_ALLOC(axis_low, uint32_t, n, false)
    w.parent = parent; w.left = left; w.right = right; w.first = first; w.last = last; w.flags = flags;
    w.bmin = bmin; w.bmax = bmax; w.word = word; w.axis_low = axis_low;
    LBVH_ALLOC(nodes, Node, 8 * (size_t)N, true)
    LBVH_ALLOC(tris, Tri, d.num_tris, true)
    LBVH_ALLOC(spheres, f4, d.num_spheres, true)
    LBVH_ALLOC(sphere_elem, int32_t, d.num_spheres, true)
    LBVH_ALLOC(cuboids, f4, 2 * (size_t)d.num_cuboids, true)
    size_t sort_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys_in, keys, n, 0, 64, c->stream);
    if (e != hipSuccess) { cleanup(); return fail(HR_ERR_DEVICE, "hipcub sort (size query): %s", hipGetErrorString(e)); }
    LBVH_ALLOC(sort_tmp, unsigned char, sort_bytes, false)
#undef LBVH_ALLOC
    hipEvent_t ea = nullptr, eb = nullptr;
    (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    const int T = 256;
    hipStream_t st = c->stream;
    (void)hipEventRecord(ea, st);
    e = hipMemsetAsync(flags, 0, sizeof(uint32_t) * (size_t)n, st);
    if (e == hipSuccess) e = hipMemsetAsync(parent, 0xff, sizeof(uint32_t) * (size_t)N, st);   // n == 1: the lone leaf is the root
    if (e == hipSuccess) {
        key_kernel<<<(n + T - 1) / T, T, 0, st>>>(p, n, keys_in);
        e = hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, keys_in, keys, n, 0, 64, st);
    }
    if (e == hipSuccess) {
        if (n > 1) hierarchy_kernel<<<(n - 1 + T - 1) / T, T, 0, st>>>(keys, n, w);
        fit_kernel<<<(n + T - 1) / T, T, 0, st>>>(p, keys, n, (uint32_t)c->max_leaf, w);
        emit_kernel<<<(8 * N + T - 1) / T, T, 0, st>>>(n, w, nodes);
        gather_kernel<<<(n + T - 1) / T, T, 0, st>>>(p, keys, n, tris, spheres, sphere_elem, d.sphere_elem, cuboids);
        e = hipGetLastError();
    }
    (void)hipEventRecord(eb, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    float ms = 0;
    if (e == hipSuccess) (void)hipEventElapsedTime(&ms, ea, eb);
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    cleanup();
    if (e != hipSuccess) return fail(HR_ERR_DEVICE, "device BVH build: %s", hipGetErrorString(e));
    c->bvh_build_ms = ms;
    d.nodes = nodes; d.num_nodes = (uint32_t)N;
    d.tris = tris; d.spheres = spheres; d.sphere_elem = sphere_elem; d.cuboids = cuboids;   // the input-order copies stay in scene_allocs until the next upload
    return HR_OK;
}

int hr_upload_scene(hr_ctx *c, const hr_scene_desc *sd) {
    if (!c || !sd) return fail(HR_ERR_INVALID, "hr_upload_scene: null argument");
    if (!sd->elements || sd->num_elements == 0) return fail(HR_ERR_INVALID, "hr_upload_scene: scene has no elements");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    free_scene(c);

    HostScene hs;
    std::string ferr;
    const bool gpu_build = c->bvh_builder == 1;
    rc = flatten_scene(sd, hs, ferr, c->max_leaf, gpu_build ? 0.0 : c->split_ratio, !gpu_build);
    if (rc) return fail(rc, "hr_upload_scene: %s", ferr.c_str());
    Scene &d = c->dsc;
    d = hs.view();
    int r;
    if ((r = upload(c, hs.tris, &d.tris))) return r;
    if ((r = upload(c, hs.spheres, &d.spheres))) return r;
    if ((r = upload(c, hs.sphere_elem, &d.sphere_elem))) return r;
    if ((r = upload(c, hs.cuboids, &d.cuboids))) return r;
    if ((r = upload(c, hs.materials, &d.materials))) return r;
    if ((r = upload(c, hs.images, &d.images))) return r;
    if ((r = upload(c, hs.emitters, &d.emitters))) return r;
    if ((r = upload(c, hs.texels, &d.texels))) return r;
    c->bvh_build_ms = 0;
    if (gpu_build) { if ((r = build_bvh_on_device(c, hs))) return r; }
    else if ((r = upload(c, hs.nodes, &d.nodes))) return r;
    c->st_nodes = d.num_nodes; c->st_tris = d.num_tris; c->st_spheres = d.num_spheres; c->st_cuboids = d.num_cuboids;
    c->have_scene = true;
    return HR_OK;
}

int hr_set_resolution(hr_ctx *c, uint32_t w, uint32_t h) {
    if (!c || !w || !h) return fail(HR_ERR_INVALID, "hr_set_resolution: bad argument");
    if ((uint64_t)w * h > (1ull << 27)) return fail(HR_ERR_UNSUPPORTED, "resolution too large");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    bool external = c->accum && c->accum != c->accum_own;
    if (c->accum_own) { HIP_TRY(hipFree(c->accum_own)); c->accum_own = nullptr; }
    if (c->post_tmp) { HIP_TRY(hipFree(c->post_tmp)); c->post_tmp = nullptr; }
    if (c->d_rgb8) { HIP_TRY(hipFree(c->d_rgb8)); c->d_rgb8 = nullptr; }
    c->W = w; c->H = h;
    size_t n = (size_t)w * h * 3;
    HIP_TRY(hipMalloc((void **)&c->accum_own, n * sizeof(float)));
    HIP_TRY(hipMemset(c->accum_own, 0, n * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&c->post_tmp, n * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&c->d_rgb8, n));
    if (!external) c->accum = c->accum_own;
    return HR_OK;
}

int hr_bind_accumulator(hr_ctx *c, float *device_rgb) {
    if (!c) return fail(HR_ERR_INVALID, "hr_bind_accumulator: null ctx");
    int rc = sync_all(c);
    if (rc) return rc;
    c->accum = device_rgb ? device_rgb : c->accum_own;
    return HR_OK;
}
void *hr_accumulator_device_ptr(hr_ctx *c) { return c ? c->accum : nullptr; }

int hr_set_stream(hr_ctx *c, void *s) {
    if (!c) return fail(HR_ERR_INVALID, "hr_set_stream: null ctx");
    int rc = sync_all(c);
    if (rc) return rc;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return HR_OK;
}

int hr_clear(hr_ctx *c) {
    if (!c) return fail(HR_ERR_INVALID, "hr_clear: null ctx");
    if (!c->accum) return fail(HR_ERR_NO_TARGET, "hr_clear: no accumulator (call hr_set_resolution)");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->accum, 0, (size_t)c->W * c->H * 3 * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(Counters), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->seed_ms = c->trace_ms = c->post_ms = 0;
    c->seed_launches = c->trace_launches = 0;
    c->paths_rendered = 0;
    return HR_OK;
}

static int ensure_draws(hr_ctx *c, size_t items) {
    if (items <= c->draws_cap) return HR_OK;
    for (int i = 0; i < 2; i++) {
        if (c->tails[i]) { HIP_TRY(hipFree(c->tails[i])); c->tails[i] = nullptr; }
        if (c->lens[i]) { HIP_TRY(hipFree(c->lens[i])); c->lens[i] = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->tails[i], (items + 2) * ISAAC_TAIL * 64 * sizeof(u64)));   // + 2 slabs for the padding lanes of the last group
        HIP_TRY(hipMalloc((void **)&c->lens[i], items * 64 * sizeof(uint32_t)));
    }
    c->draws_cap = items;
    return HR_OK;
}

static int launch_seed(hr_ctx *c, const RenderParams &rp, int slot, hipStream_t st) {
    uint64_t paths = (uint64_t)rp.tiles_x * rp.tiles_y * rp.num_k * 64u;
    uint32_t grid = (uint32_t)std::min<uint64_t>((paths + SEED_COLS - 1) / SEED_COLS, (uint64_t)c->num_cus);
    EventPair ev;
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventRecord(ev.a, st));
    if (c->debug_skip & 2) {
    } else if (c->seed_mode == 1) {
        if (!c->ring) HIP_TRY(hipMalloc((void **)&c->ring, (size_t)c->num_cus * SEED_RING_WORDS_MAX * sizeof(u64)));
#define HR_LAUNCH_PC(HEAD) hipLaunchKernelGGL(seed_pc_kernel<HEAD>, dim3(grid), dim3(256), SEED_LDS_BYTES, st, rp, c->dsc.cam.lens_shape, c->ring, c->tails[slot], c->lens[slot], c->d_counters)
        switch (c->seed_split) {
            case 8: HR_LAUNCH_PC(8); break;
            case 12: HR_LAUNCH_PC(12); break;
            case 20: HR_LAUNCH_PC(20); break;
            case 24: HR_LAUNCH_PC(24); break;
            default: HR_LAUNCH_PC(16); break;
        }
#undef HR_LAUNCH_PC
    } else
        hipLaunchKernelGGL(seed_isaac64_kernel, dim3(grid), dim3(64 * SEED_WAVES), SEED_LDS_BYTES, st, rp, c->dsc.cam.lens_shape, c->tails[slot],
                           c->lens[slot], c->d_counters);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev.b, st));
    c->seed_events.push_back(ev);
    c->seed_launches++;
    return HR_OK;
}

int hr_render(hr_ctx *c, uint32_t s_begin, uint32_t s_end, uint32_t stride) {
    if (!c || !stride) return fail(HR_ERR_INVALID, "hr_render: bad argument");
    if (!c->have_scene) return fail(HR_ERR_NO_SCENE, "hr_render: no scene uploaded");
    if (!c->accum || !c->W) return fail(HR_ERR_NO_TARGET, "hr_render: hr_set_resolution not called");
    if (s_end <= s_begin) return HR_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint32_t total_k = (s_end - s_begin + stride - 1) / stride;
    RenderParams rp{};
    rp.width = c->W; rp.height = c->H;
    rp.tiles_x = (c->W + 3) / 4; rp.tiles_y = (c->H + 3) / 4;
    rp.stride = stride;
    rp.adv_den = c->adv_den;
    rp.leaf_den = c->leaf_den;
    rp.pad[0] = c->seed_prio;
    rp.pad[1] = c->init_prio;
    rp.pad[2] = (uint32_t)c->debug_skip;
    uint32_t tiles = rp.tiles_x * rp.tiles_y;
    // the raw-draw hand-off costs 32 KiB per (tile, sampling): keep each of the two buffers under max_tail_bytes
    uint32_t batch = std::max<uint32_t>(1, c->batch);
    {
        uint64_t per_sampling = (uint64_t)tiles * ISAAC_TAIL * 64 * sizeof(u64);
        uint64_t fit = std::max<uint64_t>(1, c->max_tail_bytes / std::max<uint64_t>(1, per_sampling));
        batch = (uint32_t)std::min<uint64_t>(batch, fit);
    }
    int rc = ensure_draws(c, (size_t)tiles * batch);
    if (rc) return rc;
    for (uint32_t done = 0; done < total_k; done += batch) {
        uint32_t nk = std::min(batch, total_k - done);
        rp.sampling_begin = s_begin + done * stride;
        rp.num_k = nk;
        int slot = (int)(c->batch_counter & 1);
        c->batch_counter++;
        hipStream_t sstream = c->seed_stream;  // (alternating two seed streams to overlap kernel tails was measured: no gain)
        // seed of this batch may only overwrite draws[slot] once the trace that read it has finished
        if (c->trace_pending[slot]) HIP_TRY(hipStreamWaitEvent(sstream, c->trace_done[slot], 0));
        if ((rc = launch_seed(c, rp, slot, sstream))) return rc;
        HIP_TRY(hipEventRecord(c->seed_done[slot], sstream));
        c->seed_pending[slot] = true;
        HIP_TRY(hipStreamWaitEvent(c->stream, c->seed_done[slot], 0));
        EventPair ev;
        HIP_TRY(hipEventCreate(&ev.a));
        HIP_TRY(hipEventCreate(&ev.b));
        HIP_TRY(hipEventRecord(ev.a, c->stream));
        // persistent waves: enough workgroups to fill every CU (6 per CU covers every occupancy variant), never more
        // waves than tiles
        uint32_t grid = std::min<uint32_t>((uint32_t)c->num_cus * 6u, (tiles + TRACE_WAVES - 1) / TRACE_WAVES);
        HIP_TRY(hipMemsetAsync(c->d_tile_counter + slot, 0, sizeof(uint32_t), c->stream));
        {
            dim3 g(grid), b(64 * TRACE_WAVES);
#define HR_LAUNCH_TRACE(C, W) hipLaunchKernelGGL((trace_kernel<C, W>), g, b, 0, c->stream, c->dsc, rp, c->tails[slot], c->lens[slot], c->accum, c->d_counters, c->d_tile_counter + slot)
            if (c->debug_skip & 16) {
            } else if (c->counters) HR_LAUNCH_TRACE(true, 3);
            else if (c->min_waves == 4) HR_LAUNCH_TRACE(false, 4);
            else if (c->min_waves == 5) HR_LAUNCH_TRACE(false, 5);
            else if (c->min_waves == 6) HR_LAUNCH_TRACE(false, 6);
            else HR_LAUNCH_TRACE(false, 3);
#undef HR_LAUNCH_TRACE
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev.b, c->stream));
        c->trace_events.push_back(ev);
        c->trace_launches++;
        HIP_TRY(hipEventRecord(c->trace_done[slot], c->stream));
        c->trace_pending[slot] = true;
        c->paths_rendered += (uint64_t)c->W * c->H * 4 * nk;
        if (c->trace_events.size() > 4096) {  // keep the event list bounded on very long renders
            if ((rc = sync_all(c))) return rc;
        }
    }
    return HR_OK;
}

int hr_render_debug(hr_ctx *c, int mode) {
    if (!c || mode < 0 || mode > 3) return fail(HR_ERR_INVALID, "hr_render_debug: mode must be 0..3");
    if (!c->have_scene) return fail(HR_ERR_NO_SCENE, "hr_render_debug: no scene uploaded");
    if (!c->accum || !c->W) return fail(HR_ERR_NO_TARGET, "hr_render_debug: hr_set_resolution not called");
    HIP_TRY(hipSetDevice(c->device));
    RenderParams rp{};
    rp.width = c->W; rp.height = c->H;
    hipLaunchKernelGGL(debug_render_kernel, dim3((c->W + 15) / 16, (c->H + 15) / 16), dim3(16, 16), 0, c->stream, c->dsc, rp, mode, c->accum);
    HIP_TRY(hipGetLastError());
    return HR_OK;
}

int hr_synchronize(hr_ctx *c) {
    if (!c) return fail(HR_ERR_INVALID, "hr_synchronize: null ctx");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    Counters h;
    HIP_TRY(hipMemcpy(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
    if (h.rng_overflow) return fail(HR_ERR_RNG_WINDOW, "%llu paths needed more than %d ISAAC-64 outputs for the lens rejection loop", h.rng_overflow, ISAAC_TAIL);
    return HR_OK;
}

int hr_read_accumulator(hr_ctx *c, float *host) {
    if (!c || !host) return fail(HR_ERR_INVALID, "hr_read_accumulator: null argument");
    if (!c->accum) return fail(HR_ERR_NO_TARGET, "hr_read_accumulator: no accumulator");
    int rc = hr_synchronize(c);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(host, c->accum, (size_t)c->W * c->H * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return HR_OK;
}
int hr_write_accumulator(hr_ctx *c, const float *host) {
    if (!c || !host) return fail(HR_ERR_INVALID, "hr_write_accumulator: null argument");
    if (!c->accum) return fail(HR_ERR_NO_TARGET, "hr_write_accumulator: no accumulator");
    int rc = sync_all(c);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(c->accum, host, (size_t)c->W * c->H * 3 * sizeof(float), hipMemcpyHostToDevice));
    return HR_OK;
}

int hr_resolve(hr_ctx *c, uint32_t samplings, uint8_t *host_rgb8) {
    if (!c || !host_rgb8 || !samplings) return fail(HR_ERR_INVALID, "hr_resolve: bad argument");
    if (!c->accum) return fail(HR_ERR_NO_TARGET, "hr_resolve: no accumulator");
    int rc = hr_synchronize(c);
    if (rc) return rc;
    uint32_t n = c->W * c->H;
    float scale = 1.0f / (float)(samplings * 4u);
    EventPair ev;
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventRecord(ev.a, c->stream));
    hipLaunchKernelGGL(tonemap_gamma_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->accum, c->post_tmp, n, scale);
    hipLaunchKernelGGL(bilateral_quantise_kernel, dim3((c->W + 31) / 32, (c->H + 7) / 8), dim3(32, 8), 0, c->stream, c->post_tmp, c->d_rgb8, c->W, c->H);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev.b, c->stream));
    c->post_events.push_back(ev);
    HIP_TRY(hipMemcpyAsync(host_rgb8, c->d_rgb8, (size_t)n * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return drain_events(c);
}

int hr_get_stats(hr_ctx *c, hr_stats *out) {
    if (!c || !out) return fail(HR_ERR_INVALID, "hr_get_stats: null argument");
    int rc = sync_all(c);
    if (rc) return rc;
    Counters h;
    HIP_TRY(hipMemcpy(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->paths = c->counters ? h.paths : c->paths_rendered;
    out->rays = h.rays; out->node_tests = h.node_tests; out->tri_tests = h.tri_tests;
    out->sphere_tests = h.sphere_tests; out->cuboid_tests = h.cuboid_tests; out->rng_overflow = h.rng_overflow;
    out->seed_kernel_ms = c->seed_ms; out->trace_kernel_ms = c->trace_ms; out->post_kernel_ms = c->post_ms;
    out->seed_launches = c->seed_launches; out->trace_launches = c->trace_launches;
    out->bvh_build_ms = c->bvh_build_ms;
    out->shade_calls = h.shade_calls; out->shade_lanes = h.shade_lanes; out->box_passes = h.box_passes; out->box_lanes = h.box_lanes;
    out->leaf_calls = h.leaf_calls; out->leaf_lanes = h.leaf_lanes; out->outer_iters = h.outer_iters;
    for (int i = 0; i < 4; i++) out->phase_cycles[i] = h.phase_cycles[i];
    out->bvh_nodes = c->st_nodes; out->triangles = c->st_tris; out->spheres = c->st_spheres; out->cuboids = c->st_cuboids;
    return HR_OK;
}

int hr_set_option(hr_ctx *c, const char *key, double value) {
    if (!c || !key) return fail(HR_ERR_INVALID, "hr_set_option: null argument");
    std::string k = key;
    if (k == "counters") { c->counters = value != 0.0; return HR_OK; }
    if (k == "batch") {
        if (value < 1 || value > 64) return fail(HR_ERR_INVALID, "batch must be in [1,64]");
        int rc = sync_all(c);
        if (rc) return rc;
        c->batch = (uint32_t)value;
        return HR_OK;
    }
    if (k == "adv_den") {
        if (value < 1 || value > 64) return fail(HR_ERR_INVALID, "adv_den must be in [1,64]");
        c->adv_den = (uint32_t)value;
        return HR_OK;
    }
    if (k == "min_waves") {
        if (value < 3 || value > 6) return fail(HR_ERR_INVALID, "min_waves must be in [3,6]");
        c->min_waves = (int)value;
        return HR_OK;
    }
    if (k == "max_tail_gib") {
        if (value < 1 || value > 128) return fail(HR_ERR_INVALID, "max_tail_gib must be in [1,128]");
        c->max_tail_bytes = (uint64_t)value << 30;
        return HR_OK;
    }
    if (k == "seed_prio") {
        if (value < 0 || value > 3) return fail(HR_ERR_INVALID, "seed_prio must be in [0,3]");
        c->seed_prio = (uint32_t)value;
        return HR_OK;
    }
    if (k == "seed_split") {
        if (value != 8 && value != 12 && value != 16 && value != 20 && value != 24) return fail(HR_ERR_INVALID, "seed_split must be 8, 12, 16, 20 or 24");
        c->seed_split = (int)value;
        return HR_OK;
    }
    if (k == "init_prio") {
        if (value < 0 || value > 3) return fail(HR_ERR_INVALID, "init_prio must be in [0,3]");
        c->init_prio = (uint32_t)value;
        return HR_OK;
    }
    if (k == "debug_skip") { c->debug_skip = (int)value; return HR_OK; }
    if (k == "seed_mode") {
        if (value != 0 && value != 1) return fail(HR_ERR_INVALID, "seed_mode must be 1 (producer / consumer waves, default) or 0 (fused kernel)");
        int rc = sync_all(c);
        if (rc) return rc;
        c->seed_mode = (int)value;
        return HR_OK;
    }
    if (k == "split_ratio") {  // early split clipping of triangle references (0 = off), next hr_upload_scene
        if (value < 0 || value > 1000) return fail(HR_ERR_INVALID, "split_ratio must be in [0,1000]");
        c->split_ratio = value;
        return HR_OK;
    }
    if (k == "bvh_builder") {  // 0 = host binned SAH, 1 = device LBVH; takes effect at the next hr_upload_scene
        if (value != 0 && value != 1) return fail(HR_ERR_INVALID, "bvh_builder must be 0 (host SAH) or 1 (device LBVH)");
        c->bvh_builder = (int)value;
        return HR_OK;
    }
    if (k == "max_leaf") {  // takes effect at the next hr_upload_scene
        if (value < 1 || value > 15) return fail(HR_ERR_INVALID, "max_leaf must be in [1,15]");
        c->max_leaf = (int)value;
        return HR_OK;
    }
    if (k == "leaf_den") {
        if (value < 1 || value > 64) return fail(HR_ERR_INVALID, "leaf_den must be in [1,64]");
        c->leaf_den = (uint32_t)value;
        return HR_OK;
    }
    if (k == "rng_window") {
        if ((int)value != ISAAC_TAIL) return fail(HR_ERR_UNSUPPORTED, "rng_window is fixed at %d in this build", ISAAC_TAIL);
        return HR_OK;
    }
    return fail(HR_ERR_INVALID, "unknown option '%s'", key);
}

int hr_debug_draws(hr_ctx *c, uint32_t sampling, uint32_t first_path, uint32_t num_paths, uint32_t window, uint64_t *host_out) {
    if (!c || !host_out || !num_paths) return fail(HR_ERR_INVALID, "hr_debug_draws: bad argument");
    if (!c->W) return fail(HR_ERR_NO_TARGET, "hr_debug_draws: hr_set_resolution not called");
    if (window == 0 || window > (uint32_t)ISAAC_TAIL) return fail(HR_ERR_INVALID, "window must be in [1,%d]", ISAAC_TAIL);
    if ((uint64_t)first_path + num_paths > (uint64_t)c->W * c->H * 4) return fail(HR_ERR_INVALID, "path range outside the image");
    HIP_TRY(hipSetDevice(c->device));
    u64 *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (size_t)num_paths * window * 8));
    hipLaunchKernelGGL(seed_debug_kernel, dim3((num_paths + 63) / 64), dim3(64), 256 * 64 * 8, c->stream, c->W, c->H, sampling, first_path, num_paths,
                       (int)window, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(host_out, d, (size_t)num_paths * window * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(HR_ERR_DEVICE, "hr_debug_draws: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_debug_path_draws(hr_ctx *c, uint32_t sampling, float *host_out) {
    // the 20 fp32 draws per path exactly as the production seed kernel hands them to the trace kernel,
    // re-ordered to pixel-major paths: out[((y*W + x)*4 + sub) * 20 + d]
    if (!c || !host_out) return fail(HR_ERR_INVALID, "hr_debug_path_draws: bad argument");
    if (!c->W) return fail(HR_ERR_NO_TARGET, "hr_debug_path_draws: hr_set_resolution not called");
    if (!c->have_scene) return fail(HR_ERR_NO_SCENE, "hr_debug_path_draws: no scene (lens shape needed)");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    RenderParams rp{};
    rp.width = c->W; rp.height = c->H; rp.tiles_x = (c->W + 3) / 4; rp.tiles_y = (c->H + 3) / 4;
    rp.sampling_begin = sampling; rp.stride = 1; rp.num_k = 1;
    uint32_t tiles = rp.tiles_x * rp.tiles_y;
    if ((rc = ensure_draws(c, tiles))) return rc;
    if ((rc = launch_seed(c, rp, 0, c->stream))) return rc;
    std::vector<u64> h((size_t)tiles * ISAAC_TAIL * 64);
    std::vector<uint32_t> hl((size_t)tiles * 64);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(h.data(), c->tails[0], h.size() * sizeof(u64), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hl.data(), c->lens[0], hl.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t t = 0; t < tiles; t++)
        for (uint32_t j = 0; j < 64; j++) {
            uint32_t tx = t % rp.tiles_x, ty = t / rp.tiles_x, pix = j >> 2, sub = j & 3;
            uint32_t px = tx * 4 + (pix & 3), py = ty * 4 + (pix >> 2);
            if (px >= c->W || py >= c->H) continue;
            const u64 *col = &h[(size_t)t * ISAAC_TAIL * 64 + j];
            uint32_t a = hl[(size_t)t * 64 + j];
            float *o = &host_out[(((size_t)py * c->W + px) * 4 + sub) * DRAWS_PER_PATH];
            o[0] = draw_lens_f32(col[(2 * a) * 64]);
            o[1] = draw_lens_f32(col[(2 * a + 1) * 64]);
            for (uint32_t d = 2; d < (uint32_t)DRAWS_PER_PATH; d++) o[d] = draw_f32(col[(2 * a + d) * 64]);
        }
    return drain_events(c);
}

int hr_debug_intersect(hr_ctx *c, uint32_t n, const float *rays, float *out, int32_t *out_element) {
    if (!c || !rays || !out || !out_element || !n) return fail(HR_ERR_INVALID, "hr_debug_intersect: bad argument");
    if (!c->have_scene) return fail(HR_ERR_NO_SCENE, "hr_debug_intersect: no scene uploaded");
    HIP_TRY(hipSetDevice(c->device));
    float *d_rays = nullptr, *d_out = nullptr;
    int32_t *d_el = nullptr;
    hipError_t e = hipMalloc((void **)&d_rays, (size_t)n * 6 * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, (size_t)n * 8 * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_el, (size_t)n * 4);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)n * 6 * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(intersect_debug_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->dsc, n, d_rays, d_out, d_el);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * 8 * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_element, d_el, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_rays); (void)hipFree(d_out); (void)hipFree(d_el);
    if (e != hipSuccess) return fail(HR_ERR_DEVICE, "hr_debug_intersect: %s", hipGetErrorString(e));
    return HR_OK;
}

}  // extern "C"
