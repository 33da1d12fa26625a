// hr_api.hip — the device context and the C ABI of include/hanamaru_hip.h (+ the test / measurement entry points of hanamaru_hip_debug.h) (gfx950).  One translation unit with its kernels:
//   seed_kernels.h   seed_seg_kernel (default: the ISAAC-64 init sweep as three runs computed side by side from states the producer
//                    waves work out ahead in registers, consumer waves run the LDS-bound round), seed_pc_kernel (the same roles with
//                    a ring of generator words), seed_isaac64_kernel (fused form), seed_debug_kernel
//   trace_kernel.h   trace_kernel — the path-tracing megakernel: persistent waves pull 4x4-pixel tiles from a global
//                    counter, one lane per path, stackless threaded-BVH traversal in box / leaf phases, finished lanes are
//                    refilled with ballot / mbcnt prefix ranks; no LDS, so it co-resides with the seed kernel of the NEXT
//                    batch (own stream) — plus trace_debug_kernel (the same traversal, for hr_debug_trace), intersect_debug_kernel
//                    and debug_render_kernel (renderer.rs:101-146)
//   wf_kernels.h     the split pipeline (option trace_mode 1): wf_start_kernel, wf_traverse_kernel (the same traversal at <= 64 VGPRs),
//                    wf_shade_kernel — the megakernel cut at scene.intersect, the path parked in HBM between the two
//   post_kernels.h   tonemap_gamma_kernel, bilateral_quantise_kernel
//   gpu_bvh.h        the device BVH builders' kernels (option bvh_builder = 1 LBVH, 2 PLOC)
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
#include "hanamaru_hip_debug.h"
#include "hr_comm.h"
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

#include "seed_kernels.h"
#include "trace_kernel.h"
#include "wf_kernels.h"
#include "post_kernels.h"

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
    // multi-GPU: RCCL communicator of this rank, and the all-reduced accumulator (valid until the next render / clear / write)
    hrcomm::Comm comm = nullptr;
    int comm_world = 0, comm_rank = 0;
    int comm_path = HR_COMM_NONE;              // how the group was formed (hr_comm_info)
    uint64_t allreduces = 0;                   // collectives this context has enqueued since its communicator was made
    std::vector<hr_ctx *> same_device_peers;   // hr_comm_init_local over contexts that share ONE device: summed by a kernel, not by RCCL
    float *accum_total = nullptr;
    bool total_valid = false;
    // seed -> trace hand-off, double buffered (slot = batch & 1)
    float *recs[2] = {nullptr, nullptr};     // 128-byte record per path (device_scene.h)
    uint32_t *ovf = nullptr;                 // per consumer wave: list of the paths it re-derives at the end of a launch (seed_fixup_wave)
    uint32_t ovf_cap = 0;                    // entries per consumer wave the list is allocated for (ensure_ovf)
    u64 *ovf_win = nullptr;                  // per consumer wave: raw-output window of that fix-up
    size_t draws_cap = 0;                    // items (tile x sampling) per buffer
    uint64_t rec_lo_off = 0;                 // floats from a record buffer's start to its twin (the draws' residuals, precise shading); 0: the buffers have none
    int draw_residuals = 1;                  // debug option: precise shading's records carry the draws' residuals (0: the fp32 draws alone, the A/B)
    u64 *ring = nullptr;                     // seed kernels' hand-off ring (three-run kernel: 120 KiB per CU of 16-register states; ring kernel: <= 680 KiB per CU)
    int seed_split = 16;                      // option seed_split: init blocks done by the producer waves (8, 12, 16, 20, 24, 28)
    uint32_t init_prio = 1;                  // s_setprio of the producer waves
    hipEvent_t seed_done[2] = {nullptr, nullptr}, trace_done[2] = {nullptr, nullptr};
    bool seed_pending[2] = {false, false}, trace_pending[2] = {false, false};
    uint64_t batch_counter = 0;
    Counters *d_counters = nullptr;
    uint32_t *d_tile_counter = nullptr;      // [2]: next tile of the trace launch in each slot
    // options (hr_set_option)
    bool counters = false;
    uint32_t batch = 0;                      // samplings per launch; 0 = automatic: about 33 M paths per launch (4 at 1080p, more for small images)
    uint32_t adv_den = 2, leaf_den = 2;      // trace-kernel phase thresholds
    int min_waves = 5;                       // occupancy variant of the trace kernel
    uint32_t trace_wgs = 6;                  // trace-kernel workgroups per CU in the grid (persistent waves)
    uint32_t tail_div = 16;                  // debug: the last 1 / tail_div of a launch's tiles go out one sampling at a time (0 = off)
    uint32_t trace_grid = 0, trace_budget = 0;   // debug: absolute grid size (0 = trace_wgs per CU) / workgroups that stay (0 = all)
    uint32_t node_unroll = 2;                // box phase: node visits per pass of the loop
    uint32_t kchunk = 0;                     // samplings per work unit of the trace kernel (0 = 4)
    int trace_boost = -1;                    // which kernel's waves come first: -1 = governed on the device from the kernels' own time stamps (default), 0 .. 4 = fixed level
    GovDev *gov = nullptr;                   // the governor's state (device memory; device_scene.h)
    bool quant_nodes = true;                 // trace kernel walks the 16-byte quantised nodes (host-built trees; next upload)
    int max_leaf = 4;                        // BVH leaf size (next upload)
    double split_ratio = -1.0;               // early split clipping: -1 = automatic (kept when it cuts the SAH cost by more than 7 %), 0 = off, > 0 = ratio
    int bvh_builder = -1;                    // -1 = by scene size (default: host SAH below AUTO_BUILDER_PRIMS primitives, device PLOC from there on), 0 = host binned SAH (bvh_build.cpp), 1 = device LBVH, 2 = device PLOC (gpu_bvh.h); next upload
    int builder_in_use = 0;                  // what the last hr_upload_scene built with (0 | 1 | 2)
    double bvh_build_ms = 0;                 // device builder: key + sort + hierarchy + fit + emit + gather kernels
    uint64_t max_tail_bytes = 20ull << 30;   // cap of each hand-off buffer
    int seed_mode = 2;                       // 2 = three-run seed kernel (default), 1 = producer / consumer kernel with the state ring, 0 = fused seed kernel
    int seed_prof = 0;                       // phase timing build of the seed kernel (three-run kernel; ring kernel: splits 16 and 20)
    uint32_t seed_prio = 3;                  // s_setprio of the seed / round kernel's waves
    uint32_t nee_cull = 7;                   // debug option nee_cull: mask of nee_setup's shortcuts in force (1 far side | 2 GGX below the horizon; bit 2 reserved); 0 = trace every NEE shadow ray (bit-identical image, more rays)
    uint32_t rr_start = 0;                   // Russian roulette from this path iteration on (0 = off: the reference has none; NOT image-preserving)
    uint32_t ploc_top = hr::lbvh::PLOC_TOP_CLUSTERS;   // builder 2: clusters the bottom-up merges leave for the top-down build (1 = merge to the root)
    // the split pipeline (wf_kernels.h): queues of the launch being traced, sized for the largest launch so far
    int trace_mode = 0;                      // IN FORCE (resolve_modes): 0 = megakernel (trace_kernel), 1 = split: traversal kernel + shading kernel per path iteration
    int trace_mode_opt = -1;                 // debug option trace_mode: -1 = automatic (the split pipeline for precise shading of scenes with many triangles), 0 / 1 = pinned
    int precise_opt = -1;                    // option precise_shading: -1 = automatic (on for scenes without triangle meshes), 0 = off, 1 = on
    bool precise = false;                    // IN FORCE (resolve_modes) — option precise_shading: the bounce geometry in f64 (prec_core.h) — in the megakernel (path_advance<.., PREC>), or in the split pipeline's shading kernel with trace_mode 1
    bool wf_has_prec = false;                // the queues hold the residual quads
    WfQueues wf{};
    void *wf_block = nullptr;                // one allocation behind every pointer of wf
    uint32_t wf_adv_den = 2, wf_trav_wgs = 8, wf_shade_wgs = 8;   // debug: traversal kernel leaves its walk when 1/wf_adv_den of the lanes are done; workgroups per CU
    int debug_skip = 0;                      // timing experiments only (garbage image): 2 = skip the seed kernel, 4 = no ring stores, 8 = no ring fills, 16 = skip the trace kernel
    // timing (HIP events around every launch, summed when the streams are drained)
    std::vector<EventPair> seed_events, trace_events, post_events, debug_events;
    double seed_ms = 0, trace_ms = 0, post_ms = 0, debug_ms = 0;
    uint64_t seed_launches = 0, trace_launches = 0, debug_launches = 0;
    uint64_t paths_rendered = 0;
    // hr_mark / hr_wait: markers on the main stream, oldest first
    std::vector<std::pair<uint64_t, hipEvent_t>> markers;
    uint64_t next_ticket = 1;
};

// caller-owned accumulators (hr_bind_accumulator): one context per buffer.  accumulate_kernel adds a launch's radiance with plain loads and
// stores (no atomics since round 3), so two contexts accumulating into one buffer would race silently: a second binding is refused.
#include <map>
#include <mutex>
static std::mutex g_bound_mu;
static std::map<const void *, std::pair<const hr_ctx *, size_t>> g_bound;   // buffer -> (context, bytes)
static void unbind_accumulator(hr_ctx *c) {
    std::lock_guard<std::mutex> lk(g_bound_mu);
    for (auto it = g_bound.begin(); it != g_bound.end();) it = it->second.first == c ? g_bound.erase(it) : std::next(it);
}
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
// Priority governor.  All waves run at priority 0 except the seed kernel's consumer waves (3); which of the REST comes first decides which
// of the two kernels is the slower one.  Five levels, from "the seed kernel's producer waves first" to "the trace kernel first":
//   0  producers at init_prio (1), trace kernel at 0          3  producers at 0, the trace kernel's box phases at 1
//   1  producers alternate between init_prio and 0 per group   4  ... box and leaf phases at 1
//   2  producers at 0, trace kernel at 0
// The decision is taken ON THE DEVICE: a launch is enqueued many launches before it runs (hr_render never blocks), so a level put into
// its arguments by the host would be decided from measurements that are tens of launches old — or, inside one long hr_render call,
// never.  Both kernels stamp their first start and last end into GovDev (s_memrealtime), this one-thread kernel runs behind every trace
// kernel (in the gap in which the trace stream waits for the next seed kernel anyway), and kernels read GovDev::level when they start.
// A launch counts if its seed kernel ran beside the trace kernel of the launch before and its trace kernel beside the seed kernel of
// the launch after (the first and last launches of a burst do not).  An untried neighbouring level is tried when the balance asks for
// it (one kernel more than 1.5 % behind the other), otherwise the level with the best smoothed max(seed, trace) wins.
static const int GOV_LEVELS = 5;
__global__ void governor_kernel(GovDev *g, uint32_t slot) {
    typedef unsigned long long u64t;
    const uint32_t other = slot ^ 1u;
    const u64t none = ~0ull;
    const u64t s0 = g->t0[0][slot], s1 = g->t1[0][slot], r0 = g->t0[1][slot], r1 = g->t1[1][slot];
    const bool have = s0 != none && s1 > s0 && r0 != none && r1 > r0;
    if (have && g->fixed < 0) {
        const u64t p0 = g->prev_t0, p1 = g->prev_t1;
        // the next launch's seed kernel: running (its waves are updating these stamps with atomicMin / atomicMax right now; n1 not final) or done
        const u64t n0 = __hip_atomic_load(&g->t0[0][other], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), n1 = __hip_atomic_load(&g->t1[0][other], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float ov_seed = 0.0f, ov_trace = 0.0f;
        if (p1 > p0) {
            const u64t lo = p0 > s0 ? p0 : s0, hi = p1 < s1 ? p1 : s1;
            if (hi > lo) ov_seed = (float)(hi - lo) / (float)(s1 - s0);
        }
        if (n0 != none && n0 < r1) {
            const u64t lo = n0 > r0 ? n0 : r0, hi = (n1 > n0 && n1 < r1) ? n1 : r1;
            if (hi > lo) ov_trace = (float)(hi - lo) / (float)(r1 - r0);
        }
        const int L = (int)g->lvl[0][slot];
        const float seed_t = (float)(s1 - s0), trace_t = (float)(r1 - r0), m = seed_t > trace_t ? seed_t : trace_t;
        // (a trace kernel that takes much longer than a seed kernel can never have one beside it for 70 % of its time — rtcamp6_v2 / _v1: 48 and
        // 38 ms against 26 — and is the slower kernel beyond doubt: such launches count too.  Until round 4 they did not, and the governor sat
        // at level 0, the wrong end, on exactly the scenes where the trace kernel needs the slots: +2.5 % there.)
        const bool beside = ov_trace > 0.7f || trace_t > 1.25f * seed_t;
        // The wave budget comes first.  While the trace kernel is the faster kernel of the pair at level 0, workgroups it can do without are
        // taken away (one step per judged launch, down to budget_lo), and given back step by step as soon as it comes within 3 % of the
        // seed kernel; the priority levels only come into play with every workgroup in place.  Measured on the headline (1080p,
        // 256 CUs; trace / seed ms per launch): all 1,536 workgroups 19.4 / 25.3, 896 20.6 / 24.7, 768 22.0 / 24.4, 704 23.3 / 24.3,
        // 640 24.5 / 24.3 — the seed kernel gains what the trace kernel's waves no longer take, +3.5 % on the pair at 704 - 768.
        const uint32_t B = g->bud[slot] ? g->bud[slot] - 1u : ~0u;   // what every workgroup of the judged launch obeyed (stored + 1; unset: no trace kernel stamped it)
        // (a trace kernel far shorter than the seed kernel — the sphere scenes: 6 ms against 24 — never covers 70 % of a seed kernel, but that it
        // has workgroups to spare is beyond doubt: it is judged for the budget when it ran beside the next launch's seed kernel itself)
        const bool spare = trace_t < 0.7f * seed_t && ov_trace > 0.7f;
        if ((ov_seed > 0.7f || spare) && beside && L == (int)g->lvl[1][slot] && L >= 0 && L < GOV_LEVELS) {
            g->decisions++;
            bool budget_moved = false;
            if (L == 0 && g->level == 0 && B == g->budget && g->budget_step) {
                const uint32_t nb = gov_budget_next(B, trace_t / seed_t, g->budget_lo, g->budget_hi, g->budget_step, g->thr_down, g->thr_up);
                if (nb != B) {
                    budget_moved = true;
                    g->budget_moves++;
                    __hip_atomic_store(&g->budget, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int k = 0; k < GOV_LEVELS; k++) g->known[k] = 0;   // another balance: what the levels were worth is to be learnt again
                }
            }
            if (!budget_moved && B == 0 && g->budget == 0) {
                g->known[L] = g->known[L] > 0 ? 0.5f * (g->known[L] + m) : m;
                if (L == g->level) {   // (a launch that started before the last change of level: noted, nothing decided from it)
                    int next = L;
                    if (trace_t > 1.015f * seed_t && L < GOV_LEVELS - 1 && g->known[L + 1] == 0) next = L + 1;
                    else if (seed_t > 1.015f * trace_t && L > 0 && g->known[L - 1] == 0) next = L - 1;
                    else
                        for (int k = 0; k < GOV_LEVELS; k++)
                            if (g->known[k] > 0 && g->known[k] < 0.995f * g->known[next]) next = k;
                    if (next != L) { g->moves++; __hip_atomic_store(&g->level, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                }
            }
        }
    }
    if (r0 != none && r1 > r0) { g->prev_t0 = r0; g->prev_t1 = r1; }
    for (int k = 0; k < 2; k++) { g->t0[k][slot] = none; g->t1[k][slot] = 0; }
    g->bud[slot] = 0;   // unset: the next launch in this slot fixes its own
    if (g->fixed >= 0) __hip_atomic_store(&g->level, g->fixed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// What options precise_shading / trace_mode mean for the scene in place.  Precise shading has two homes that render the same bits
// (path_advance<.., PREC> in the megakernel at 128 VGPRs; the split pipeline's shading kernel), so which one runs is a question of speed only:
// the megakernel form costs 1.9 - 3.7 % on scenes without meshes (the trace side stays hidden behind the seed kernel), on every mesh scene the
// split form is the faster one (7 - 30 % below fp32 shading; profiles/r06_precise_pipelines.txt).  AUTOMATIC precise shading: on for scenes
// without triangle meshes — small spheres are what multiplies an fp32 ray's error, and there it costs little —, off where it costs.
static void resolve_modes(hr_ctx *c) {
    const bool has_scene = c->have_scene;
    const uint32_t tris = has_scene ? c->dsc.num_tris : 0u;
    c->precise = c->precise_opt == 1 || (c->precise_opt < 0 && has_scene && tris == 0u && !c->rr_start);
    c->trace_mode = c->trace_mode_opt >= 0 ? c->trace_mode_opt : (c->precise && tris > 0u ? 1 : 0);
}
// a new scene, resolution or option: the balance of the two kernels is another one.  The governor starts at level 0 — next to a trace
// kernel that needs 16 ms per 33 M paths on the reference's scenes the seed kernel (24 ms) is the slower one almost everywhere.
// (Callers have synchronised the context: no kernel is stamping.)
static int govern_reset(hr_ctx *c) {
    resolve_modes(c);
    if (!c->gov) return HR_OK;
    GovDev h;
    memset(&h, 0, sizeof h);
    for (int k = 0; k < 2; k++)
        for (int sl = 0; sl < 2; sl++) h.t0[k][sl] = ~0ull;
    h.fixed = c->trace_boost;
    h.level = c->trace_boost >= 0 ? c->trace_boost : 0;
    // the wave budget is governed with the level (a fixed level pins it at "all"): 2.5 .. 3.5 workgroups per CU in steps of a quarter
    if (c->trace_boost < 0) { h.budget_step = (uint32_t)c->num_cus / 4u; h.budget_lo = (uint32_t)c->num_cus * 5u / 2u; h.budget_hi = (uint32_t)c->num_cus * 7u / 2u; }
    h.thr_down = 0.88f; h.thr_up = 0.97f;
    if (c->trace_mode == 1 && c->trace_boost < 0) {
        // the split pipeline's traversal kernel: 3 .. 7 workgroups of four 64-VGPR waves per CU in steps of a half, "all" = 8 (device_scene.h gov_budget_next)
        h.budget_step = (uint32_t)c->num_cus / 2u; h.budget_lo = (uint32_t)c->num_cus * 3u; h.budget_hi = (uint32_t)c->num_cus * 7u;
        h.thr_down = 0.96f; h.thr_up = 1.02f;
    }
    HIP_TRY(hipMemcpy(c->gov, &h, sizeof h, hipMemcpyHostToDevice));
    return HR_OK;
}
// the accumulator of `c` is about to change: totals that include it are stale — its own and, in a same-device group, its peers'
static void invalidate_totals(hr_ctx *c) {
    c->total_valid = false;
    for (hr_ctx *p : c->same_device_peers) p->total_valid = false;
}
static int drain_events(hr_ctx *c) {
    auto sum = [](std::vector<EventPair> &ev, double &acc) -> hipError_t {
        // a pair whose query fails is dropped with the rest (left in the list it would fail every later drain, i.e. every later API call)
        hipError_t first = hipSuccess;
        for (auto &e : ev) {
            float ms = 0;
            hipError_t r = hipEventElapsedTime(&ms, e.a, e.b);
            if (r == hipSuccess) acc += ms;
            else if (first == hipSuccess) first = r;
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        ev.clear();
        return first;
    };
    HIP_TRY(sum(c->seed_events, c->seed_ms));
    HIP_TRY(sum(c->trace_events, c->trace_ms));
    HIP_TRY(sum(c->post_events, c->post_ms));
    HIP_TRY(sum(c->debug_events, c->debug_ms));
    return HR_OK;
}
// Long renders: retire the event pairs of launches that have finished (both kernels), oldest first, without waiting for anything —
// the lists stay a few launches long however many samplings one hr_render call covers, and no drain ever has to stop the pipeline.
static void retire_finished_launches(hr_ctx *c) {
    size_t n = 0;
    const size_t limit = std::min(c->seed_events.size(), c->trace_events.size());
    while (n < limit && hipEventQuery(c->seed_events[n].b) == hipSuccess && hipEventQuery(c->trace_events[n].b) == hipSuccess) n++;
    if (n > 2) n -= 2; else return;
    for (size_t i = 0; i < n; i++) {
        float sm = 0, tm = 0;
        if (hipEventElapsedTime(&sm, c->seed_events[i].a, c->seed_events[i].b) == hipSuccess) c->seed_ms += sm;
        if (hipEventElapsedTime(&tm, c->trace_events[i].a, c->trace_events[i].b) == hipSuccess) c->trace_ms += tm;
        (void)hipEventDestroy(c->seed_events[i].a); (void)hipEventDestroy(c->seed_events[i].b);
        (void)hipEventDestroy(c->trace_events[i].a); (void)hipEventDestroy(c->trace_events[i].b);
    }
    c->seed_events.erase(c->seed_events.begin(), c->seed_events.begin() + (long)n);
    c->trace_events.erase(c->trace_events.begin(), c->trace_events.begin() + (long)n);
}
static int sync_all(hr_ctx *c) {
    HIP_TRY(hipStreamSynchronize(c->seed_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->trace_pending[0] = c->trace_pending[1] = false;
    c->seed_pending[0] = c->seed_pending[1] = false;
    for (auto &m : c->markers) (void)hipEventDestroy(m.second);
    c->markers.clear();
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
    HIP_TRY(hipMalloc((void **)&c->gov, sizeof(GovDev)));
    { int grc = govern_reset(c); if (grc) return grc; }
    HIP_TRY(hipFuncSetAttribute((const void *)seed_isaac64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<28>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_pc_kernel<20, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_seg_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_seg_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
#if defined(HR_EXPERIMENTS)
    HIP_TRY(hipFuncSetAttribute((const void *)seed_w5_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_w5_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_ps_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_ps_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_ps_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)seed_ps_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES));
#endif
    HIP_TRY(hipFuncSetAttribute((const void *)seed_debug_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 8));
    // The seed kernel owns all 160 KiB of a CU's LDS and runs next to the trace kernel of the previous batch: a trace kernel that
    // uses ANY LDS (the compiler promotes small private arrays to LDS unless told not to, see the Makefile) could not share a CU
    // with it — the two would silently run one after the other, 40 % slower.  Refuse to start in that state.
    {
        const void *trace_variants[] = {(const void *)trace_kernel<false, 5, true>, (const void *)trace_kernel<false, 5, false>, (const void *)trace_kernel<false, 4, true>,
                                        (const void *)trace_kernel<false, 6, true>, (const void *)trace_kernel<true, 3, true>, (const void *)trace_kernel<true, 3, false>,
                                        (const void *)trace_kernel<false, 5, true, true>, (const void *)trace_kernel<false, 5, false, true>,
                                        (const void *)trace_kernel<true, 3, true, true>, (const void *)trace_kernel<true, 3, false, true>,
                                        (const void *)trace_kernel<false, 3, true, false, true>, (const void *)trace_kernel<false, 3, false, false, true>,
                                        (const void *)trace_kernel<false, 4, true, false, false, true>, (const void *)trace_kernel<false, 4, false, false, false, true>};
        for (const void *f : trace_variants) {
            hipFuncAttributes fa;
            HIP_TRY(hipFuncGetAttributes(&fa, f));
            if (fa.sharedSizeBytes != 0) return fail(HR_ERR_DEVICE, "build error: a trace kernel variant uses %zu bytes of LDS (it must use none to run beside the seed kernel)", (size_t)fa.sharedSizeBytes);
        }
    }
    HIP_TRY(hipMalloc((void **)&c->ovf_win, (size_t)c->num_cus * 2 * SEED_WIN_WORDS * sizeof(u64)));
    return HR_OK;
}

int hr_destroy(hr_ctx *c) {
    if (!c) return HR_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    unbind_accumulator(c);
    free_scene(c);
    for (auto *ev : {&c->seed_events, &c->trace_events, &c->post_events, &c->debug_events})
        for (auto &e : *ev) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &m : c->markers) (void)hipEventDestroy(m.second);
    if (c->accum_own) (void)hipFree(c->accum_own);
    for (int i = 0; i < 2; i++) {
        if (c->recs[i]) (void)hipFree(c->recs[i]);
        if (c->seed_done[i]) (void)hipEventDestroy(c->seed_done[i]);
        if (c->trace_done[i]) (void)hipEventDestroy(c->trace_done[i]);
    }
    if (c->ring) (void)hipFree(c->ring);
    if (c->wf_block) (void)hipFree(c->wf_block);
    if (c->ovf) (void)hipFree(c->ovf);
    if (c->ovf_win) (void)hipFree(c->ovf_win);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_tile_counter) (void)hipFree(c->d_tile_counter);
    if (c->gov) (void)hipFree(c->gov);
    if (c->post_tmp) (void)hipFree(c->post_tmp);
    if (c->d_rgb8) (void)hipFree(c->d_rgb8);
    if (c->accum_total) (void)hipFree(c->accum_total);
    if (c->comm && hrcomm::api().CommDestroy) (void)hrcomm::api().CommDestroy(c->comm);
    for (hr_ctx *p : c->same_device_peers) if (p != c) { p->same_device_peers.clear(); p->comm_world = 0; p->total_valid = false; p->comm_path = HR_COMM_NONE; p->allreduces = 0; }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->seed_stream) (void)hipStreamDestroy(c->seed_stream);
    delete c;
    return HR_OK;
}

// Device builders (option bvh_builder = 1 LBVH, 2 PLOC; gpu_bvh.h): the primitive arrays were uploaded in input order; split long
// thin triangles into references (early split clipping), build the tree over them, emit it in both record formats (16-byte quantised
// records in per-octant near-first preorder — what the trace kernel walks — and the 32-byte fp32 records), re-store the primitives
// in leaf order and point the scene at the results.  Scratch is freed before returning.
static int build_bvh_on_device(hr_ctx *c, const HostScene &hs, const Tri *tris_in) {
    using namespace lbvh;
    Scene &d = c->dsc;
    Prims p{};
    p.tris = tris_in; p.num_tris = d.num_tris; p.spheres = d.spheres; p.num_spheres = d.num_spheres; p.cuboids = d.cuboids; p.num_cuboids = d.num_cuboids;
    p.ref_tri = nullptr; p.ref_box = nullptr;
    double scene_sa = 0.0;
    for (int a = 0; a < 3; a++) {
        double ext = hs.scene_max[a] - hs.scene_min[a];
        p.smin[a] = (float)hs.scene_min[a];
        p.sinv[a] = ext > 0 ? (float)(1.0 / ext) : 0.0f;
    }
    {
        const double e0 = hs.scene_max[0] - hs.scene_min[0], e1 = hs.scene_max[1] - hs.scene_min[1], e2 = hs.scene_max[2] - hs.scene_min[2];
        if (e0 >= 0 && e1 >= 0 && e2 >= 0) scene_sa = 2.0 * (e0 * e1 + e1 * e2 + e2 * e0);
    }
    std::vector<void *> scratch;
    hipEvent_t ea = nullptr, eb = nullptr;   // around everything the build puts on the stream
    auto cleanup = [&]() {
        for (void *q : scratch) (void)hipFree(q);
        scratch.clear();
        if (ea) { (void)hipEventDestroy(ea); ea = nullptr; }
        if (eb) { (void)hipEventDestroy(eb); eb = nullptr; }
    };
    auto alloc = [&](size_t bytes, bool keep) -> void * {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
        (keep ? c->scene_allocs : scratch).push_back(q);
        return q;
    };
#define LBVH_ALLOC(var, type, count, keep)                                                                        \
    type *var = (type *)alloc(sizeof(type) * (size_t)(count), keep);                                                \
    if (!var) { cleanup(); return fail(HR_ERR_DEVICE, "hr_upload_scene: out of device memory in the BVH build"); }
    (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    (void)hipEventRecord(ea, c->stream);
    // Early split clipping on the device (option split_ratio: -1 = on with the host builder's automatic ratio of 2, 0 = off, > 0 = that
    // ratio; the host builder's automatic mode also builds the unsplit tree and keeps the better one, the device always keeps the split):
    // pieces per triangle, a scan, then the pieces' boxes and owners.  The builders below then see one primitive per piece.
    if (c->split_ratio != 0.0 && d.num_tris > 0) {
        SplitParams sp{c->split_ratio < 0 ? 2.0 : c->split_ratio, 1e-4 * scene_sa, SPLIT_MAX_DEPTH};
        const uint32_t nt = d.num_tris;
        LBVH_ALLOC(split_counts, uint32_t, nt, false)
        LBVH_ALLOC(split_offsets, uint32_t, nt, false)
        size_t sbytes = 0;
        hipError_t se = hipcub::DeviceScan::ExclusiveSum(nullptr, sbytes, split_counts, split_offsets, (int)nt, c->stream);
        LBVH_ALLOC(split_tmp, unsigned char, sbytes, false)
        split_count_kernel<<<(nt + 127) / 128, 128, 0, c->stream>>>(tris_in, nt, sp, split_counts);
        if (se == hipSuccess) se = hipcub::DeviceScan::ExclusiveSum(split_tmp, sbytes, split_counts, split_offsets, (int)nt, c->stream);
        uint32_t last[2] = {0, 0};
        if (se == hipSuccess) se = hipMemcpyAsync(&last[0], split_offsets + (nt - 1), 4, hipMemcpyDeviceToHost, c->stream);
        if (se == hipSuccess) se = hipMemcpyAsync(&last[1], split_counts + (nt - 1), 4, hipMemcpyDeviceToHost, c->stream);
        if (se == hipSuccess) se = hipStreamSynchronize(c->stream);
        if (se != hipSuccess) { cleanup(); return fail(HR_ERR_DEVICE, "device split clipping: %s", hipGetErrorString(se)); }
        const uint64_t refs = (uint64_t)last[0] + last[1];
        // (bounded together with the spheres and cuboids: the builder's n = refs + spheres + cuboids indexes its sort keys and INFO_COUNT with
        // 24 bits; beyond that the split references are dropped and the triangles go in as they are)
        if (refs > nt && refs + p.num_spheres + p.num_cuboids < MAX_PRIMS_PER_TYPE) {
            LBVH_ALLOC(ref_tri, uint32_t, refs, false)
            LBVH_ALLOC(ref_box, float, 6 * refs, false)
            split_emit_kernel<<<(nt + 127) / 128, 128, 0, c->stream>>>(tris_in, nt, sp, split_offsets, ref_tri, ref_box);
            p.ref_tri = ref_tri; p.ref_box = ref_box; p.num_tris = (uint32_t)refs;
        }
    }
    const int n = (int)(p.num_tris + p.num_spheres + p.num_cuboids);
    p.index_bits = key_index_bits_for((uint64_t)n);
    if (n <= 0 || (uint64_t)n >= (1ull << p.index_bits)) { cleanup(); return fail(HR_ERR_UNSUPPORTED, "device BVH build: %d primitives do not fit the %d index bits of the sort keys", n, p.index_bits); }
    const int N = 2 * n - 1;
    LBVH_ALLOC(keys_in, mkey_t, n, false)
    LBVH_ALLOC(keys, mkey_t, n, false)
    Work w{};
    LBVH_ALLOC(parent, uint32_t, N, false) LBVH_ALLOC(left, uint32_t, n, false) LBVH_ALLOC(right, uint32_t, n, false)
    LBVH_ALLOC(flags, uint32_t, n, false)
    LBVH_ALLOC(bmin, float, 3 * (size_t)N, false) LBVH_ALLOC(bmax, float, 3 * (size_t)N, false)
    LBVH_ALLOC(info, uint32_t, N, false) LBVH_ALLOC(tc, u64t, N, false) LBVH_ALLOC(size, uint32_t, N, false)
    LBVH_ALLOC(word, uint32_t, N, false) LBVH_ALLOC(axis_low, uint32_t, n, false)
    LBVH_ALLOC(prim_pos, uint32_t, n, false) LBVH_ALLOC(cl_a, uint32_t, n, false) LBVH_ALLOC(cl_b, uint32_t, n, false) LBVH_ALLOC(nn, uint32_t, n, false)
    LBVH_ALLOC(frame, float, 8, false)
    w.parent = parent; w.left = left; w.right = right; w.flags = flags;
    w.bmin = bmin; w.bmax = bmax; w.info = info; w.tc = tc; w.size = size; w.axis_low = axis_low; w.word = word;
    // the emitted tree has size[root] <= 2n-1 records per octant (collapsed subtrees are one record): sized for the worst case
    LBVH_ALLOC(nodes, Node, 8 * (size_t)N + 1, true)
    LBVH_ALLOC(qnodes, QNode, 8 * ((size_t)N + 1), true)
    LBVH_ALLOC(tris, TriT, p.num_tris, true)
    LBVH_ALLOC(tri_shade, TriS, p.num_tris, true)
    LBVH_ALLOC(tri_face, uint32_t, p.num_tris, true)
    LBVH_ALLOC(spheres, f4, d.num_spheres, true)
    LBVH_ALLOC(sphere_elem, int32_t, d.num_spheres, true)
    LBVH_ALLOC(sphere_lo, f4, d.num_spheres, true)
    LBVH_ALLOC(cuboids, f4, 2 * (size_t)d.num_cuboids, true)
    size_t sort_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys_in, keys, n, 0, 64, c->stream);
    if (e != hipSuccess) { cleanup(); return fail(HR_ERR_DEVICE, "hipcub sort (size query): %s", hipGetErrorString(e)); }
    LBVH_ALLOC(sort_tmp, unsigned char, sort_bytes, false)
    // multi-workgroup PLOC: packed role counters, their scan, the two-slot iteration state
    const bool ploc_multi = c->builder_in_use == 2 && n > 1;
    size_t scan_bytes = 0;
    if (ploc_multi) {
        e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (u64t *)nullptr, (u64t *)nullptr, n, c->stream);
        if (e != hipSuccess) { cleanup(); return fail(HR_ERR_DEVICE, "hipcub scan (size query): %s", hipGetErrorString(e)); }
    }
    LBVH_ALLOC(ploc_flags, u64t, ploc_multi ? n : 1, false)
    LBVH_ALLOC(ploc_pos, u64t, ploc_multi ? n : 1, false)
    LBVH_ALLOC(scan_tmp, unsigned char, scan_bytes, false)
    LBVH_ALLOC(ploc_state, PlocState, 2, false)
    const uint32_t top_cap = ploc_multi ? std::min<uint32_t>((uint32_t)n, c->ploc_top) : 1u;   // clusters the top-down build may be handed
    LBVH_ALLOC(top_boxes, float, 6 * (size_t)top_cap, false)
    LBVH_ALLOC(top_counts, uint32_t, top_cap, false)
    LBVH_ALLOC(top_left, int32_t, top_cap, false)
    LBVH_ALLOC(top_right, int32_t, top_cap, false)
#undef LBVH_ALLOC
    const int T = 256;
    hipStream_t st = c->stream;
    e = hipMemsetAsync(flags, 0, sizeof(uint32_t) * (size_t)n, st);
    if (e == hipSuccess) e = hipMemsetAsync(parent, 0xff, sizeof(uint32_t) * (size_t)N, st);   // n == 1: the lone leaf is the root
    if (e == hipSuccess) {
        key_kernel<<<(n + T - 1) / T, T, 0, st>>>(p, n, keys_in);
        e = hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, keys_in, keys, n, 0, 64, st);
    }
    if (e == hipSuccess) {
        leaf_kernel<<<(n + T - 1) / T, T, 0, st>>>(p, keys, n, w);
        if (n > 1 && c->builder_in_use == 1) hierarchy_kernel<<<(n - 1 + T - 1) / T, T, 0, st>>>(keys, n, w);
        if (ploc_multi) {
            ploc_init_kernel<<<(n + T - 1) / T, T, 0, st>>>(n, cl_a, ploc_state);
            uint32_t *cur = cl_a, *nxt = cl_b;
            uint32_t m_known = (uint32_t)n;   // the host's upper bound of the cluster count (refreshed every few iterations)
            int it = 0;
            for (; it < 4096 && m_known > c->ploc_top && e == hipSuccess; it++) {
                const uint32_t g = (m_known + T - 1) / T;
                const PlocState *sin = ploc_state + (it & 1);
                ploc_nn_kernel<<<g, T, 0, st>>>(w, cur, nn, sin);
                ploc_role_kernel<<<g, T, 0, st>>>(nn, ploc_flags, m_known, sin);
                e = hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, ploc_flags, ploc_pos, (int)m_known, st);
                ploc_merge_kernel<<<g, T, 0, st>>>(w, cur, nxt, nn, ploc_flags, ploc_pos, sin, ploc_state + ((it + 1) & 1));
                std::swap(cur, nxt);
                if ((it & 3) == 3 || m_known <= 4u * c->ploc_top) {   // every fourth iteration (every one near the end): how many are left?
                    PlocState hs{};
                    if (e == hipSuccess) e = hipMemcpyAsync(&hs, ploc_state + ((it + 1) & 1), sizeof hs, hipMemcpyDeviceToHost, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);
                    if (e == hipSuccess) m_known = hs.m;
                }
            }
            if (e == hipSuccess && m_known > top_cap) e = hipErrorUnknown;   // (the loop ends at <= ploc_top clusters, the size of the top buffers)
            // the top of the tree: binned SAH over the clusters that are left, on the host (a few thousand boxes)
            if (e == hipSuccess && m_known > 1u) {
                const uint32_t m = m_known;
                std::vector<float> hb(6 * (size_t)m);
                std::vector<uint32_t> hc(m);
                ploc_top_gather_kernel<<<(m + T - 1) / T, T, 0, st>>>(w, cur, m, top_boxes, top_counts);
                e = hipMemcpyAsync(hb.data(), top_boxes, hb.size() * sizeof(float), hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipMemcpyAsync(hc.data(), top_counts, m * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                std::vector<int32_t> tl, tr;
                if (e == hipSuccess) {
                    build_top_tree(hb.data(), hc.data(), m, tl, tr);
                    if (tl.size() != (size_t)m - 1) e = hipErrorUnknown;
                }
                if (e == hipSuccess) e = hipMemcpyAsync(top_left, tl.data(), tl.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipMemcpyAsync(top_right, tr.data(), tr.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
                if (e == hipSuccess) {
                    ploc_top_apply_kernel<<<(m - 1 + T - 1) / T, T, 0, st>>>(w, m - 1, top_left, top_right, cur);
                    e = hipStreamSynchronize(st);   // tl / tr are host vectors about to go out of scope
                }
            }
        }
        fit_kernel<<<(n + T - 1) / T, T, 0, st>>>(n, (uint32_t)c->max_leaf, w);
        finish_kernel<<<(N + T - 1) / T, T, 0, st>>>(p, n, w, prim_pos);
        frame_kernel<<<1, 64, 0, st>>>(w, frame);
        emit_kernel<<<(8 * N + T - 1) / T, T, 0, st>>>(n, w, frame, nodes, qnodes);
        gather_kernel<<<(n + T - 1) / T, T, 0, st>>>(p, keys, prim_pos, n, tris, tri_shade, tri_face, spheres, sphere_elem, d.sphere_elem, sphere_lo, d.sphere_lo, cuboids);
        e = hipGetLastError();
    }
    (void)hipEventRecord(eb, st);
    float hframe[8] = {0, 0, 0, 1, 1, 1, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(hframe, frame, sizeof hframe, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    float ms = 0;
    if (e == hipSuccess) (void)hipEventElapsedTime(&ms, ea, eb);
    cleanup();
    if (e != hipSuccess) return fail(HR_ERR_DEVICE, "device BVH build: %s", hipGetErrorString(e));
    uint32_t total = 0;
    memcpy(&total, &hframe[6], sizeof total);
    if (total == 0 || total > (uint32_t)N) return fail(HR_ERR_DEVICE, "device BVH build: implausible record count %u for %d primitives", total, n);
    if (c->quant_nodes && (uint64_t)(total + 1u) * 8u * sizeof(QNode) >= (1ull << 31)) return fail(HR_ERR_UNSUPPORTED, "device BVH build: %u records per octant exceed the 2^31-byte offset range of the quantised records (set quant_nodes = 0)", total);
    c->bvh_build_ms = ms;
    d.nodes = nodes; d.num_nodes = total;
    d.qnodes = c->quant_nodes ? qnodes : nullptr;
    for (int a = 0; a < 3; a++) { d.qmin[a] = hframe[a]; d.qstep[a] = hframe[3 + a]; }

    d.tris = tris; d.tri_shade = tri_shade; d.tri_face = tri_face; d.spheres = spheres; d.sphere_elem = sphere_elem; d.sphere_lo = sphere_lo; d.cuboids = cuboids;   // the input-order copies stay in scene_allocs until the next upload
    d.num_tris = p.num_tris;   // leaf-ordered records: one per reference (a split triangle appears once per piece)
    return HR_OK;
}

int hr_upload_scene(hr_ctx *c, const hr_scene_desc *sd) {
    if (!c || !sd) return fail(HR_ERR_INVALID, "hr_upload_scene: null argument");
    if (!sd->elements || sd->num_elements == 0) return fail(HR_ERR_INVALID, "hr_upload_scene: scene has no elements");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;

    HostScene hs;
    std::string ferr;
    // Which builder: the host's binned-SAH build with split clipping gives the best tree (1 - 3 % fewer node tests than the device PLOC
    // build) but is one thread — 12 k triangles take 20 ms, 10^6 six seconds, 4 x 10^6 twenty-five — while the device builds 4 x 10^6 in
    // 38 ms.  By default the scene's size decides: below AUTO_BUILDER_PRIMS primitives (host build < 1 s) the host tree, above it PLOC.
    static const uint64_t AUTO_BUILDER_PRIMS = 200000;
    int builder = c->bvh_builder;
    if (builder < 0) {
        uint64_t prims = 0;
        for (uint32_t e = 0; e < sd->num_elements; e++) prims += sd->elements[e].kind == HR_MESH ? sd->elements[e].num_faces : 1;
        builder = prims >= AUTO_BUILDER_PRIMS ? 2 : 0;
    }
    c->builder_in_use = builder;
    const bool gpu_build = builder != 0;
    rc = flatten_scene(sd, hs, ferr, c->max_leaf, gpu_build ? 0.0 : c->split_ratio, !gpu_build);
    if (rc) return fail(rc, "hr_upload_scene: %s", ferr.c_str());   // a description that is refused leaves the scene in place
    free_scene(c);
    Scene &d = c->dsc;
    d = hs.view();
    int r;
    const Tri *tris_in = nullptr;   // the geometry records: input of the device builders, or (host-built tree, leaf order) of the derivation below
    if ((r = upload(c, hs.tris, &tris_in))) return r;
    if ((r = upload(c, hs.spheres, &d.spheres))) return r;
    if ((r = upload(c, hs.sphere_elem, &d.sphere_elem))) return r;
    if ((r = upload(c, hs.sphere_lo, &d.sphere_lo))) return r;
    if ((r = upload(c, hs.cuboids, &d.cuboids))) return r;
    if ((r = upload(c, hs.cuboid_lo, &d.cuboid_lo))) return r;
    if ((r = upload(c, hs.tri_exact, &d.tri_exact))) return r;
    if ((r = upload(c, hs.materials, &d.materials))) return r;
    if ((r = upload(c, hs.images, &d.images))) return r;
    if ((r = upload(c, hs.emitters, &d.emitters))) return r;
    if ((r = upload(c, hs.texels, &d.texels))) return r;
    if ((r = upload(c, std::vector<CameraD>(1, hs.camd), &d.camd))) return r;
    if (!hs.sky_quads.empty()) { if ((r = upload(c, hs.sky_quads, &d.sky_quads))) return r; }
    else d.sky_quads = nullptr;
    c->bvh_build_ms = 0;
    if (gpu_build) { if ((r = build_bvh_on_device(c, hs, tris_in))) return r; }
    else {
        // the records the kernels read for a triangle, derived on the device (as the device builders' gather does)
        TriT *tt = nullptr; TriS *tsh = nullptr;
        const size_t nt = hs.tris.size();
        HIP_TRY(hipMalloc((void **)&tt, std::max<size_t>(nt * sizeof(TriT), 16)));
        c->scene_allocs.push_back(tt);
        HIP_TRY(hipMalloc((void **)&tsh, std::max<size_t>(nt * sizeof(TriS), 16)));
        c->scene_allocs.push_back(tsh);
        uint32_t *tfc = nullptr;
        HIP_TRY(hipMalloc((void **)&tfc, std::max<size_t>(nt * sizeof(uint32_t), 16)));
        c->scene_allocs.push_back(tfc);
        if (nt) {
            lbvh::tri_derive_kernel<<<(unsigned)((nt + 255) / 256), 256, 0, c->stream>>>(tris_in, (uint32_t)nt, tt, tsh, tfc);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        d.tris = tt; d.tri_shade = tsh; d.tri_face = tfc;
        if ((r = upload(c, hs.nodes, &d.nodes))) return r;
        d.qnodes = nullptr;
        if (c->quant_nodes && hs.qnodes.size() * sizeof(QNode) >= (1ull << 31)) return fail(HR_ERR_UNSUPPORTED, "hr_upload_scene: the quantised BVH records exceed their 2^31-byte offset range (set quant_nodes = 0)");
        if (c->quant_nodes && !hs.qnodes.empty() && (r = upload(c, hs.qnodes, &d.qnodes))) return r;
    }
    c->st_nodes = d.num_nodes; c->st_tris = hs.num_input_tris; c->st_spheres = d.num_spheres; c->st_cuboids = d.num_cuboids;
    c->have_scene = true;
    if ((r = govern_reset(c))) return r;   // another scene: the balance of the two kernels is another one
    return HR_OK;
}

int hr_set_resolution(hr_ctx *c, uint32_t w, uint32_t h) {
    if (!c || !w || !h) return fail(HR_ERR_INVALID, "hr_set_resolution: bad argument");
    if ((uint64_t)w * h > (1ull << 27)) return fail(HR_ERR_UNSUPPORTED, "resolution too large");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    // no target while the buffers are being replaced (a failed allocation leaves the context without one, not with dangling
    // pointers); a caller-bound accumulator was sized for the old resolution: it is unbound, the caller rebinds
    c->accum = nullptr; c->W = c->H = 0; c->total_valid = false;
    unbind_accumulator(c);
    if (c->accum_own) { HIP_TRY(hipFree(c->accum_own)); c->accum_own = nullptr; }
    if (c->post_tmp) { HIP_TRY(hipFree(c->post_tmp)); c->post_tmp = nullptr; }
    if (c->d_rgb8) { HIP_TRY(hipFree(c->d_rgb8)); c->d_rgb8 = nullptr; }
    if (c->accum_total) { HIP_TRY(hipFree(c->accum_total)); c->accum_total = nullptr; }
    size_t n = (size_t)w * h * 3;
    HIP_TRY(hipMalloc((void **)&c->accum_own, n * sizeof(float)));
    HIP_TRY(hipMemset(c->accum_own, 0, n * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&c->post_tmp, n * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&c->d_rgb8, n));
    c->W = w; c->H = h;
    c->accum = c->accum_own;
    return govern_reset(c);
}

int hr_bind_accumulator(hr_ctx *c, float *device_rgb) {
    if (!c) return fail(HR_ERR_INVALID, "hr_bind_accumulator: null ctx");
    if (!c->W) return fail(HR_ERR_NO_TARGET, "hr_bind_accumulator: hr_set_resolution not called");
    HIP_TRY(hipSetDevice(c->device));
    if (device_rgb) {
        // what can be checked of a caller's pointer is checked: device memory, of this context's device, float-aligned, and W x H x 3 floats
        // inside the allocation it points into (a tensor of another shape or dtype would otherwise be overrun by plain stores, silently)
        const size_t need = (size_t)c->W * c->H * 3 * sizeof(float);
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, device_rgb) != hipSuccess || at.type != hipMemoryTypeDevice) {
            (void)hipGetLastError();
            return fail(HR_ERR_INVALID, "hr_bind_accumulator: %p is not device memory", (void *)device_rgb);
        }
        if (at.device != c->device) return fail(HR_ERR_INVALID, "hr_bind_accumulator: the buffer lives on device %d, the context on device %d", at.device, c->device);
        if ((uintptr_t)device_rgb % sizeof(float)) return fail(HR_ERR_INVALID, "hr_bind_accumulator: the buffer is not aligned for floats");
        void *base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)device_rgb) == hipSuccess) {
            if ((const char *)device_rgb + need > (const char *)base + size)
                return fail(HR_ERR_INVALID, "hr_bind_accumulator: the buffer is too small (%zu bytes from this address to the end of its allocation, %u x %u x 3 floats = %zu needed)",
                            (size_t)((const char *)base + size - (const char *)device_rgb), c->W, c->H, need);
        } else (void)hipGetLastError();
    }
    int rc = sync_all(c);
    if (rc) return rc;
    {
        // look-up, release of this context's old binding and the new entry under ONE lock: two threads binding one buffer to two contexts
        // cannot both pass.  The registry holds byte ranges: a buffer that overlaps another context's is refused like an equal one.
        // (The caller binds NULL before it frees a bound buffer: an entry left behind would refuse whoever is handed the address next.)
        std::lock_guard<std::mutex> lk(g_bound_mu);
        const size_t bytes = (size_t)c->W * c->H * 3 * sizeof(float);
        if (device_rgb)
            for (const auto &kv : g_bound) {
                const char *a = (const char *)kv.first, *b = (const char *)device_rgb;
                if (kv.second.first != c && a < b + bytes && b < a + kv.second.second)
                    return fail(HR_ERR_INVALID, "hr_bind_accumulator: this buffer is already bound to another context (one context per accumulator: the launch's radiance is added with plain loads and stores)");
            }
        for (auto it = g_bound.begin(); it != g_bound.end();) it = it->second.first == c ? g_bound.erase(it) : std::next(it);
        if (device_rgb) g_bound[device_rgb] = std::make_pair((const hr_ctx *)c, bytes);
    }
    c->accum = device_rgb ? device_rgb : c->accum_own;
    invalidate_totals(c);
    return HR_OK;
}
void *hr_accumulator_device_ptr(hr_ctx *c) { return c ? c->accum : nullptr; }

int hr_set_stream(hr_ctx *c, void *s) {
    if (!c) return fail(HR_ERR_INVALID, "hr_set_stream: null ctx");
    HIP_TRY(hipSetDevice(c->device));
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
    invalidate_totals(c);
    HIP_TRY(hipMemsetAsync(c->accum, 0, (size_t)c->W * c->H * 3 * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(Counters), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->seed_ms = c->trace_ms = c->post_ms = c->debug_ms = 0;
    c->seed_launches = c->trace_launches = c->debug_launches = 0;
    c->paths_rendered = 0;
    return HR_OK;
}

// Precise shading computes with the reference's f64 draws: the seed kernel (the default one, seed_mode 2) writes what rounding a draw to fp32
// took away into the records' twin behind the records (device_scene.h RenderParams::rec_lo_off).
static bool draws_twin(const hr_ctx *c) { return c->precise && c->seed_mode == 2 && c->draw_residuals; }
static uint64_t rec_lo_off(const hr_ctx *c) { return draws_twin(c) ? c->rec_lo_off : 0; }
static int ensure_draws(hr_ctx *c, size_t items) {
    const bool twin = draws_twin(c);
    if (items <= c->draws_cap && (!twin || c->rec_lo_off)) return HR_OK;
    int rc = sync_all(c);   // kernels of an earlier hr_render may still be reading the buffers that are about to be replaced
    if (rc) return rc;
    items = std::max(items, c->draws_cap);
    c->draws_cap = 0;       // stays 0 if an allocation below fails: the next call starts over
    c->rec_lo_off = 0;
    const size_t floats = (items + SEED_SPARE_ITEMS) * REC_ITEM_FLOATS;   // + spare items for the padding lanes of the last group
    for (int i = 0; i < 2; i++) {
        if (c->recs[i]) { HIP_TRY(hipFree(c->recs[i])); c->recs[i] = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->recs[i], floats * (twin ? 2 : 1) * sizeof(float)));
    }
    c->draws_cap = items;
    c->rec_lo_off = twin ? floats : 0;
    return HR_OK;
}

// Fix-up lists of the seed kernel's consumer waves: a path whose lens rejection loop rejects its first LENS_FAST attempts (round
// lens: (1 - pi/4)^5 = 4.6e-4 of the paths) is queued by the wave that seeded it.  The paths per wave grow with the launch and
// shrink with the CU count, so the lists are sized per launch: 8 x the expected count (the count is Poisson: 8 x is > 30 sigma away
// for any launch that matters), never below SEED_OVF_MIN.
static int ensure_ovf(hr_ctx *c, uint64_t paths_per_launch) {
    const uint64_t groups = (paths_per_launch + SEED_COLS - 1) / SEED_COLS;
    const uint64_t waves = 2 * std::min<uint64_t>(std::max<uint64_t>(groups, 1), (uint64_t)c->num_cus);
    const double expected = 4.7e-4 * (double)paths_per_launch / (double)waves;
    const uint64_t want = std::max<uint64_t>(SEED_OVF_MIN, ((uint64_t)(8.0 * expected) + 64 + 255) / 256 * 256);
    if (want <= c->ovf_cap) return HR_OK;
    int rc = sync_all(c);   // a seed kernel in flight may still be writing its lists
    if (rc) return rc;
    if (c->ovf) { HIP_TRY(hipFree(c->ovf)); c->ovf = nullptr; }
    c->ovf_cap = 0;
    HIP_TRY(hipMalloc((void **)&c->ovf, (size_t)c->num_cus * 2 * want * sizeof(uint32_t)));
    c->ovf_cap = (uint32_t)want;
    return HR_OK;
}

// Queues of the split pipeline for launches of up to `paths` paths: a step's rays are at most one main ray and one shadow ray per emitter
// for every path (renderer.rs:274), both parities of the ray queue, one hit per ray, both parities of the live-path state.
static int ensure_wf(hr_ctx *c, uint64_t paths) {
    // per sub-queue: the items (tile x sampling) it owns x 64 paths, and per path a main ray + one shadow ray per emitter
    const uint64_t sub_paths = ((paths / 64u + WF_SUBQ - 1u) / WF_SUBQ) * 64u, sub_rays = sub_paths * (1ull + c->dsc.num_emitters);
    if (sub_paths <= c->wf.cap_paths && sub_rays <= c->wf.cap_rays && c->wf_block && (c->wf_has_prec || !c->precise)) return HR_OK;
    if (sub_rays * WF_SUBQ >= 0xffffffffull) return fail(HR_ERR_UNSUPPORTED, "split pipeline: %llu ray slots per launch exceed the 32-bit queue index (reduce option batch)", (unsigned long long)(sub_rays * WF_SUBQ));
    int rc = sync_all(c);
    if (rc) return rc;
    if (c->wf_block) { HIP_TRY(hipFree(c->wf_block)); c->wf_block = nullptr; }
    c->wf.cap_paths = c->wf.cap_rays = 0;
    const uint64_t rays = sub_rays * WF_SUBQ, pths = sub_paths * WF_SUBQ;
    const size_t ray_q = (size_t)rays * sizeof(f4), st_q = (size_t)pths * sizeof(f4), cnt = (WF_STEPS + 2) * WF_SUBQ * sizeof(WfCounts);
    const size_t total = cnt + 4 * ray_q + (size_t)rays * sizeof(WfHitRec) + (c->precise ? 10 : 6) * st_q;
    hipError_t e = hipMalloc(&c->wf_block, total);
    if (e != hipSuccess) { c->wf_block = nullptr; return fail(HR_ERR_DEVICE, "split pipeline: %.1f GiB of queues: %s", (double)total / (1ull << 30), hipGetErrorString(e)); }
    char *b = (char *)c->wf_block;
    c->wf.counts = (WfCounts *)b; b += cnt;
    for (int i = 0; i < 2; i++) { c->wf.ray_a[i] = (f4 *)b; b += ray_q; c->wf.ray_b[i] = (f4 *)b; b += ray_q; }
    c->wf.hits = (WfHitRec *)b; b += (size_t)rays * sizeof(WfHitRec);
    for (int i = 0; i < 2; i++) { c->wf.st_a[i] = (f4 *)b; b += st_q; c->wf.st_b[i] = (f4 *)b; b += st_q; c->wf.st_c[i] = (f4 *)b; b += st_q; }
    for (int i = 0; i < 2; i++) { c->wf.st_d[i] = c->wf.st_e[i] = c->wf.st_f[i] = nullptr; c->wf.tag[i] = nullptr; }
    if (c->precise) for (int i = 0; i < 2; i++) { c->wf.st_d[i] = (f4 *)b; b += st_q; c->wf.st_e[i] = (f4 *)b; b += st_q; }
    c->wf_has_prec = c->precise;
    c->wf.cap_paths = (uint32_t)sub_paths; c->wf.cap_rays = (uint32_t)sub_rays;
    return HR_OK;
}
// One launch through the split pipeline, on the main stream: camera rays, then per path iteration the traversal kernel over the step's rays
// and the shading kernel over its live paths.  Empty steps (every path has ended) are two kernels that read one counter and leave.
static int launch_split(hr_ctx *c, const RenderParams &rp, int slot, std::vector<hipEvent_t> *marks = nullptr, uint32_t *plog = nullptr, const WfQueues *queues = nullptr) {
    const WfQueues wq = queues ? *queues : c->wf;
    hipStream_t st = c->stream;
    auto mark = [&]() -> hipError_t { if (!marks) return hipSuccess; hipEvent_t e; hipError_t r = hipEventCreate(&e); if (r != hipSuccess) return r; marks->push_back(e); return hipEventRecord(e, st); };
    HIP_TRY(hipMemsetAsync(wq.counts, 0, (WF_STEPS + 2) * WF_SUBQ * sizeof(WfCounts), st));
    const uint32_t items = rp.tiles_x * rp.tiles_y * rp.num_k;
    const bool qn = c->dsc.qnodes != nullptr;
    HIP_TRY(mark());
    // grids: whole multiples of WF_SUBQ waves (16 workgroups of 4), so that every sub-queue has the same number of waves
    auto grid_of = [&](uint32_t wgs_per_cu) { return dim3(std::max<uint32_t>(16u, (uint32_t)c->num_cus * wgs_per_cu / 16u * 16u)); };
    (void)items;
    if (c->precise) hipLaunchKernelGGL(wf_start_kernel<true>, grid_of(8u), dim3(256), 0, st, c->dsc, rp, c->recs[slot], wq);
    else hipLaunchKernelGGL(wf_start_kernel<false>, grid_of(8u), dim3(256), 0, st, c->dsc, rp, c->recs[slot], wq);
    HIP_TRY(mark());
    RenderParams rt = rp;
    rt.adv_den = c->wf_adv_den;
    const dim3 gt = grid_of(c->wf_trav_wgs), gs = grid_of(c->wf_shade_wgs), b(256);
    for (uint32_t step = 1; step <= WF_STEPS; step++) {
        if (c->counters && !plog) {
            if (qn) hipLaunchKernelGGL((wf_traverse_kernel<true, true>), gt, b, 0, st, c->dsc, rt, wq, step, c->d_counters);
            else hipLaunchKernelGGL((wf_traverse_kernel<true, false>), gt, b, 0, st, c->dsc, rt, wq, step, c->d_counters);
        } else {
            if (qn) hipLaunchKernelGGL((wf_traverse_kernel<false, true>), gt, b, 0, st, c->dsc, rt, wq, step, c->d_counters);
            else hipLaunchKernelGGL((wf_traverse_kernel<false, false>), gt, b, 0, st, c->dsc, rt, wq, step, c->d_counters);
        }
        HIP_TRY(mark());
#define HR_LAUNCH_SHADE(C, P, L) hipLaunchKernelGGL((wf_shade_kernel<C, P, L>), gs, b, 0, st, c->dsc, rp, c->recs[slot], wq, step, c->d_counters, plog)
        if (plog) { if (c->precise) HR_LAUNCH_SHADE(false, true, true); else HR_LAUNCH_SHADE(false, false, true); }
        else if (c->counters) { if (c->precise) HR_LAUNCH_SHADE(true, true, false); else HR_LAUNCH_SHADE(true, false, false); }
        else if (c->precise) HR_LAUNCH_SHADE(false, true, false);
        else HR_LAUNCH_SHADE(false, false, false);
#undef HR_LAUNCH_SHADE
        HIP_TRY(mark());
    }
    HIP_TRY(hipGetLastError());
    return HR_OK;
}

static int launch_seed(hr_ctx *c, const RenderParams &rp, int slot, hipStream_t st) {
    uint64_t paths = (uint64_t)rp.tiles_x * rp.tiles_y * rp.num_k * 64u;
    uint32_t grid = (uint32_t)std::min<uint64_t>((paths + SEED_COLS - 1) / SEED_COLS, (uint64_t)c->num_cus);
    EventPair ev{nullptr, nullptr};
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventRecord(ev.a, st));
    if (c->debug_skip & 2) {
    } else if (c->seed_mode == 2) {
        if (!c->ring) HIP_TRY(hipMalloc((void **)&c->ring, (size_t)c->num_cus * SEED_RING_WORDS_MAX * sizeof(u64)));
#define HR_LAUNCH_SEG(P, L) hipLaunchKernelGGL((seed_seg_kernel<P, L>), dim3(grid), dim3(256), SEED_LDS_BYTES, st, rp, c->dsc.cam.lens_shape, c->ring, c->recs[slot], c->ovf, c->ovf_win, c->d_counters)
        if (rp.rec_lo_off) { if (c->seed_prof) HR_LAUNCH_SEG(true, true); else HR_LAUNCH_SEG(false, true); }   // + the draws' residuals (precise shading)
        else if (c->seed_prof) HR_LAUNCH_SEG(true, false);
        else HR_LAUNCH_SEG(false, false);
#undef HR_LAUNCH_SEG
#if defined(HR_EXPERIMENTS)
    } else if (c->seed_mode == 4) {
        if (!c->ring) HIP_TRY(hipMalloc((void **)&c->ring, (size_t)c->num_cus * SEED_RING_WORDS_MAX * sizeof(u64)));
        if (c->seed_prof) hipLaunchKernelGGL((seed_w5_kernel<true>), dim3(grid), dim3(320), SEED_LDS_BYTES, st, rp, c->dsc.cam.lens_shape, c->ring, c->recs[slot], c->ovf, c->ovf_win, c->d_counters);
        else hipLaunchKernelGGL((seed_w5_kernel<false>), dim3(grid), dim3(320), SEED_LDS_BYTES, st, rp, c->dsc.cam.lens_shape, c->ring, c->recs[slot], c->ovf, c->ovf_win, c->d_counters);
    } else if (c->seed_mode == 3) {
        if (!c->ring) HIP_TRY(hipMalloc((void **)&c->ring, (size_t)c->num_cus * SEED_RING_WORDS_MAX * sizeof(u64)));
#define HR_LAUNCH_PS(P) hipLaunchKernelGGL(seed_ps_kernel<P>, dim3(grid), dim3(256), SEED_LDS_BYTES, st, rp, c->dsc.cam.lens_shape, c->ring, c->recs[slot], c->ovf, c->ovf_win, c->d_counters)
        switch (c->seed_prof) { case 1: HR_LAUNCH_PS(1); break; case 2: HR_LAUNCH_PS(2); break; case 3: HR_LAUNCH_PS(3); break; default: HR_LAUNCH_PS(0); break; }
#undef HR_LAUNCH_PS
#endif
    } else if (c->seed_mode == 1) {
        if (!c->ring) HIP_TRY(hipMalloc((void **)&c->ring, (size_t)c->num_cus * SEED_RING_WORDS_MAX * sizeof(u64)));
#define HR_LAUNCH_PC(HEAD) hipLaunchKernelGGL(seed_pc_kernel<HEAD>, dim3(grid), dim3(256), SEED_LDS_BYTES, st, rp, c->dsc.cam.lens_shape, c->ring, c->recs[slot], c->ovf, c->ovf_win, c->d_counters)
        if (c->seed_prof && c->seed_split == 20) hipLaunchKernelGGL((seed_pc_kernel<20, true>), dim3(grid), dim3(256), SEED_LDS_BYTES, st, rp, c->dsc.cam.lens_shape, c->ring, c->recs[slot], c->ovf, c->ovf_win, c->d_counters);
        else if (c->seed_prof) hipLaunchKernelGGL((seed_pc_kernel<16, true>), dim3(grid), dim3(256), SEED_LDS_BYTES, st, rp, c->dsc.cam.lens_shape, c->ring, c->recs[slot], c->ovf, c->ovf_win, c->d_counters);
        else switch (c->seed_split) {
            case 8: HR_LAUNCH_PC(8); break;
            case 12: HR_LAUNCH_PC(12); break;
            case 20: HR_LAUNCH_PC(20); break;
            case 24: HR_LAUNCH_PC(24); break;
            case 28: HR_LAUNCH_PC(28); break;
            default: HR_LAUNCH_PC(16); break;
        }
#undef HR_LAUNCH_PC
    } else
        hipLaunchKernelGGL(seed_isaac64_kernel, dim3(grid), dim3(64 * SEED_WAVES), SEED_LDS_BYTES, st, rp, c->dsc.cam.lens_shape, c->recs[slot], c->ovf, c->ovf_win,
                           c->d_counters);
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
    invalidate_totals(c);
    uint32_t total_k = (s_end - s_begin + stride - 1) / stride;
    RenderParams rp{};
    rp.width = c->W; rp.height = c->H;
    rp.tiles_x = (c->W + 3) / 4; rp.tiles_y = (c->H + 3) / 4;
    rp.stride = stride;
    rp.adv_den = c->adv_den;
    rp.leaf_den = c->leaf_den;
    rp.node_unroll = c->node_unroll; rp.kchunk = c->kchunk;
    rp.pad[0] = c->seed_prio;
    rp.pad[2] = (uint32_t)c->debug_skip;
    uint32_t tiles = rp.tiles_x * rp.tiles_y;
    // the hand-off costs 8 KiB per (tile, sampling): keep each of the two buffers under max_tail_bytes
    uint32_t batch = c->batch;
    if (!batch) {   // automatic: launches of the size the kernels are tuned on (4 samplings of 1920x1080), at most 64 samplings
        const uint64_t per_sampling_paths = (uint64_t)tiles * 64u;
        batch = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(4, (33177600ull + per_sampling_paths - 1) / per_sampling_paths));
    }
    {
        uint64_t per_sampling = (uint64_t)tiles * REC_ITEM_FLOATS * sizeof(float) * (draws_twin(c) ? 2 : 1);
        uint64_t fit = std::max<uint64_t>(1, c->max_tail_bytes / std::max<uint64_t>(1, per_sampling));
        batch = (uint32_t)std::min<uint64_t>(batch, fit);
    }
    if (c->precise_opt == 1 && c->rr_start) return fail(HR_ERR_UNSUPPORTED, "hr_render: russian_roulette and precise_shading exclude each other (the roulette estimator has no f64 instantiation)");
    const bool split = c->trace_mode == 1 && !c->rr_start;   // (the roulette estimator lives in the megakernel only)
    if (split) {
        // the split pipeline's queues are sized for the worst case (a main ray + a shadow ray per emitter for every path, both parities): keep
        // them under the same cap as a hand-off buffer — a 3840x2160 launch then holds one sampling (33 M paths) instead of four
        const uint64_t per_path = (1ull + c->dsc.num_emitters) * (4 * sizeof(f4) + sizeof(WfHitRec)) + (c->precise ? 10 : 6) * sizeof(f4);
        const uint64_t fit = std::max<uint64_t>(1, c->max_tail_bytes / std::max<uint64_t>(1, (uint64_t)tiles * 64u * per_path));
        batch = (uint32_t)std::min<uint64_t>(batch, fit);
    }
    int rc = ensure_draws(c, (size_t)tiles * batch);
    if (rc) return rc;
    rp.rec_lo_off = rec_lo_off(c);
    if ((rc = ensure_ovf(c, (uint64_t)tiles * 64u * batch))) return rc;
    if (split && (rc = ensure_wf(c, (uint64_t)tiles * 64u * batch))) return rc;
    rp.ovf_cap = c->ovf_cap;
    rp.rr_start = c->rr_start;
    rp.nee_cull_off = ~c->nee_cull & 7u;
    rp.tail_div = c->tail_div;
    for (uint32_t done = 0; done < total_k; done += batch) {
        uint32_t nk = std::min(batch, total_k - done);
        rp.sampling_begin = s_begin + done * stride;
        rp.num_k = nk;
        int slot = (int)(c->batch_counter & 1);
        c->batch_counter++;
        // the priority governor lives on the device (governor_kernel above): the kernels read its level when they start
        rp.gov = c->gov; rp.gov_slot = (uint32_t)slot;
        rp.trace_boost = 0;
        rp.pad[1] = c->init_prio;   // the producer waves' priority at level 0
        hipStream_t sstream = c->seed_stream;  // (alternating two seed streams to overlap kernel tails was measured: no gain)
        // seed of this batch may only overwrite draws[slot] once the trace that read it has finished
        if (c->trace_pending[slot]) HIP_TRY(hipStreamWaitEvent(sstream, c->trace_done[slot], 0));
        if ((rc = launch_seed(c, rp, slot, sstream))) return rc;
        HIP_TRY(hipEventRecord(c->seed_done[slot], sstream));
        c->seed_pending[slot] = true;
        HIP_TRY(hipStreamWaitEvent(c->stream, c->seed_done[slot], 0));
        EventPair ev{nullptr, nullptr};
        HIP_TRY(hipEventCreate(&ev.a));
        HIP_TRY(hipEventCreate(&ev.b));
        HIP_TRY(hipEventRecord(ev.a, c->stream));
        // persistent waves: enough workgroups to fill every CU (6 per CU covers every occupancy variant), never more
        // waves than tiles
        const uint32_t kch = c->kchunk ? c->kchunk : TRACE_KCHUNK;
        const uint64_t units = (uint64_t)tiles * ((nk + kch - 1) / kch);   // work units of the trace kernel
        uint32_t grid = (uint32_t)std::min<uint64_t>(c->trace_grid ? c->trace_grid : (uint64_t)c->num_cus * c->trace_wgs, (units + TRACE_WAVES - 1) / TRACE_WAVES);
        rp.wg_budget = c->trace_budget;
        HIP_TRY(hipMemsetAsync(c->d_tile_counter + slot, 0, sizeof(uint32_t), c->stream));
        {
            dim3 g(grid), b(64 * TRACE_WAVES);
#define HR_LAUNCH_TRACE(C, W, Q) hipLaunchKernelGGL((trace_kernel<C, W, Q>), g, b, 0, c->stream, c->dsc, rp, c->recs[slot], c->d_counters, c->d_tile_counter + slot)
            const bool qn = c->dsc.qnodes != nullptr;
#define HR_LAUNCH_TRACE_RR(C, W, Q) hipLaunchKernelGGL((trace_kernel<C, W, Q, true>), g, b, 0, c->stream, c->dsc, rp, c->recs[slot], c->d_counters, c->d_tile_counter + slot)
            if (c->debug_skip & 16) {
            } else if (split) {
                if ((rc = launch_split(c, rp, slot))) return rc;
            } else if (c->rr_start) {   // the non-parity estimator has its own instantiations (one occupancy variant)
                if (c->counters) { if (qn) HR_LAUNCH_TRACE_RR(true, 3, true); else HR_LAUNCH_TRACE_RR(true, 3, false); }
                else if (qn) HR_LAUNCH_TRACE_RR(false, 5, true);
                else HR_LAUNCH_TRACE_RR(false, 5, false);
            } else if (c->precise) {    // precise shading: path_advance<.., PREC>, 128 VGPRs
#define HR_LAUNCH_TRACE_PREC(C, W, Q) hipLaunchKernelGGL((trace_kernel<C, W, Q, false, false, true>), g, b, 0, c->stream, c->dsc, rp, c->recs[slot], c->d_counters, c->d_tile_counter + slot)
                if (c->counters) { if (qn) HR_LAUNCH_TRACE_PREC(true, 3, true); else HR_LAUNCH_TRACE_PREC(true, 3, false); }
                else if (qn && c->min_waves == 6) HR_LAUNCH_TRACE_PREC(false, 5, true);      // debug option min_waves 6 -> the 96-VGPR form, 4 -> the 168-VGPR form (A/B only)
                else if (qn && c->min_waves == 4) HR_LAUNCH_TRACE_PREC(false, 3, true);
                else if (qn) HR_LAUNCH_TRACE_PREC(false, 4, true);
                else HR_LAUNCH_TRACE_PREC(false, 4, false);
#undef HR_LAUNCH_TRACE_PREC
            } else if (c->counters) { if (qn) HR_LAUNCH_TRACE(true, 3, true); else HR_LAUNCH_TRACE(true, 3, false); }
            else if (!qn) HR_LAUNCH_TRACE(false, 5, false);
            else if (c->min_waves == 4) HR_LAUNCH_TRACE(false, 4, true);
            else if (c->min_waves == 6) HR_LAUNCH_TRACE(false, 6, true);
            else HR_LAUNCH_TRACE(false, 5, true);
#undef HR_LAUNCH_TRACE
#undef HR_LAUNCH_TRACE_RR
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev.b, c->stream));
        c->trace_events.push_back(ev);
        c->trace_launches++;
        // the launch's radiance into the accumulator (the trace kernel left every path's in its record), in the gap in which this
        // stream waits for the next seed kernel anyway
        if (!(c->debug_skip & 16)) {
            hipLaunchKernelGGL(accumulate_kernel, dim3((tiles + 3) / 4), dim3(256), 0, c->stream, rp, c->recs[slot], c->accum);
            HIP_TRY(hipGetLastError());
        }
        // the governor judges the launch that has just finished and frees its stamps; the seed kernel that reuses the slot waits for
        // trace_done, recorded behind it
        hipLaunchKernelGGL(governor_kernel, dim3(1), dim3(1), 0, c->stream, c->gov, (uint32_t)slot);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->trace_done[slot], c->stream));
        c->trace_pending[slot] = true;
        c->paths_rendered += (uint64_t)c->W * c->H * 4 * nk;
        if (c->trace_events.size() >= 64 && c->seed_events.size() == c->trace_events.size()) retire_finished_launches(c);
        if (c->trace_events.size() > 4096) {  // (never reached while launches finish: the host would have to be 4,096 launches ahead)
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
    invalidate_totals(c);
    RenderParams rp{};
    rp.width = c->W; rp.height = c->H;
    rp.tiles_x = (c->W + 3) / 4; rp.tiles_y = (c->H + 3) / 4;
    rp.leaf_den = c->leaf_den; rp.node_unroll = c->node_unroll;
    EventPair ev{nullptr, nullptr};
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventRecord(ev.a, c->stream));
    {
        const uint32_t tiles = rp.tiles_x * rp.tiles_y;
        dim3 g((tiles + TRACE_WAVES - 1) / TRACE_WAVES), b(64 * TRACE_WAVES);
        const bool qn = c->dsc.qnodes != nullptr;
#define HR_LAUNCH_DEBUG(C, Q) hipLaunchKernelGGL((debug_render_kernel<C, Q>), g, b, 0, c->stream, c->dsc, rp, mode, c->accum, c->d_counters)
        if (c->counters) { if (qn) HR_LAUNCH_DEBUG(true, true); else HR_LAUNCH_DEBUG(true, false); }
        else { if (qn) HR_LAUNCH_DEBUG(false, true); else HR_LAUNCH_DEBUG(false, false); }
#undef HR_LAUNCH_DEBUG
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev.b, c->stream));
    c->debug_events.push_back(ev);
    c->debug_launches++;
    return HR_OK;
}

int hr_synchronize(hr_ctx *c) {
    if (!c) return fail(HR_ERR_INVALID, "hr_synchronize: null ctx");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    Counters h;
    HIP_TRY(hipMemcpy(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
    if (h.rng_overflow) return fail(HR_ERR_RNG_WINDOW, "%llu paths needed more than %d ISAAC-64 outputs for the lens rejection loop (or the fix-up queue overflowed)", h.rng_overflow, ISAAC_TAIL);
    return HR_OK;
}

int hr_mark(hr_ctx *c, uint64_t *ticket) {
    if (!c || !ticket) return fail(HR_ERR_INVALID, "hr_mark: null argument");
    HIP_TRY(hipSetDevice(c->device));
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, c->stream);
    if (e != hipSuccess) { (void)hipEventDestroy(ev); return fail(HR_ERR_DEVICE, "hr_mark: %s", hipGetErrorString(e)); }
    *ticket = c->next_ticket++;
    c->markers.emplace_back(*ticket, ev);
    return HR_OK;
}
int hr_wait(hr_ctx *c, uint64_t ticket) {
    if (!c) return fail(HR_ERR_INVALID, "hr_wait: null ctx");
    HIP_TRY(hipSetDevice(c->device));
    size_t n = 0;
    while (n < c->markers.size() && c->markers[n].first <= ticket) n++;
    if (!n) return HR_OK;   // already waited for (or swept by hr_synchronize)
    hipError_t e = hipEventSynchronize(c->markers[n - 1].second);
    for (size_t i = 0; i < n; i++) (void)hipEventDestroy(c->markers[i].second);
    c->markers.erase(c->markers.begin(), c->markers.begin() + (long)n);
    if (e != hipSuccess) return fail(HR_ERR_DEVICE, "hr_wait: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_read_accumulator(hr_ctx *c, float *host) {
    if (!c || !host) return fail(HR_ERR_INVALID, "hr_read_accumulator: null argument");
    if (!c->accum) return fail(HR_ERR_NO_TARGET, "hr_read_accumulator: no accumulator");
    int rc = hr_synchronize(c);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(host, c->total_valid ? c->accum_total : c->accum, (size_t)c->W * c->H * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return HR_OK;
}
int hr_write_accumulator(hr_ctx *c, const float *host) {
    if (!c || !host) return fail(HR_ERR_INVALID, "hr_write_accumulator: null argument");
    if (!c->accum) return fail(HR_ERR_NO_TARGET, "hr_write_accumulator: no accumulator");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    invalidate_totals(c);
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
    EventPair ev{nullptr, nullptr};
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventRecord(ev.a, c->stream));
    hipLaunchKernelGGL(tonemap_gamma_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->total_valid ? c->accum_total : c->accum, c->post_tmp, n, scale);
    hipLaunchKernelGGL(bilateral_quantise_kernel, dim3((c->W + 31) / 32, (c->H + 7) / 8), dim3(32, 8), 0, c->stream, c->post_tmp, c->d_rgb8, c->W, c->H);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev.b, c->stream));
    c->post_events.push_back(ev);
    HIP_TRY(hipMemcpyAsync(host_rgb8, c->d_rgb8, (size_t)n * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return drain_events(c);
}

// ---- multi-GPU: one all-reduce of the accumulators over RCCL (hr_comm.h) ------------------------------------------------------
#define NCCL_TRY(expr)                                                                                                    \
    do {                                                                                                                  \
        int r_ = (expr);                                                                                                  \
        if (r_ != 0) return fail(HR_ERR_DEVICE, "%s failed: %s", #expr, hrcomm::api().GetErrorString ? hrcomm::api().GetErrorString(r_) : "?"); \
    } while (0)

int hr_comm_get_unique_id(void *id_out) {
    if (!id_out) return fail(HR_ERR_INVALID, "hr_comm_get_unique_id: null argument");
    if (!hrcomm::load()) return fail(HR_ERR_UNSUPPORTED, "%s", hrcomm::api().error.c_str());
    hrcomm::UniqueId id;
    NCCL_TRY(hrcomm::api().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return HR_OK;
}
__global__ void add_accumulator_kernel(float *__restrict__ total, const float *__restrict__ part, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) total[i] += part[i];
}
static int comm_release(hr_ctx *c) {
    if (c->comm) { NCCL_TRY(hrcomm::api().CommDestroy(c->comm)); c->comm = nullptr; }
    for (hr_ctx *p : c->same_device_peers) if (p != c) { p->same_device_peers.clear(); p->comm_world = 0; p->total_valid = false; p->comm_path = HR_COMM_NONE; p->allreduces = 0; }
    c->same_device_peers.clear();
    c->comm_world = 0; c->comm_rank = 0; c->total_valid = false;
    c->comm_path = HR_COMM_NONE; c->allreduces = 0;
    return HR_OK;
}
int hr_comm_init_rank(hr_ctx *c, const void *id, int world_size, int rank) {
    if (!c || !id || world_size < 1 || rank < 0 || rank >= world_size) return fail(HR_ERR_INVALID, "hr_comm_init_rank: bad argument");
    if (!hrcomm::load()) return fail(HR_ERR_UNSUPPORTED, "%s", hrcomm::api().error.c_str());
    HIP_TRY(hipSetDevice(c->device));
    int rc = comm_release(c);
    if (rc) return rc;
    hrcomm::UniqueId uid;
    memcpy(&uid, id, sizeof uid);
    NCCL_TRY(hrcomm::api().CommInitRank(&c->comm, world_size, uid, rank));
    c->comm_world = world_size; c->comm_rank = rank; c->comm_path = HR_COMM_RCCL_RANK;
    return HR_OK;
}
int hr_comm_init_local(hr_ctx **ctxs, int n) {
    if (!ctxs || n < 1) return fail(HR_ERR_INVALID, "hr_comm_init_local: bad argument");
    for (int i = 0; i < n; i++) if (!ctxs[i]) return fail(HR_ERR_INVALID, "hr_comm_init_local: null context");
    {
        // all contexts on ONE device (RCCL wants one rank per device): the "collective" is a sum kernel on that device
        bool same = n > 1;
        for (int i = 1; i < n; i++) same = same && ctxs[i]->device == ctxs[0]->device;
        if (same) {
            for (int i = 0; i < n; i++) { int rc = comm_release(ctxs[i]); if (rc) return rc; }
            for (int i = 0; i < n; i++) { ctxs[i]->same_device_peers.assign(ctxs, ctxs + n); ctxs[i]->comm_world = n; ctxs[i]->comm_rank = i; ctxs[i]->comm_path = HR_COMM_SAME_DEVICE_SUM; }
            return HR_OK;
        }
    }
    if (!hrcomm::load()) return fail(HR_ERR_UNSUPPORTED, "%s", hrcomm::api().error.c_str());
    std::vector<int> devs(n);
    std::vector<hrcomm::Comm> comms(n, nullptr);
    for (int i = 0; i < n; i++) {
        int rc = comm_release(ctxs[i]);
        if (rc) return rc;
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; j++) if (devs[j] == devs[i]) return fail(HR_ERR_INVALID, "hr_comm_init_local: device %d appears twice (RCCL needs one rank per device)", devs[i]);
    }
    NCCL_TRY(hrcomm::api().CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; i++) { ctxs[i]->comm = comms[i]; ctxs[i]->comm_world = n; ctxs[i]->comm_rank = i; ctxs[i]->comm_path = HR_COMM_RCCL_GROUP; }
    return HR_OK;
}
int hr_comm_destroy(hr_ctx *c) {
    if (!c) return fail(HR_ERR_INVALID, "hr_comm_destroy: null ctx");
    if (!c->comm && c->same_device_peers.empty()) return HR_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    return comm_release(c);
}
// enqueue this rank's part of the collective behind its render work (caller: inside a group when it drives several ranks)
static int allreduce_enqueue(hr_ctx *c) {
    if (!c->comm && c->same_device_peers.empty()) return fail(HR_ERR_INVALID, "hr_allreduce_accumulator: no communicator (hr_comm_init_rank / hr_comm_init_local)");
    if (!c->accum || !c->W) return fail(HR_ERR_NO_TARGET, "hr_allreduce_accumulator: no accumulator");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->W * c->H * 3;
    if (!c->accum_total) HIP_TRY(hipMalloc((void **)&c->accum_total, n * sizeof(float)));
    if (!c->same_device_peers.empty()) {
        for (hr_ctx *p : c->same_device_peers) {
            if (p->W != c->W || p->H != c->H || !p->accum) return fail(HR_ERR_INVALID, "hr_allreduce_accumulator: the contexts of the group differ in resolution");
            if (p != c) { int rc = sync_all(p); if (rc) return rc; }   // the peers' render work (their own streams)
        }
        // rank order, so that every context of the group gets bit-identical totals (as an all-reduce delivers them)
        int rc = sync_all(c);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(c->accum_total, c->same_device_peers[0]->accum, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        for (size_t k = 1; k < c->same_device_peers.size(); k++)
            hipLaunchKernelGGL(add_accumulator_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->accum_total, c->same_device_peers[k]->accum, n);
        HIP_TRY(hipGetLastError());
        // the adds read the PEERS' accumulators from this context's stream: they are finished before the call returns, so that a
        // peer's next hr_render / hr_clear / hr_write_accumulator cannot race with them
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->total_valid = true;
        c->allreduces++;
        return HR_OK;
    }
    NCCL_TRY(hrcomm::api().AllReduce(c->accum, c->accum_total, n, hrcomm::kFloat, hrcomm::kSum, c->comm, c->stream));
    c->allreduces++;
    return HR_OK;   // total_valid is set by the callers once the collective is known to be enqueued (group end)
}
int hr_allreduce_accumulator(hr_ctx *c) {
    if (!c) return fail(HR_ERR_INVALID, "hr_allreduce_accumulator: null ctx");
    int rc = allreduce_enqueue(c);
    if (rc == HR_OK) c->total_valid = true;
    return rc;
}
int hr_allreduce_accumulators(hr_ctx **ctxs, int n) {
    if (!ctxs || n < 1) return fail(HR_ERR_INVALID, "hr_allreduce_accumulators: bad argument");
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || (!ctxs[i]->comm && ctxs[i]->same_device_peers.empty())) return fail(HR_ERR_INVALID, "hr_allreduce_accumulators: context %d has no communicator", i);
    if (!ctxs[0]->same_device_peers.empty()) {
        for (int i = 0; i < n; i++) { int rc = allreduce_enqueue(ctxs[i]); if (rc) return rc; }
        return HR_OK;
    }
    NCCL_TRY(hrcomm::api().GroupStart());
    int rc = HR_OK;
    for (int i = 0; i < n && rc == HR_OK; i++) rc = allreduce_enqueue(ctxs[i]);
    int r_ = hrcomm::api().GroupEnd();
    if (rc) return rc;
    if (r_ != 0) return fail(HR_ERR_DEVICE, "ncclGroupEnd failed: %s", hrcomm::api().GetErrorString(r_));
    for (int i = 0; i < n; i++) ctxs[i]->total_valid = true;
    return HR_OK;
}
void *hr_total_device_ptr(hr_ctx *c) { return c && c->total_valid ? c->accum_total : nullptr; }

// What the communicator says about itself — asked of RCCL, not remembered from the init call: the evidence a bench line needs that
// its all-reduce ran over N ranks (ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion).
int hr_comm_info(hr_ctx *c, hr_comm_info_t *out) {
    if (!c || !out) return fail(HR_ERR_INVALID, "hr_comm_info: null argument");
    memset(out, 0, sizeof *out);
    out->path = c->comm_path; out->device = c->device; out->allreduces = c->allreduces;
    if (c->comm) {
        int v = 0;
        NCCL_TRY(hrcomm::api().CommCount(c->comm, &v)); out->nranks = v;
        NCCL_TRY(hrcomm::api().CommUserRank(c->comm, &v)); out->rank = v;
        NCCL_TRY(hrcomm::api().CommCuDevice(c->comm, &v)); out->device = v;
        NCCL_TRY(hrcomm::api().GetVersion(&v)); out->rccl_version = v;
    } else if (!c->same_device_peers.empty()) {
        out->nranks = (int32_t)c->same_device_peers.size(); out->rank = c->comm_rank;
    }
    return HR_OK;
}

int hr_comm_library(char *path_out, size_t cap, int *reused_out) {
    if (!path_out || !cap) return fail(HR_ERR_INVALID, "hr_comm_library: null argument");
    if (!hrcomm::load()) return fail(HR_ERR_UNSUPPORTED, "%s", hrcomm::api().error.c_str());
    snprintf(path_out, cap, "%s", hrcomm::api().path.c_str());
    if (reused_out) *reused_out = hrcomm::api().reused ? 1 : 0;
    return HR_OK;
}

// Sum of an accumulator in f64, per channel, on the device: sum_of(rank's own accumulators) == sum(all-reduced total) is the checksum of
// the exchange (bench.py multi_gpu.checksum).  Deterministic: a fixed grid, every workgroup leaves its partial sums (waves in order), the
// host adds the 1,024 partial sums in order — two contexts that hold the same total report the same sum to the last bit.
static const unsigned ACC_SUM_BLOCKS = 1024;
__global__ __launch_bounds__(256) void accumulator_sum_kernel(const float *__restrict__ a, size_t pixels, double *__restrict__ out) {
    double s[3] = {0.0, 0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += (size_t)gridDim.x * blockDim.x) {
        s[0] += (double)a[i * 3]; s[1] += (double)a[i * 3 + 1]; s[2] += (double)a[i * 3 + 2];
    }
    __shared__ double part[4][3];
    for (int k = 0; k < 3; k++) {
        double x = s[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3) out[blockIdx.x * 3 + threadIdx.x] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
}
int hr_accumulator_sum(hr_ctx *c, int which, double out_rgb[3]) {
    if (!c || !out_rgb || which < 0 || which > 1) return fail(HR_ERR_INVALID, "hr_accumulator_sum: bad argument (which: 0 = this context's own accumulator, 1 = the all-reduced total)");
    if (!c->accum || !c->W) return fail(HR_ERR_NO_TARGET, "hr_accumulator_sum: no accumulator");
    if (which == 1 && !c->total_valid) return fail(HR_ERR_INVALID, "hr_accumulator_sum: no all-reduced total (hr_allreduce_accumulator first)");
    int rc = hr_synchronize(c);
    if (rc) return rc;
    double *d = nullptr;
    std::vector<double> h(ACC_SUM_BLOCKS * 3);
    HIP_TRY(hipMalloc((void **)&d, h.size() * sizeof(double)));
    hipLaunchKernelGGL(accumulator_sum_kernel, dim3(ACC_SUM_BLOCKS), dim3(256), 0, c->stream, which ? c->accum_total : c->accum, (size_t)c->W * c->H, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(HR_ERR_DEVICE, "hr_accumulator_sum: %s", hipGetErrorString(e));
    out_rgb[0] = out_rgb[1] = out_rgb[2] = 0.0;
    for (unsigned b = 0; b < ACC_SUM_BLOCKS; b++) for (int k = 0; k < 3; k++) out_rgb[k] += h[b * 3 + k];
    return HR_OK;
}

int hr_get_stats(hr_ctx *c, hr_stats *out) {
    if (!c || !out) return fail(HR_ERR_INVALID, "hr_get_stats: null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    Counters h;
    HIP_TRY(hipMemcpyAsync(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memset(out, 0, sizeof *out);
    out->paths = c->counters ? h.paths : c->paths_rendered;
    out->rays = h.rays; out->node_tests = h.node_tests; out->tri_tests = h.tri_tests;
    out->sphere_tests = h.sphere_tests; out->cuboid_tests = h.cuboid_tests; out->rng_overflow = h.rng_overflow; out->shadow_culled = h.shadow_culled;
    out->seed_kernel_ms = c->seed_ms; out->trace_kernel_ms = c->trace_ms; out->post_kernel_ms = c->post_ms;
    out->seed_launches = c->seed_launches; out->trace_launches = c->trace_launches;
    out->bvh_build_ms = c->bvh_build_ms; out->bvh_builder_used = (uint64_t)c->builder_in_use;
    out->shading_in_force = c->precise ? (c->trace_mode == 1 ? 2u : 1u) : (c->trace_mode == 1 ? 3u : 0u);
    out->debug_kernel_ms = c->debug_ms; out->debug_launches = c->debug_launches;
    {
        GovDev g;
        HIP_TRY(hipMemcpyAsync(&g, c->gov, sizeof g, hipMemcpyDeviceToHost, c->stream));   // on the context's own stream: a null-stream copy could serialise against other contexts' launches
        HIP_TRY(hipStreamSynchronize(c->stream));
        out->governor_level = (uint64_t)(g.level < 0 ? 0 : g.level); out->governor_decisions = g.decisions; out->governor_moves = g.moves;
        out->governor_budget = g.budget; out->governor_budget_moves = g.budget_moves;
    }
    out->shade_calls = h.shade_calls; out->shade_lanes = h.shade_lanes; out->box_passes = h.box_passes; out->box_lanes = h.box_lanes;
    out->leaf_calls = h.leaf_calls; out->leaf_lanes = h.leaf_lanes; out->outer_iters = h.outer_iters;
    for (int i = 0; i < 4; i++) out->phase_cycles[i] = h.phase_cycles[i];
    for (int i = 0; i < 8; i++) out->seed_phase_cycles[i] = h.seed_phase[i];
    out->bvh_nodes = c->st_nodes; out->triangles = c->st_tris; out->spheres = c->st_spheres; out->cuboids = c->st_cuboids;
    return HR_OK;
}

int hr_set_option(hr_ctx *c, const char *key, double value) {
    if (!c || !key) return fail(HR_ERR_INVALID, "hr_set_option: null argument");
    HIP_TRY(hipSetDevice(c->device));
    std::string k = key;
    if (k == "counters") { c->counters = value != 0.0; return HR_OK; }
    if (k == "batch") {
        if (value < 0 || value > 64) return fail(HR_ERR_INVALID, "batch must be in [1,64], or 0 for automatic");
        int rc = sync_all(c);
        if (rc) return rc;
        c->batch = (uint32_t)value;
        return HR_OK;
    }
    if (k == "trace_boost") {
        if (value != -1 && !(value >= 0 && value <= 4 && value == (int)value)) return fail(HR_ERR_INVALID, "trace_boost must be -1 (governed by the measured kernel times) or a level 0 .. 4");
        int rc = sync_all(c);
        if (rc) return rc;
        c->trace_boost = (int)value;
        return govern_reset(c);
    }
    if (k == "quant_nodes") { c->quant_nodes = value != 0.0; return HR_OK; }
    if (k == "max_tail_gib") {
        if (value < 1 || value > 128) return fail(HR_ERR_INVALID, "max_tail_gib must be in [1,128]");
        c->max_tail_bytes = (uint64_t)value << 30;
        return HR_OK;
    }
    if (k == "split_ratio") {  // early split clipping of triangle references (-1 = automatic, 0 = off), next hr_upload_scene
        if ((value < 0 && value != -1) || value > 1000) return fail(HR_ERR_INVALID, "split_ratio must be -1 (automatic), 0 (off) or in (0,1000]");
        c->split_ratio = value;
        return HR_OK;
    }
    if (k == "bvh_builder") {  // takes effect at the next hr_upload_scene
        if (value != -1 && value != 0 && value != 1 && value != 2) return fail(HR_ERR_INVALID, "bvh_builder must be -1 (by scene size), 0 (host SAH), 1 (device LBVH) or 2 (device PLOC)");
        c->bvh_builder = (int)value;
        return HR_OK;
    }
    if (k == "max_leaf") {  // takes effect at the next hr_upload_scene
        if (value < 1 || value > 15) return fail(HR_ERR_INVALID, "max_leaf must be in [1,15]");
        c->max_leaf = (int)value;
        return HR_OK;
    }
    if (k == "rng_window") {
        if ((int)value != ISAAC_TAIL) return fail(HR_ERR_UNSUPPORTED, "rng_window is fixed at %d in this build", ISAAC_TAIL);
        return HR_OK;
    }
    if (k == "precise_shading") {   // the bounce geometry in f64 (split pipeline): closer to the reference's f64 arithmetic, a few per cent slower
        if (value != -1 && value != 0 && value != 1) return fail(HR_ERR_INVALID, "precise_shading must be -1 (automatic), 0 or 1");
        int rc = sync_all(c);
        if (rc) return rc;
        c->precise_opt = (int)value;
        return govern_reset(c);
    }
    if (k == "russian_roulette") {  // NOT image-preserving (see the header): 0 = off, else the first path iteration that plays
        if (value != 0 && (value < 2 || value > 9)) return fail(HR_ERR_INVALID, "russian_roulette must be 0 (off) or the first iteration that plays, in [2,9]");
        c->rr_start = (uint32_t)value;
        resolve_modes(c);     // (automatic precise shading stands back: the roulette estimator has no f64 instantiation)
        return HR_OK;
    }
    return fail(HR_ERR_INVALID, "unknown option '%s' (measurement knobs live behind hr_set_debug_option)", key);
}

// Measurement / experiment knobs.  Kept apart from hr_set_option on purpose: a host that only uses hr_set_option cannot change the
// kernels' schedule, and cannot reach "debug_skip", which produces a garbage image.
int hr_set_debug_option(hr_ctx *c, const char *key, double value) {
    if (!c || !key) return fail(HR_ERR_INVALID, "hr_set_debug_option: null argument");
    HIP_TRY(hipSetDevice(c->device));
    std::string k = key;
    if (k == "adv_den") {
        if (value < 1 || value > 64) return fail(HR_ERR_INVALID, "adv_den must be in [1,64]");
        c->adv_den = (uint32_t)value;
        return HR_OK;
    }
    if (k == "leaf_den") {
        if (value < 1 || value > 64) return fail(HR_ERR_INVALID, "leaf_den must be in [1,64]");
        c->leaf_den = (uint32_t)value;
        return HR_OK;
    }
    if (k == "min_waves") {
        if (value < 4 || value > 6) return fail(HR_ERR_INVALID, "min_waves must be in [4,6]");
        c->min_waves = (int)value;
        return HR_OK;
    }
    if (k == "kchunk") {
        if (value < 0 || value > 64) return fail(HR_ERR_INVALID, "kchunk must be in [1,64], or 0 for the default");
        c->kchunk = (uint32_t)value;
        return HR_OK;
    }
    if (k == "node_unroll") {
        if (value != 1 && value != 2) return fail(HR_ERR_INVALID, "node_unroll must be 1 or 2");
        c->node_unroll = (uint32_t)value;
        return HR_OK;
    }
    if (k == "tail_div") { c->tail_div = (uint32_t)value; return HR_OK; }
    if (k == "trace_grid") { c->trace_grid = (uint32_t)value; return HR_OK; }
    if (k == "trace_budget") { c->trace_budget = (uint32_t)value; return HR_OK; }
    if (k == "trace_wgs") {
        if (value < 1 || value > 8) return fail(HR_ERR_INVALID, "trace_wgs must be in [1,8]");
        c->trace_wgs = (uint32_t)value;
        return HR_OK;
    }
    if (k == "seed_prio") {
        if (value < 0 || value > 3) return fail(HR_ERR_INVALID, "seed_prio must be in [0,3]");
        c->seed_prio = (uint32_t)value;
        return HR_OK;
    }
    if (k == "init_prio") {
        if (value < 0 || value > 3) return fail(HR_ERR_INVALID, "init_prio must be in [0,3]");
        c->init_prio = (uint32_t)value;
        return HR_OK;
    }
    if (k == "seed_split") {
        if (value != 8 && value != 12 && value != 16 && value != 20 && value != 24 && value != 28) return fail(HR_ERR_INVALID, "seed_split must be 8, 12, 16, 20, 24 or 28");
        c->seed_split = (int)value;
        return HR_OK;
    }
    if (k == "seed_mode") {
        if (value != 0 && value != 1 && value != 2 && value != 3 && value != 4) return fail(HR_ERR_INVALID, "seed_mode must be 4 (five-wave four-run kernel), 3 (phase-shifted four-run kernel), 2 (three-run kernel), 1 (producer / consumer kernel with a state ring) or 0 (fused kernel)");
#if !defined(HR_EXPERIMENTS)
        if (value >= 3) return fail(HR_ERR_UNSUPPORTED, "seed_mode %d is a measured experiment (slower than the default): build with `make EXPERIMENTS=1` to have it", (int)value);
#endif
        int rc = sync_all(c);
        if (rc) return rc;
        c->seed_mode = (int)value;
        return HR_OK;
    }
    if (k == "seed_prof") { c->seed_prof = (int)value; return HR_OK; }
    if (k == "ploc_top") {
        if (value < 1 || value > (1 << 16)) return fail(HR_ERR_INVALID, "ploc_top must be in [1,65536]");
        c->ploc_top = (uint32_t)value;
        return HR_OK;
    }
    if (k == "trace_mode") {
        if (value != -1 && value != 0 && value != 1) return fail(HR_ERR_INVALID, "trace_mode must be -1 (automatic), 0 (megakernel) or 1 (split: traversal kernel + shading kernel)");
        int rc = sync_all(c);
        if (rc) return rc;
        c->trace_mode_opt = (int)value;
        return govern_reset(c);
    }
    if (k == "draw_residuals") { if (value != 0 && value != 1) return fail(HR_ERR_INVALID, "draw_residuals must be 0 or 1"); c->draw_residuals = (int)value; return HR_OK; }
    if (k == "wf_adv_den") { if (value < 0 || value > 64) return fail(HR_ERR_INVALID, "wf_adv_den must be in [0,64]"); c->wf_adv_den = (uint32_t)value; return HR_OK; }
    if (k == "wf_trav_wgs") { if (value < 1 || value > 16) return fail(HR_ERR_INVALID, "wf_trav_wgs must be in [1,16]"); c->wf_trav_wgs = (uint32_t)value; return HR_OK; }
    if (k == "wf_shade_wgs") { if (value < 1 || value > 16) return fail(HR_ERR_INVALID, "wf_shade_wgs must be in [1,16]"); c->wf_shade_wgs = (uint32_t)value; return HR_OK; }
    if (k == "debug_skip") { c->debug_skip = (int)value; return HR_OK; }
    if (k == "nee_cull") { c->nee_cull = (uint32_t)value & 7u; return HR_OK; }
    return fail(HR_ERR_INVALID, "unknown debug option '%s'", key);
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

static int path_draws_out(hr_ctx *c, uint32_t sampling, float *host_out, bool residuals, const char *who) {
    // the 20 fp32 draws per path exactly as the production seed kernel hands them to the trace kernel (residuals: the same slots of the
    // records' twin), re-ordered to pixel-major paths: out[((y*W + x)*4 + sub) * 20 + d]
    if (!c || !host_out) return fail(HR_ERR_INVALID, "%s: bad argument", who);
    if (!c->W) return fail(HR_ERR_NO_TARGET, "%s: hr_set_resolution not called", who);
    if (!c->have_scene) return fail(HR_ERR_NO_SCENE, "%s: no scene (lens shape needed)", who);
    if (residuals && !draws_twin(c)) return fail(HR_ERR_UNSUPPORTED, "%s: no residuals are written (needs precise shading in force, seed_mode 2, draw_residuals 1)", who);
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    RenderParams rp{};
    rp.width = c->W; rp.height = c->H; rp.tiles_x = (c->W + 3) / 4; rp.tiles_y = (c->H + 3) / 4;
    rp.sampling_begin = sampling; rp.stride = 1; rp.num_k = 1;
    uint32_t tiles = rp.tiles_x * rp.tiles_y;
    if ((rc = ensure_draws(c, tiles))) return rc;
    if (residuals) rp.rec_lo_off = rec_lo_off(c);
    if ((rc = ensure_ovf(c, (uint64_t)tiles * 64u))) return rc;
    rp.ovf_cap = c->ovf_cap;
    if ((rc = launch_seed(c, rp, 0, c->stream))) return rc;
    std::vector<float> h((size_t)tiles * REC_ITEM_FLOATS), lo;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(h.data(), c->recs[0], h.size() * sizeof(float), hipMemcpyDeviceToHost));
    if (residuals) {
        lo.resize(h.size());
        HIP_TRY(hipMemcpy(lo.data(), c->recs[0] + rp.rec_lo_off, lo.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    for (uint32_t t = 0; t < tiles; t++)
        for (uint32_t j = 0; j < 64; j++) {
            uint32_t tx = t % rp.tiles_x, ty = t / rp.tiles_x, pix = j >> 2, sub = j & 3;
            uint32_t px = tx * 4 + (pix & 3), py = ty * 4 + (pix >> 2);
            if (px >= c->W || py >= c->H) continue;
            const float *rec = &h[(size_t)t * REC_ITEM_FLOATS];
            const uint32_t lb = j * 4u;
            uint32_t a = float_as_uint(rec[rec_slot(lb, REC_HEAD)]);
            float *o = &host_out[(((size_t)py * c->W + px) * 4 + sub) * DRAWS_PER_PATH];
            if (residuals) {
                const float *rl = &lo[(size_t)t * REC_ITEM_FLOATS];
                for (uint32_t d = 0; d < (uint32_t)DRAWS_PER_PATH; d++) o[d] = rl[rec_slot(lb, 2 * a + d)];
                continue;
            }
            o[0] = rec[rec_slot(lb, REC_HEAD + 1)];
            o[1] = rec[rec_slot(lb, REC_HEAD + 2)];
            for (uint32_t d = 2; d < (uint32_t)DRAWS_PER_PATH; d++) o[d] = rec[rec_slot(lb, 2 * a + d)];
        }
    return drain_events(c);
}
int hr_debug_path_draws(hr_ctx *c, uint32_t sampling, float *host_out) { return path_draws_out(c, sampling, host_out, false, "hr_debug_path_draws"); }
int hr_debug_path_draw_residuals(hr_ctx *c, uint32_t sampling, float *host_out) { return path_draws_out(c, sampling, host_out, true, "hr_debug_path_draw_residuals"); }

int hr_debug_path_log(hr_ctx *c, uint32_t sampling, uint32_t *host_out) {
    // one sampling through the production pipeline — the seed kernel, then the LOG instantiation of trace_kernel (same traversal, same
    // path_advance) — with every path's radiance, ray count and event log written out instead of being accumulated
    if (!c || !host_out) return fail(HR_ERR_INVALID, "hr_debug_path_log: bad argument");
    if (!c->W) return fail(HR_ERR_NO_TARGET, "hr_debug_path_log: hr_set_resolution not called");
    if (!c->have_scene) return fail(HR_ERR_NO_SCENE, "hr_debug_path_log: no scene uploaded");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    RenderParams rp{};
    rp.width = c->W; rp.height = c->H; rp.tiles_x = (c->W + 3) / 4; rp.tiles_y = (c->H + 3) / 4;
    rp.sampling_begin = sampling; rp.stride = 1; rp.num_k = 1;
    rp.adv_den = c->adv_den; rp.leaf_den = c->leaf_den; rp.node_unroll = c->node_unroll; rp.kchunk = c->kchunk;
    rp.pad[0] = c->seed_prio;
    rp.nee_cull_off = ~c->nee_cull & 7u;
    const uint32_t tiles = rp.tiles_x * rp.tiles_y;
    if ((rc = ensure_draws(c, tiles))) return rc;
    rp.rec_lo_off = rec_lo_off(c);
    if ((rc = ensure_ovf(c, (uint64_t)tiles * 64u))) return rc;
    rp.ovf_cap = c->ovf_cap;
    if ((rc = launch_seed(c, rp, 0, c->stream))) return rc;
    const size_t words = (size_t)c->W * c->H * 4u * 8u;
    uint32_t *d_log = nullptr;
    const bool split = c->trace_mode == 1 && !c->rr_start;
    if (split && (rc = ensure_wf(c, (uint64_t)tiles * 64u))) return rc;
    HIP_TRY(hipMalloc((void **)&d_log, words * sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(d_log, 0, words * sizeof(uint32_t), c->stream);
    void *log_block = nullptr;
    if (e == hipSuccess && split) {
        // the split pipeline's LOG instantiation: the event log rides in two more state quads and a tag per ray slot, allocated for this call only
        WfQueues wq = c->wf;
        const size_t st_q = (size_t)wq.cap_paths * WF_SUBQ * sizeof(f4), tg_q = (size_t)wq.cap_rays * WF_SUBQ * sizeof(uint32_t);
        e = hipMalloc(&log_block, 2 * st_q + 2 * tg_q);
        if (e == hipSuccess) {
            char *b = (char *)log_block;
            for (int i = 0; i < 2; i++) { wq.st_f[i] = (f4 *)b; b += st_q; }
            for (int i = 0; i < 2; i++) { wq.tag[i] = (uint32_t *)b; b += tg_q; }
            rc = launch_split(c, rp, 0, nullptr, d_log, &wq);
            if (rc) { (void)hipFree(log_block); (void)hipFree(d_log); return rc; }
        }
    } else if (e == hipSuccess) {
        e = hipMemsetAsync(c->d_tile_counter, 0, sizeof(uint32_t), c->stream);
        if (e == hipSuccess) {
            const uint32_t kch = c->kchunk ? c->kchunk : TRACE_KCHUNK;
            const uint64_t units = (uint64_t)tiles * ((1u + kch - 1) / kch);
            const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)c->num_cus * c->trace_wgs, (units + TRACE_WAVES - 1) / TRACE_WAVES);
            dim3 g(grid), b(64 * TRACE_WAVES);
            if (c->precise) {
                if (c->dsc.qnodes) hipLaunchKernelGGL((trace_kernel<false, 3, true, false, true, true>), g, b, 0, c->stream, c->dsc, rp, c->recs[0], c->d_counters, c->d_tile_counter, d_log);
                else hipLaunchKernelGGL((trace_kernel<false, 3, false, false, true, true>), g, b, 0, c->stream, c->dsc, rp, c->recs[0], c->d_counters, c->d_tile_counter, d_log);
            } else if (c->dsc.qnodes) hipLaunchKernelGGL((trace_kernel<false, 3, true, false, true>), g, b, 0, c->stream, c->dsc, rp, c->recs[0], c->d_counters, c->d_tile_counter, d_log);
            else hipLaunchKernelGGL((trace_kernel<false, 3, false, false, true>), g, b, 0, c->stream, c->dsc, rp, c->recs[0], c->d_counters, c->d_tile_counter, d_log);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(host_out, d_log, words * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_log);
    if (log_block) (void)hipFree(log_block);
    if (e != hipSuccess) return fail(HR_ERR_DEVICE, "hr_debug_path_log: %s", hipGetErrorString(e));
    return drain_events(c);
}

int hr_debug_wf_profile(hr_ctx *c, uint32_t sampling, uint32_t num_k, double *ms_out, uint32_t *counts_out) {
    // one launch of the split pipeline with the chip to itself, an event between every two kernels: ms_out[0] = wf_start_kernel,
    // ms_out[2 s - 1] / ms_out[2 s] = traversal / shading kernel of step s = 1 .. WF_STEPS; counts_out[2 s] / [2 s + 1] = rays / live paths of step s
    if (!c || !ms_out || !counts_out || !num_k) return fail(HR_ERR_INVALID, "hr_debug_wf_profile: bad argument");
    if (!c->W) return fail(HR_ERR_NO_TARGET, "hr_debug_wf_profile: hr_set_resolution not called");
    if (!c->have_scene) return fail(HR_ERR_NO_SCENE, "hr_debug_wf_profile: no scene uploaded");
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    RenderParams rp{};
    rp.width = c->W; rp.height = c->H; rp.tiles_x = (c->W + 3) / 4; rp.tiles_y = (c->H + 3) / 4;
    rp.sampling_begin = sampling; rp.stride = 1; rp.num_k = num_k;
    rp.adv_den = c->adv_den; rp.leaf_den = c->leaf_den; rp.node_unroll = c->node_unroll;
    rp.pad[0] = c->seed_prio;
    rp.nee_cull_off = ~c->nee_cull & 7u;
    const uint32_t tiles = rp.tiles_x * rp.tiles_y;
    if ((rc = ensure_draws(c, (size_t)tiles * num_k))) return rc;
    rp.rec_lo_off = rec_lo_off(c);
    if ((rc = ensure_ovf(c, (uint64_t)tiles * 64u * num_k))) return rc;
    if ((rc = ensure_wf(c, (uint64_t)tiles * 64u * num_k))) return rc;
    rp.ovf_cap = c->ovf_cap;
    if ((rc = launch_seed(c, rp, 0, c->stream))) return rc;
    std::vector<hipEvent_t> marks;
    rc = launch_split(c, rp, 0, &marks);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (!rc && e == hipSuccess && marks.size() == 2u + 2u * WF_STEPS) {
        for (size_t i = 0; i + 1 < marks.size(); i++) { float ms = 0; (void)hipEventElapsedTime(&ms, marks[i], marks[i + 1]); ms_out[i] = ms; }
        std::vector<WfCounts> h((WF_STEPS + 2) * WF_SUBQ);
        e = hipMemcpy(h.data(), c->wf.counts, h.size() * sizeof(WfCounts), hipMemcpyDeviceToHost);
        for (uint32_t s = 0; s <= WF_STEPS; s++) {
            counts_out[2 * s] = counts_out[2 * s + 1] = 0;
            for (uint32_t k = 0; k < WF_SUBQ; k++) { counts_out[2 * s] += wf_rays(h[s * WF_SUBQ + k]); counts_out[2 * s + 1] += wf_paths(h[s * WF_SUBQ + k]); }
        }
    }
    for (hipEvent_t ev : marks) (void)hipEventDestroy(ev);
    if (rc) return rc;
    if (e != hipSuccess) return fail(HR_ERR_DEVICE, "hr_debug_wf_profile: %s", hipGetErrorString(e));
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

int hr_debug_trace(hr_ctx *c, uint32_t n, const float *rays, const float *shadow_len, float *out, int32_t *out_element) {
    if (!c || !rays || !out || !out_element || !n) return fail(HR_ERR_INVALID, "hr_debug_trace: bad argument");
    if (!c->have_scene) return fail(HR_ERR_NO_SCENE, "hr_debug_trace: no scene uploaded");
    HIP_TRY(hipSetDevice(c->device));
    float *d_rays = nullptr, *d_out = nullptr, *d_sl = nullptr;
    int32_t *d_el = nullptr;
    hipError_t e = hipMalloc((void **)&d_rays, (size_t)n * 6 * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, (size_t)n * 8 * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_el, (size_t)n * 4);
    if (e == hipSuccess && shadow_len) e = hipMalloc((void **)&d_sl, (size_t)n * 4);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)n * 6 * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && shadow_len) e = hipMemcpy(d_sl, shadow_len, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        RenderParams rp{};
        rp.leaf_den = c->leaf_den; rp.node_unroll = c->node_unroll;
        // the record format hr_render walks on this scene; timed with HIP events (hr_stats.debug_kernel_ms), counted with option "counters"
        EventPair ev{nullptr, nullptr};
        bool timed = hipEventCreate(&ev.a) == hipSuccess && hipEventCreate(&ev.b) == hipSuccess && hipEventRecord(ev.a, c->stream) == hipSuccess;
        const dim3 g((n + 63) / 64), b(64);
        if (c->counters) {
            if (c->dsc.qnodes) hipLaunchKernelGGL((trace_debug_kernel<true, true>), g, b, 0, c->stream, c->dsc, rp, n, d_rays, d_sl, d_out, d_el, c->d_counters);
            else hipLaunchKernelGGL((trace_debug_kernel<false, true>), g, b, 0, c->stream, c->dsc, rp, n, d_rays, d_sl, d_out, d_el, c->d_counters);
        } else if (c->dsc.qnodes) hipLaunchKernelGGL((trace_debug_kernel<true>), g, b, 0, c->stream, c->dsc, rp, n, d_rays, d_sl, d_out, d_el);
        else hipLaunchKernelGGL((trace_debug_kernel<false>), g, b, 0, c->stream, c->dsc, rp, n, d_rays, d_sl, d_out, d_el);
        e = hipGetLastError();
        timed = timed && e == hipSuccess && hipEventRecord(ev.b, c->stream) == hipSuccess;
        if (timed) { c->debug_events.push_back(ev); c->debug_launches++; }   // only a pair that was really recorded is ever queried
        else { if (ev.a) (void)hipEventDestroy(ev.a); if (ev.b) (void)hipEventDestroy(ev.b); }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * 8 * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_element, d_el, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_rays); (void)hipFree(d_out); (void)hipFree(d_el); (void)hipFree(d_sl);
    if (e != hipSuccess) return fail(HR_ERR_DEVICE, "hr_debug_trace: %s", hipGetErrorString(e));
    return drain_events(c);
}

}  // extern "C"
