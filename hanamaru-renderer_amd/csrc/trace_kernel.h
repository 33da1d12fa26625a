// The path-tracing megakernel and the two small debug kernels that share its per-lane code — included by hr_api.hip only (one translation unit: the kernels and the C ABI that launches them).
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.h"
#include "isaac_core.h"
#include "pt_core.h"

using namespace hr;

// ballot of a bool without HIP's int round trip (__ballot(int) compiles to v_cndmask 0/1 + v_cmp_ne; this is the lane mask itself)
__device__ __forceinline__ unsigned long long wave_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Persistent waves: a workgroup is 4 independent waves (single-wave workgroups cap residency at ~8 waves per CU);
// every wave pulls work units from a global counter until none are left.  A unit = one 4x4-pixel tile x up to TRACE_KCHUNK
// samplings of the batch (64 paths per sampling; small images with many samplings per launch still give every wave several units); finished lanes are refilled from the tile's path queue, and when that runs dry the wave pulls the
// next tile while its slow lanes are still working, so lanes only starve at the very end of a launch
// (measured before: with one tile per wave the mean box-phase pass had 19.6 of 64 lanes active).
// No barriers, no LDS.
static const int TRACE_WAVES = 4;
static const uint32_t TRACE_KCHUNK = 4;   // samplings per work unit (default of RenderParams::kchunk)

// wave-uniform statistics of the counters build (phase invocations / lanes served, wave-cycles per phase)
struct WaveStats {
    uint32_t ph[7];
    unsigned long long pc[4], tmark;
};
#define HR_PHASE_BEGIN(ws) do { if (CNT) (ws).tmark = __builtin_readcyclecounter(); } while (0)
#define HR_PHASE_END(ws, i) do { if (CNT) (ws).pc[i] += __builtin_readcyclecounter() - (ws).tmark; } while (0)

// Phase C of the megakernel — the production traversal — as one function, so that the unit-level entry point hr_debug_trace walks
// exactly the code the renderer walks (trace_debug_kernel below).  Traversal as two well-filled phases.  Box phase: lanes walk
// nodes until 1/leaf_den of the traversing lanes have parked a leaf; leaf phase: those lanes test their primitives together.
// The whole of C is left as soon as 1/adv_den of the live lanes wait for phase A (adv_den = 0: only when no lane traverses).
// QN: the tree is walked on the 16-byte quantised nodes (the default for host- and device-built trees); false: on the 32-byte
// fp32 records of the same tree (option quant_nodes = 0).  A template parameter, not a branch: each form keeps only its own
// per-ray constants in registers.
// P: what a lane holds — the megakernel's Path, or the split pipeline's TravLane (wf_core.h): `ray`, `ts` and a shadow_early_out(P &).
template <bool CNT, bool QN, class P>
__device__ __forceinline__ void traverse_wave(const Scene &sc, const RenderParams &rp, P &p, const bool active, const uint32_t n_active, const uint32_t adv_den,
                                              const uint32_t leaf_den, LaneCounters &lc, WaveStats &ws, uint32_t &tick, const uint32_t boost_mask) {
    for (;;) {
        const bool trav = active && !trace_done(p.ts);
        const uint32_t n_trav = (uint32_t)__popcll(wave_ballot(trav));
        if (!n_trav || (n_active - n_trav) * adv_den >= n_active) break;
        // lanes allowed to be still walking when the leaf phase starts
        const uint32_t park = leaf_den == 2u ? (n_trav + 1u) >> 1 : (n_trav + leaf_den - 1u) / leaf_den;   // (no scalar division for the default)
        const uint32_t walk_max = n_trav - park;
        HR_PHASE_BEGIN(ws);
        // Box phase above the seed kernel's producer waves (which then run at priority 0): their ahead pass is not urgent, a
        // box pass is the trace kernel's critical loop.  Decided by the priority governor (device_scene.h GovDev, hr_api.hip
        // governor_kernel) from the two kernels' own time stamps: it pays when the trace kernel is the slower of the pair and costs
        // ~1 - 3 % when the seed kernel is.  boost_mask bits 0-3: which of every four box phases of a wave run boosted (levels 3 and 4:
        // all of them); bit 4 adds the leaf phase (another +1 - 2 % where the trace kernel is far behind).
        const bool boost_box = (boost_mask >> (tick & 3u)) & 1u;
        tick++;
        if (boost_box) __builtin_amdgcn_s_setprio(1);
        // a lane may keep walking with ONE leaf parked (trace_node<SPEC>); it stops at the second.  The loop is written with its
        // wave-uniform test at the bottom: as `for (;;) { if (n_go <= walk_max) break; if (go) ... }` the compiler folds the exit
        // into the lane mask of `if (go)` and copies the live-out walk state every pass.
        bool go = trav && trace_can_walk(p.ts);
        uint32_t n_go = (uint32_t)__popcll(wave_ballot(go));
        while (n_go > walk_max) {
            if (CNT) { ws.ph[2]++; ws.ph[3] += n_go; }
            if (go) {
                if (QN) {
                    trace_qnode<CNT, true>(sc, p.ray, p.ts, &lc);
                    if (rp.node_unroll > 1u && trace_can_walk(p.ts)) trace_qnode<CNT, true>(sc, p.ray, p.ts, &lc);
                } else {
                    trace_node<CNT, true>(sc, p.ray, p.ts, &lc);
                    if (rp.node_unroll > 1u && trace_can_walk(p.ts)) trace_node<CNT, true>(sc, p.ray, p.ts, &lc);
                }
            }
            go = go && trace_can_walk(p.ts);
            n_go = (uint32_t)__popcll(wave_ballot(go));
        }
        if (boost_box) __builtin_amdgcn_s_setprio(0);
        HR_PHASE_END(ws, 2);
        HR_PHASE_BEGIN(ws);
        if (boost_mask & 16u) __builtin_amdgcn_s_setprio(1);   // top level: the leaf phase too
        if (CNT) {
            uint32_t n = (uint32_t)__popcll(wave_ballot(trav && p.ts.leaf != 0));
            if (n) { ws.ph[4]++; ws.ph[5] += n; }
        }
        if (trav && p.ts.leaf != 0) {
            trace_leaf_next<CNT>(sc, p.ray, p.ts, &lc);   // the older parked leaf; the newer one (if any) moves up
            shadow_early_out(p);
        }
        if (boost_mask & 16u) __builtin_amdgcn_s_setprio(0);
        HR_PHASE_END(ws, 3);
    }
}

// one atomic per counter per wave
template <bool CNT>
__device__ __forceinline__ void flush_counters(Counters *cnt, uint32_t lane, uint32_t npaths, const LaneCounters &lc, const WaveStats &ws) {
    if (!CNT) return;
    unsigned long long v[7] = {npaths, lc.rays, lc.node_tests, lc.tri_tests, lc.sphere_tests, lc.cuboid_tests, lc.shadow_culled};
    for (int i = 0; i < 7; i++) {
        unsigned long long x = v[i];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        v[i] = x;
    }
    if (lane == 0) {
        atomicAdd(&cnt->paths, v[0]); atomicAdd(&cnt->rays, v[1]); atomicAdd(&cnt->node_tests, v[2]);
        atomicAdd(&cnt->tri_tests, v[3]); atomicAdd(&cnt->sphere_tests, v[4]); atomicAdd(&cnt->cuboid_tests, v[5]); atomicAdd(&cnt->shadow_culled, v[6]);
        atomicAdd(&cnt->shade_calls, (unsigned long long)ws.ph[0]); atomicAdd(&cnt->shade_lanes, (unsigned long long)ws.ph[1]);
        atomicAdd(&cnt->box_passes, (unsigned long long)ws.ph[2]); atomicAdd(&cnt->box_lanes, (unsigned long long)ws.ph[3]);
        atomicAdd(&cnt->leaf_calls, (unsigned long long)ws.ph[4]); atomicAdd(&cnt->leaf_lanes, (unsigned long long)ws.ph[5]);
        atomicAdd(&cnt->outer_iters, (unsigned long long)ws.ph[6]);
        for (int i = 0; i < 4; i++) atomicAdd(&cnt->phase_cycles[i], ws.pc[i]);
    }
}

// LOG (hr_debug_path_log only): every path also leaves its event log (pt_core.h PathLog) and its own radiance in plog — eight 32-bit
// words per path, indexed ((y * W + x) * 4 + sub-sample) for the launch's first sampling: {r, g, b (float bits), rays, ev low, ev high, ev9, hash}.
// The same kernel, the same path_advance: what is logged is what hr_render computes.
// PREC (option precise_shading): path_advance shades in f64 (prec_core.h); instantiated for 128 VGPRs (MINW 4).
template <bool CNT, int MINW, bool QN, bool RR = false, bool LOG = false, bool PREC = false>
__global__ __launch_bounds__(64 * TRACE_WAVES, MINW) void trace_kernel(Scene sc, RenderParams rp, float *recs, Counters *cnt, uint32_t *tile_counter, uint32_t *plog = nullptr) {
    // the wave budget (device_scene.h GovDev::budget; debug option trace_budget pins it): surplus workgroups leave before they touch anything
    // ONE budget per launch: the governor of the launch before may store a new one while this launch's workgroups are still starting, so the
    // first wave to arrive fixes what it read in bud[slot] (compare-and-swap from "unset") and every other wave takes that — the governor then
    // judges the launch by the budget all of its workgroups really obeyed.  Stored + 1: 0 = unset (governor_kernel resets it).
    {
        uint32_t budget = rp.wg_budget;
        if (rp.gov) {
            uint32_t fixed = 0;
            if ((threadIdx.x & 63u) == 0u) {
                const uint32_t want = (budget ? budget : __hip_atomic_load(&rp.gov->budget, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1u;
                const uint32_t prev = atomicCAS(&rp.gov->bud[rp.gov_slot], 0u, want);
                fixed = prev ? prev : want;
            }
            budget = (uint32_t)__builtin_amdgcn_readfirstlane((int)fixed) - 1u;
        }
        if (budget && blockIdx.x >= budget) return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tiles = rp.tiles_x * rp.tiles_y;
    LaneCounters lc = {0, 0, 0, 0, 0, 0};
    uint32_t npaths = 0;
    WaveStats ws = {{0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0}, 0};
    const uint32_t kchunk = rp.kchunk ? rp.kchunk : TRACE_KCHUNK;
    const uint32_t nchunks = (rp.num_k + kchunk - 1u) / kchunk;
    // the launch's last tiles go out one sampling at a time: with ~20 units of 256 paths per wave the waves finish up to a unit apart, and a
    // long-path scene feels that tail; the bulk keeps the larger units (four samplings of a tile share their texels)
    const uint32_t tail_tiles = rp.tail_div ? tiles / rp.tail_div : 0u, bulk_units = (tiles - tail_tiles) * nchunks, units = bulk_units + tail_tiles * rp.num_k;
    uint32_t total = 0, cur_k0 = 0;          // wave-uniform: paths in the current unit (slot q = (k - cur_k0) * 64 + j), its first sampling
    const size_t tile_stride = (size_t)rp.num_k * REC_ITEM_FLOATS;   // floats of hand-off records per tile
    uint32_t cur_tile = 0, next = 0;          // wave-uniform: the tile of the unit being handed out and its queue head
    bool exhausted = false;
    Path p;
    p.q = PATH_IDLE;
    p.tile = 0;
    p.ts.cur = NODE_END; p.ts.leaf = 0; p.ts.leaf2 = 0;
    PathLog lg;
    plog_reset(lg);
    const uint32_t adv_den = rp.adv_den ? rp.adv_den : 2u;
    const uint32_t leaf_den = rp.leaf_den ? rp.leaf_den : 2u;
    uint32_t tick = threadIdx.x >> 6;   // wave-uniform count of box phases (the boost's duty cycle); the waves of a workgroup start out of step
    // the priority governor (device_scene.h GovDev): this kernel's level is the one in force when the wave starts; the first wave to
    // start and the last to finish give the kernel's time
    uint32_t boost_mask = rp.trace_boost;
    if (rp.gov) {
        const int32_t lvl = __hip_atomic_load(&rp.gov->level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        boost_mask = gov_trace_mask(lvl);
        if (lane == 0) {
            atomicMin(&rp.gov->t0[1][rp.gov_slot], (unsigned long long)__builtin_amdgcn_s_memrealtime());
            if (blockIdx.x == 0 && threadIdx.x == 0) rp.gov->lvl[1][rp.gov_slot] = (uint32_t)lvl;
        }
    }

    for (;;) {
        // ---- A: lanes whose ray is complete: shade / NEE / next ray (or the path ends)
        if (CNT) {
            uint32_t n = (uint32_t)__popcll(wave_ballot(p.q != PATH_IDLE && trace_done(p.ts)));
            ws.ph[6]++;
            if (n) { ws.ph[0]++; ws.ph[1] += n; }
        }
        HR_PHASE_BEGIN(ws);
        if (p.q != PATH_IDLE && trace_done(p.ts)) {
            if (path_advance<CNT, RR, LOG, PREC>(sc, rp, p, recs + (size_t)p.tile * tile_stride, &lc, rp.rr_start, rp.sampling_begin * 64u + rp.stride, &lg)) {
                // A finished path leaves its radiance in its own hand-off record (quad 0: the draws there have been consumed), and
                // accumulate_kernel below sums the records of a pixel into the accumulator behind this kernel.  Until round 3 the path
                // added its radiance straight into the accumulator with three agent-scope atomics (several waves, on other XCDs,
                // hold samplings of the same pixel): 100 M atomics per launch were 9 % of this kernel's time and cost the seed
                // kernel beside it another 1.5 % (measured by leaving them out).  The store is one 16-byte write per path, the sum
                // a 0.1-ms pass over 0.5 GB — and the summation order is fixed now: renders are bit-reproducible.
                *reinterpret_cast<f4 *>(recs + (size_t)p.tile * tile_stride + rec_slot(path_draw_base(p), 0)) = f4{p.accum.x, p.accum.y, p.accum.z, 0.0f};
                if (LOG && ((p.q >> 6) & 63u) == 0u) {   // the launch's first sampling
                    uint32_t px, py, sub;
                    tile_lane_pixel(rp, p.tile, p.q & 63u, px, py, sub);
                    uint32_t *o = plog + (((size_t)py * rp.width + px) * 4u + sub) * 8u;
                    o[0] = float_as_uint(p.accum.x); o[1] = float_as_uint(p.accum.y); o[2] = float_as_uint(p.accum.z); o[3] = lg.rays;
                    o[4] = (uint32_t)lg.ev; o[5] = (uint32_t)(lg.ev >> 32); o[6] = lg.ev9; o[7] = lg.hash;
                }
                p.q = PATH_IDLE;
            }
        }
        HR_PHASE_END(ws, 0);
        // ---- B: refill idle lanes (ballot + prefix rank = live-lane compaction); pull a new tile when the queue is dry
        HR_PHASE_BEGIN(ws);
        unsigned long long idle = wave_ballot(p.q == PATH_IDLE);
        if (idle) {
            if (next >= total && !exhausted) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(tile_counter, 1u);
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                if (t >= units) exhausted = true;
                else if (t < bulk_units) {
                    cur_tile = t / nchunks;
                    cur_k0 = (t - cur_tile * nchunks) * kchunk;
                    total = 64u * (rp.num_k - cur_k0 < kchunk ? rp.num_k - cur_k0 : kchunk);
                    next = 0;
                } else {
                    const uint32_t u = t - bulk_units, tt = u / rp.num_k;
                    cur_tile = tiles - tail_tiles + tt;
                    cur_k0 = u - tt * rp.num_k;
                    total = 64u;
                    next = 0;
                }
            }
            if (next < total) {
                uint32_t q = next + lane_rank(idle);
                if (p.q == PATH_IDLE && q < total) {
                    uint32_t k = cur_k0 + (q >> 6), j = q & 63u, px, py, sub;
                    tile_lane_pixel(rp, cur_tile, j, px, py, sub);
                    if (px < rp.width && py < rp.height) {
                        p.q = (k << 6) | j;   // slot inside the tile's batch (bits 0-5: lane of the tile -> pixel, sub-sample)
                        p.tile = cur_tile;
                        path_start(sc, rp, p, px, py, sub, recs + (size_t)cur_tile * tile_stride);
                        if (LOG) plog_reset(lg);
                        npaths++;
                    }
                }
                next += (uint32_t)__popcll(idle);
            }
        }
        HR_PHASE_END(ws, 1);
        const bool active = p.q != PATH_IDLE;
        const uint32_t n_active = (uint32_t)__popcll(wave_ballot(active));
        if (!n_active) {
            if (exhausted) break;
            continue;
        }
        // ---- C: traversal
        traverse_wave<CNT, QN>(sc, rp, p, active, n_active, adv_den, leaf_den, lc, ws, tick, boost_mask);
    }
    flush_counters<CNT>(cnt, lane, npaths, lc, ws);
    if (rp.gov && lane == 0) atomicMax(&rp.gov->t1[1][rp.gov_slot], (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// The radiance of a launch into the accumulator (renderer.rs:33-38: acc += the 2x2 sub-sample sum of calc_pixel, for every sampling
// of the launch): one wave per tile, lane j = (pixel of the tile, sub-sample) as everywhere; the lane adds up its slot of the num_k
// records the trace kernel left (quad 0 of the hand-off record), the four sub-samples of a pixel are neighbouring lanes, and the lane
// of sub-sample 0 adds the sum to the pixel — plain loads and stores, nothing else touches the accumulator while this runs.
// A path's radiance enters the image through path_radiance_in(): a NaN adds nothing and an overflow adds 1e30 ("white" after Reinhard, and 4,096
// of them still sum to a finite fp32).  Neither occurs in the reference's scenes (4,096-sampling soak of all of them: DESIGN.md §6.5); in random
// scenes a near-mirror GGX from a roughness map's ~0 texel can overflow fp32's D term where the reference's f64 carries 1e29, and one NaN of an
// approximate reciprocal was seen in 1.2e11 paths — one such path must not cost a pixel (and, through the bilateral filter, its neighbours).
__device__ __forceinline__ float path_radiance_in(float x) { return x == x ? fminf(fmaxf(x, -1e30f), 1e30f) : 0.0f; }
__global__ __launch_bounds__(256) void accumulate_kernel(RenderParams rp, const float *__restrict__ recs, float *__restrict__ accum) {
    const uint32_t lane = threadIdx.x & 63u, tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= rp.tiles_x * rp.tiles_y) return;
    uint32_t px, py, sub;
    tile_lane_pixel(rp, tile, lane, px, py, sub);
    const bool valid = px < rp.width && py < rp.height;      // (the records of lanes beyond the image's edge were never written)
    const f4 *src = reinterpret_cast<const f4 *>(recs + (size_t)tile * rp.num_k * REC_ITEM_FLOATS) + lane;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t k = 0; k < rp.num_k; k++) {
        const f4 v = valid ? src[(size_t)k * (REC_ITEM_FLOATS / 4u)] : f4{0.0f, 0.0f, 0.0f, 0.0f};
        r += path_radiance_in(v.x); g += path_radiance_in(v.y); b += path_radiance_in(v.z);
    }
    r += __shfl_xor(r, 1); g += __shfl_xor(g, 1); b += __shfl_xor(b, 1);
    r += __shfl_xor(r, 2); g += __shfl_xor(g, 2); b += __shfl_xor(b, 2);
    if (valid && sub == 0u) {
        float *dst = accum + ((size_t)py * rp.width + px) * 3;
        dst[0] += r; dst[1] += g; dst[2] += b;
    }
}

// hr_debug_trace: closest-hit / shadow queries through the PRODUCTION traversal — traverse_wave on the record format the renderer
// walks, box and leaf phases, two parked leaves, closest-hit culling, and for shadow queries (shadow_len > 0) the search limit of
// nee_setup and shadow_early_out.  One wave = 64 rays; lanes whose walk is done idle, as lanes waiting for phase A do in the megakernel.
// CNT: the counters build (option "counters"): node / primitive tests and lanes per box pass of the queries — what tools/coherence_probe.py reads
template <bool QN, bool CNT = false>
__global__ __launch_bounds__(64) void trace_debug_kernel(Scene sc, RenderParams rp, uint32_t n, const float *__restrict__ rays, const float *__restrict__ shadow_len,
                                                         float *__restrict__ out, int32_t *__restrict__ out_elem, Counters *cnt = nullptr) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    const bool active = i < n;
    const uint32_t j = active ? i : 0u;
    Path p;
    p.q = active ? 0u : PATH_IDLE;
    p.tile = 0; p.st = 1u;
    ray_set(p.ray, v3(rays[j * 6], rays[j * 6 + 1], rays[j * 6 + 2]), v3(rays[j * 6 + 3], rays[j * 6 + 4], rays[j * 6 + 5]));
    ray_quantise(sc, p.ray);
    const float sl = shadow_len ? shadow_len[j] : 0.0f;
    p.shadow_len = sl;
    if (sl > 0.0f) { trace_begin(p.ts, sl + 0.03f, p.ray.start); p.st |= 16u; }   // nee_setup's search limit, shadow phase
    else trace_begin(p.ts, T_INF, p.ray.start);
    if (!active) { p.ts.cur = NODE_END; p.ts.leaf = 0; }
    LaneCounters lc = {0, 0, 0, 0, 0, 0};
    WaveStats ws = {{0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0}, 0};
    const uint32_t n_active = (uint32_t)__popcll(wave_ballot(active));
    uint32_t tick = 0;
    traverse_wave<CNT, QN>(sc, rp, p, active, n_active, 0u, rp.leaf_den ? rp.leaf_den : 2u, lc, ws, tick, rp.trace_boost);
    if (CNT) { if (active) lc.rays++; flush_counters<CNT>(cnt, threadIdx.x & 63u, active ? 1u : 0u, lc, ws); }
    if (!active) return;
    float *o = out + (size_t)i * 8;
    int32_t elem = -1;
    if (p.ts.prim >= 0) {
        Surf s;
        hit_surface(sc, p.ray, p.ts, true, s);
        elem = s.elem;
        o[0] = 1.0f; o[1] = p.ts.t; o[2] = s.pos.x; o[3] = s.pos.y; o[4] = s.pos.z; o[5] = s.n.x; o[6] = s.n.y; o[7] = s.n.z;
    } else {
        o[0] = 0.0f; o[1] = p.ts.t;
        for (int k = 2; k < 8; k++) o[k] = 0.0f;
    }
    out_elem[i] = elem;
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

// DebugRenderer (renderer.rs:101-146) through the production traversal: one wave per 4x4-pixel tile, one lane per (pixel, sub-sample)
// as in the megakernel, pinhole rays, no RNG, no seed kernel next to it.  Depth mode (2) is the traversal-only workload of bench.py:
// camera rays, closest hits, nothing shaded.  The four sub-samples of a pixel sit in neighbouring lanes and are summed with two
// lane exchanges (renderer.rs:48-60 adds them in order; the fp32 sum differs from that order by an ulp at most).
template <bool CNT, bool QN>
__global__ __launch_bounds__(64 * TRACE_WAVES) void debug_render_kernel(Scene sc, RenderParams rp, int mode, float *__restrict__ accum, Counters *cnt) {
    const uint32_t lane = threadIdx.x & 63u, tile = blockIdx.x * TRACE_WAVES + (threadIdx.x >> 6);
    if (tile >= rp.tiles_x * rp.tiles_y) return;
    uint32_t px, py, sub;
    tile_lane_pixel(rp, tile, lane, px, py, sub);
    const bool active = px < rp.width && py < rp.height;
    LaneCounters lc = {0, 0, 0, 0, 0, 0};
    WaveStats ws = {{0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0}, 0};
    Path p;
    p.q = active ? lane : PATH_IDLE;
    p.tile = tile; p.st = 1u; p.shadow_len = 0.0f;
    debug_camera_ray(sc, rp, active ? px : 0u, active ? py : 0u, sub, p.ray);
    ray_quantise(sc, p.ray);
    trace_begin(p.ts, T_INF, p.ray.start);
    if (!active) { p.ts.cur = NODE_END; p.ts.leaf = 0; }
    const uint32_t n_active = (uint32_t)__popcll(wave_ballot(active));
    const uint32_t leaf_den = rp.leaf_den ? rp.leaf_den : 2u;
    uint32_t tick = 0;
    traverse_wave<CNT, QN>(sc, rp, p, active, n_active, 0u, leaf_den, lc, ws, tick, rp.trace_boost);
    if (CNT && active) lc.rays++;
    V3f val = v3(0, 0, 0), lit = v3(0, 0, 0);
    Ray sh = p.ray;
    const bool more = active && debug_primary(sc, p.ray, p.ts, mode, val, lit, sh);
    if (wave_ballot(more)) {   // Shading mode: the shadow ray towards the fixed light, any closest hit darkens
        p.ray = sh;
        ray_quantise(sc, p.ray);
        trace_begin(p.ts, T_INF, p.ray.start);
        if (!more) { p.ts.cur = NODE_END; p.ts.leaf = 0; }
        traverse_wave<CNT, QN>(sc, rp, p, more, (uint32_t)__popcll(wave_ballot(more)), 0u, leaf_den, lc, ws, tick, rp.trace_boost);
        if (CNT && more) lc.rays++;
        if (more) val = val + lit * (p.ts.prim >= 0 ? 0.5f : 1.0f);
    }
    val.x += __shfl_xor(val.x, 1); val.y += __shfl_xor(val.y, 1); val.z += __shfl_xor(val.z, 1);
    val.x += __shfl_xor(val.x, 2); val.y += __shfl_xor(val.y, 2); val.z += __shfl_xor(val.z, 2);
    if (active && sub == 0u) {
        float *o = accum + ((size_t)py * rp.width + px) * 3;
        o[0] += val.x; o[1] += val.y; o[2] += val.z;
    }
    flush_counters<CNT>(cnt, lane, (uint32_t)active, lc, ws);
}
