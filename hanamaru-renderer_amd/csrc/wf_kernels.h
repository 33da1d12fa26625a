// The split pipeline's kernels (trace_mode 1; per-lane code: wf_core.h) — included by hr_api.hip only.
//   wf_start_kernel     camera rays of every path of the launch (path_start)            -> step 1's rays and states
//   wf_traverse_kernel  walks a step's rays, nothing else: <= 64 VGPRs, 8 waves per SIMD  -> hits
//   wf_shade_kernel     NEE contributions of the iteration before, then the main hit     -> radiance (path ends) or the next step's rays and state
// A launch is start + WF_STEPS x (traverse, shade): iteration i's main ray and iteration i - 1's shadow rays are step i's rays.
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.h"
#include "trace_kernel.h"
#include "wf_core.h"

using namespace hr;

static const uint32_t WF_STEPS = 10;   // iterations 1 .. 9 (renderer.rs:174) + the step that collects iteration 9's shadow rays
// The queues are WF_SUBQ independent sub-queues: a path lives in sub-queue (item mod WF_SUBQ) (item = tile x sampling, its hand-off record's
// index) from its camera ray to its end.  Why: every queue append is an atomic add on a counter, device-scope atomics on ONE address run at
// ~10 ns apiece (measured: a million of them per step made the shading kernel 4x slower than its memory traffic), and 64 counters in 64
// cache lines run side by side.  Interleaved items keep the sub-queues statistically level; capacity is the strict worst case per sub-queue.
static const uint32_t WF_SUBQ = 64;
struct alignas(64) WfCounts {
    unsigned long long alloc;   // low word: rays of the step in this sub-queue, high word: live paths (one 64-bit atomic appends both)
    uint32_t head;              // the traversal kernel's queue head
    uint32_t pad[13];
};
HD uint32_t wf_rays(const WfCounts &c) { return (uint32_t)c.alloc; }
HD uint32_t wf_paths(const WfCounts &c) { return (uint32_t)(c.alloc >> 32); }

struct WfQueues {
    f4 *ray_a[2], *ray_b[2];            // [step & 1][sub-queue * cap_rays + slot]: {o, len}, {d, w}
    WfHitRec *hits;                     // same index, of the step being worked on
    f4 *st_a[2], *st_b[2], *st_c[2];    // [step & 1][sub-queue * cap_paths + position]: live-path state
    f4 *st_d[2], *st_e[2];              // precise shading: the residuals (o_lo, d_lo) of the path's main ray (wf_core.h wf_surface_f64)
    f4 *st_f[2];                        // path log (hr_debug_path_log): {ev low, ev high, ev9, hash}; the ray count rides in st_b.w
    uint32_t *tag[2];                   // path log: the emitter a shadow ray aims at, per ray slot
    WfCounts *counts;                   // [WF_STEPS + 2][WF_SUBQ]
    uint32_t cap_rays, cap_paths;       // per sub-queue
};

// The queues are written once and read once or twice, gigabytes per launch: streamed past the caches (nontemporal), so that they do not
// evict what lives there — the BVH in every XCD's L2, and the seed kernel's state ring next door.
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 ldq(const f4 *p) { const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p)); return f4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void stq(f4 *p, const f4 &a) { f4v v = {a.x, a.y, a.z, a.w}; __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(p)); }
__device__ __forceinline__ WfHitRec ldh(const WfHitRec *p) { const f4 v = ldq(reinterpret_cast<const f4 *>(p)); WfHitRec h; h.t = v.x; h.pt = float_as_uint(v.y); h.u = v.z; h.v = v.w; return h; }
__device__ __forceinline__ void sth(WfHitRec *p, const WfHitRec &h) { stq(reinterpret_cast<f4 *>(p), f4{h.t, uint_as_float(h.pt), h.u, h.v}); }


// inclusive prefix sum over the wave's lanes
__device__ __forceinline__ uint32_t wave_scan_inclusive(uint32_t v, uint32_t lane) {
    for (uint32_t off = 1; off < 64u; off <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)v, off);
        if (lane >= off) v += u;
    }
    return v;
}

// one wave per item (tile x sampling of the launch = 64 paths), lane j = (pixel of the tile, sub-sample); the waves of sub-queue k take its
// items k, k + 64, ... in turn
template <bool PREC>
__global__ __launch_bounds__(256) void wf_start_kernel(Scene sc, RenderParams rp, float *recs, WfQueues q) {
    const uint32_t lane = threadIdx.x & 63u;
    // the governor (device_scene.h GovDev): the launch's trace side starts here — its first thread stamps the start, records the level and
    // fixes the wave budget the launch's traversal kernels obey (one value per launch, as the megakernel's first wave does)
    if (rp.gov && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicMin(&rp.gov->t0[1][rp.gov_slot], (unsigned long long)__builtin_amdgcn_s_memrealtime());
        rp.gov->lvl[1][rp.gov_slot] = (uint32_t)__hip_atomic_load(&rp.gov->level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rp.gov->bud[rp.gov_slot] = (rp.wg_budget ? rp.wg_budget : __hip_atomic_load(&rp.gov->budget, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1u;
    }
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), waves = gridDim.x * 4u, items = rp.tiles_x * rp.tiles_y * rp.num_k;
    const uint32_t k = w % WF_SUBQ, per = waves / WF_SUBQ;     // (the grid is a multiple of WF_SUBQ waves)
    WfCounts *cn = q.counts + (size_t)1 * WF_SUBQ + k;
    const uint32_t rbase0 = k * q.cap_rays, pbase0 = k * q.cap_paths;
    for (uint32_t item = k + (w / WF_SUBQ) * WF_SUBQ; item < items; item += per * WF_SUBQ) {
        const uint32_t tile = item / rp.num_k, ks = item - tile * rp.num_k;
        uint32_t px, py, sub;
        tile_lane_pixel(rp, tile, lane, px, py, sub);
        const bool valid = px < rp.width && py < rp.height;
        Path p;
        p.q = (ks << 6) | lane;
        p.tile = tile;
        if (valid) path_start(sc, rp, p, px, py, sub, recs + (size_t)tile * rp.num_k * REC_ITEM_FLOATS);
        const unsigned long long m = wave_ballot(valid);
        if (!m) continue;
        const uint32_t cnt = (uint32_t)__popcll(m);
        unsigned long long at = 0;
        if (lane == 0) at = atomicAdd(&cn->alloc, ((unsigned long long)cnt << 32) | cnt);
        const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)at), pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(at >> 32));
        if (valid) {
            const uint32_t r = lane_rank(m), slot = rb + r, pos = pb + r;
            stq(&q.ray_a[1][rbase0 + slot], f4{p.ray.o.x, p.ray.o.y, p.ray.o.z, WF_MAIN_RAY});
            stq(&q.ray_b[1][rbase0 + slot], f4{p.ray.d.x, p.ray.d.y, p.ray.d.z, 0.0f});
            stq(&q.st_a[1][pbase0 + pos], f4{uint_as_float(item * 64u + lane), uint_as_float(wf_st(1u, true, (p.q >> 12) & 15u, 0u)), uint_as_float(slot), 1.0f});
            stq(&q.st_b[1][pbase0 + pos], f4{0.0f, 0.0f, 0.0f, 0.0f});
            stq(&q.st_c[1][pbase0 + pos], f4{1.0f, 1.0f, 1.0f, 0.0f});
            if (PREC) {   // what rounding the f64 camera ray took away: path_start has just parked it in the path's record
                V3f fo, fd;
                ray_fix_load(recs + (size_t)tile * rp.num_k * REC_ITEM_FLOATS, path_draw_base(p), (p.q >> 12) & 15u, fo, fd);
                stq(&q.st_d[1][pbase0 + pos], f4{fo.x, fo.y, fo.z, 0.0f});
                stq(&q.st_e[1][pbase0 + pos], f4{fd.x, fd.y, fd.z, 0.0f});
            }
            if (q.st_f[1]) stq(&q.st_f[1][pbase0 + pos], f4{0.0f, 0.0f, 0.0f, uint_as_float(0x811c9dc5u)});   // plog_reset
        }
    }
}

// Persistent waves over the step's ray queues: a lane holds one ray and its walk; lanes whose walk is done write their hit and take the next
// ray (ballot + prefix rank).  A wave drains its home sub-queue in chunks (one atomic per chunk, as the megakernel's work units), then helps
// with the others in turn.  The walk is traverse_wave — the megakernel's phase C, unchanged.
template <bool CNT, bool QN>
__global__ __launch_bounds__(256, 8) void wf_traverse_kernel(Scene sc, RenderParams rp, WfQueues q, uint32_t step, Counters *cnt) {
    const uint32_t lane = threadIdx.x & 63u;
    // the wave budget the launch's first kernel fixed (workgroups that stay; 0 = all) and the priority level its kernels started at
    uint32_t blocks = gridDim.x, boost_mask = rp.trace_boost;
    if (rp.gov) {
        const uint32_t b = rp.gov->bud[rp.gov_slot];
        if (b > 1u && b - 1u < blocks) blocks = (b - 1u) / 16u * 16u;      // whole multiples of WF_SUBQ waves
        if (blockIdx.x >= blocks) return;
        boost_mask = gov_trace_mask((int32_t)rp.gov->lvl[1][rp.gov_slot]);
    }
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
    WfCounts *cn = q.counts + (size_t)step * WF_SUBQ;
    const f4 *ra = q.ray_a[step & 1u], *rb = q.ray_b[step & 1u];
    LaneCounters lc = {0, 0, 0, 0, 0, 0};
    WaveStats ws = {{0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0}, 0};
    const uint32_t adv_den = rp.adv_den ? rp.adv_den : 2u, leaf_den = rp.leaf_den ? rp.leaf_den : 2u;
    uint32_t tick = 0;
    const uint32_t NONE = 0xffffffffu;
    uint32_t slot = NONE;
    // wave-uniform: the sub-queue being drained, its size, this wave's chunk of it; `avail` = sub-queues that still had rays when last looked at
    // (lane l looks at sub-queue l: one load for all 64 heads)
    uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w % WF_SUBQ)), n = 0, next = 0, end = 0;
    const uint32_t my_n = wf_rays(cn[lane]);
    unsigned long long avail = wave_ballot(my_n != 0u);
    // chunk size: 256 rays, less when the step is small (every wave should see several chunks), never less than a wave
    uint32_t chunk = (uint32_t)__builtin_amdgcn_readlane((int)my_n, (int)k) / (blocks * 4u / WF_SUBQ * 4u + 1u);
    chunk = chunk > 256u ? 256u : chunk < 64u ? 64u : (chunk & ~63u);
    TravLane p;
    p.ts.cur = NODE_END; p.ts.leaf = 0; p.ts.leaf2 = 0;
    for (;;) {
        const unsigned long long idle = wave_ballot(slot == NONE);
        if (idle && avail) {
            while (next >= end && avail) {
                // the next sub-queue with rays left, from the home sub-queue on
                const unsigned long long rot = (avail >> k) | (k ? avail << (64u - k) : 0ull);
                k = (k + (uint32_t)__builtin_ctzll(rot)) % WF_SUBQ;
                n = (uint32_t)__builtin_amdgcn_readlane((int)my_n, (int)k);
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(&cn[k].head, chunk);
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                if (t < n) { next = t; end = t + chunk < n ? t + chunk : n; }
                else {   // ran dry meanwhile: look at all the heads again
                    const uint32_t h = __hip_atomic_load(&cn[lane].head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    avail = wave_ballot(h < my_n);
                }
            }
            if (next < end) {
                const uint32_t e = next + lane_rank(idle);
                if (slot == NONE && e < end) {
                    const uint32_t s = k * q.cap_rays + e;
                    const f4 a = ldq(&ra[s]), b = ldq(&rb[s]);
                    slot = s;
                    wf_lane_begin(sc, p, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), a.w);
                }
                next += (uint32_t)__popcll(idle);
            }
        }
        const bool exhausted = !avail;
        const bool active = slot != NONE;
        const uint32_t n_active = (uint32_t)__popcll(wave_ballot(active));
        if (!n_active) {
            if (exhausted) break;
            continue;
        }
        traverse_wave<CNT, QN>(sc, rp, p, active, n_active, exhausted ? 0u : adv_den, leaf_den, lc, ws, tick, boost_mask);
        if (active && trace_done(p.ts)) {
            sth(&q.hits[slot], wf_hit_pack(p.ts));
            slot = NONE;
        }
    }
    flush_counters<CNT>(cnt, lane, 0u, lc, ws);
}

// One lane per live path of the step; the waves of sub-queue k take its blocks of 64 states in turn and append to sub-queue k of the next step.
// PREC: precise shading (wf_surface_f64: the bounce's geometry in f64, the main ray's residuals carried in the state).
// LOG (hr_debug_path_log): the per-path event log of pt_core.h rides along in the state and is written out when the path ends — eight words
// per path as trace_kernel's LOG instantiation writes them; as there, the shortcut "GGX below the horizon" is off (the log records the
// visibility verdict also where the BSDF is zero).
template <bool CNT, bool PREC, bool LOG>
__global__ __launch_bounds__(256) void wf_shade_kernel(Scene sc, RenderParams rp, float *recs, WfQueues q, uint32_t step, Counters *cnt, uint32_t *plog) {
    const uint32_t lane = threadIdx.x & 63u, in = step & 1u, out = in ^ 1u;
    // the launch's last kernel (the stream runs them one after the other; this one is all but empty): its first thread stamps the trace side's end
    if (rp.gov && step == WF_STEPS && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&rp.gov->t1[1][rp.gov_slot], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), waves = gridDim.x * 4u;
    const uint32_t sq = w % WF_SUBQ, per = waves / WF_SUBQ;
    const uint32_t n = wf_paths(q.counts[(size_t)step * WF_SUBQ + sq]);
    WfCounts *cn_out = q.counts + (size_t)(step + 1u) * WF_SUBQ + sq;
    const uint32_t rq = sq * q.cap_rays, pq = sq * q.cap_paths;
    const f4 *ra = q.ray_a[in] + rq, *rb = q.ray_b[in] + rq;
    const WfHitRec *hits = q.hits + rq;
    const uint32_t cull = (LOG ? 5u : 7u) & ~rp.nee_cull_off;
    LaneCounters lc = {0, 0, 0, 0, 0, 0};
    uint32_t finished = 0;
    for (uint32_t base = (w / WF_SUBQ) * 64u; base < n; base += per * 64u) {
        const uint32_t i = base + lane;
        const bool live = i < n;
        const uint32_t ii = pq + (live ? i : base);
        WfPath p;
        PathLog lg;
        {
            const f4 a = ldq(&q.st_a[in][ii]), b = ldq(&q.st_b[in][ii]), c = ldq(&q.st_c[in][ii]);
            p.pid = float_as_uint(a.x); p.st = float_as_uint(a.y); p.raybase = float_as_uint(a.z); p.cur_refl = a.w;
            p.accum = v3(b.x, b.y, b.z); p.refl = v3(c.x, c.y, c.z);
            if (LOG) {
                const f4 f = ldq(&q.st_f[in][ii]);
                lg.ev = (unsigned long long)float_as_uint(f.x) | ((unsigned long long)float_as_uint(f.y) << 32);
                lg.ev9 = float_as_uint(f.z); lg.hash = float_as_uint(f.w); lg.rays = float_as_uint(b.w);
            }
        }
        // the shadow rays of the iteration before, in the reference's order (renderer.rs:274)
        const uint32_t ns = live ? wf_shadow_rays(p) : 0u;
        const uint32_t it_shadow = wf_has_main(p) ? wf_iter(p) - 1u : wf_iter(p);   // the iteration those rays belong to
        for (uint32_t k = 0; k < ns; k++) {
            const uint32_t s = p.raybase + k;
            const f4 a = ldq(&ra[s]), b = ldq(&rb[s]);
            const WfHitRec h = ldh(&hits[s]);
            wf_contribute<CNT>(sc, p, h, v3(a.x, a.y, a.z), a.w, v3(b.x, b.y, b.z), b.w, &lc);
            if (LOG) {
                lg.rays++;
                const float dt = h.t - a.w;
                if (h.pt != WF_MISS && dt * dt < OFFSET_F * 4.0f) plog_or(lg, it_shadow, 16u << (q.tag[in][rq + s] & 3u));
            }
        }
        bool fin = true, bounce = false;
        uint32_t ns_new = 0;
        WfBounce bc;
        WfBounceX bx;
        bc.nee = false;
        const float *rec = recs + wf_rec_base(p.pid);
        if (live && wf_has_main(p)) {
            p.refl = p.refl * p.cur_refl;      // renderer.rs:197, second factor (1 in step 1)
            const uint32_t s = p.raybase + ns;
            const f4 a = ldq(&ra[s]), b = ldq(&rb[s]);
            if (PREC) {
                const f4 fo = ldq(&q.st_d[in][ii]), fd = ldq(&q.st_e[in][ii]);
                fin = wf_surface_f64<CNT, LOG>(sc, p, rec, rp.rec_lo_off ? rec + rp.rec_lo_off : nullptr, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(fo.x, fo.y, fo.z), v3(fd.x, fd.y, fd.z), ldh(&hits[s]), bc, bx, &lc, &lg);
            } else
                fin = wf_surface<CNT, LOG>(sc, p, rec, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), ldh(&hits[s]), bc, &lc, &lg);
            if (!fin) {
                if (bc.nee)
                    for (uint32_t k = 0; k < sc.num_emitters; k++) {
                        V3f d; float len;
                        if (wf_nee_ray(sc, bc, k, cull, d, len)) ns_new++;
                        else { if (CNT) lc.shadow_culled++; if (LOG) lg.rays++; }
                    }
                bounce = wf_bounces(p, bc);
                fin = !ns_new && !bounce;
            }
        }
        if (live && fin) {
            *reinterpret_cast<f4 *>(recs + wf_rec_base(p.pid)) = f4{p.accum.x, p.accum.y, p.accum.z, 0.0f};   // quad 0 of the record: accumulate_kernel sums them
            finished++;
            if (LOG) {
                const uint32_t item = p.pid >> 6, tile = item / rp.num_k;
                if (item - tile * rp.num_k == 0u) {   // the launch's first sampling
                    uint32_t px, py, sub;
                    tile_lane_pixel(rp, tile, p.pid & 63u, px, py, sub);
                    uint32_t *o = plog + (((size_t)py * rp.width + px) * 4u + sub) * 8u;
                    o[0] = float_as_uint(p.accum.x); o[1] = float_as_uint(p.accum.y); o[2] = float_as_uint(p.accum.z); o[3] = lg.rays;
                    o[4] = (uint32_t)lg.ev; o[5] = (uint32_t)(lg.ev >> 32); o[6] = lg.ev9; o[7] = lg.hash;
                }
            }
        }
        const bool go = live && !fin;
        const unsigned long long gm = wave_ballot(go);
        if (!gm) continue;
        const uint32_t mine = go ? ns_new + (bounce ? 1u : 0u) : 0u;
        const uint32_t incl = wave_scan_inclusive(mine, lane);
        const uint32_t total = (uint32_t)__shfl((int)incl, 63);
        unsigned long long at = 0;
        if (lane == 0) at = atomicAdd(&cn_out->alloc, ((unsigned long long)__popcll(gm) << 32) | total);
        const uint32_t rbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)at), pbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(at >> 32));
        if (go) {
            const uint32_t first = rbase + incl - mine;
            uint32_t s = rq + first;
            if (bc.nee)
                for (uint32_t k = 0; k < sc.num_emitters; k++) {
                    V3f d; float len;
                    if (wf_nee_ray(sc, bc, k, cull, d, len)) {
                        stq(&q.ray_a[out][s], f4{bc.next_o.x, bc.next_o.y, bc.next_o.z, len});
                        stq(&q.ray_b[out][s], f4{d.x, d.y, d.z, wf_nee_weight(sc, bc, k, d, len)});
                        if (LOG) q.tag[out][s] = k;
                        s++;
                    }
                }
            if (bounce) {
                stq(&q.ray_a[out][s], f4{bc.next_o.x, bc.next_o.y, bc.next_o.z, WF_MAIN_RAY});
                stq(&q.ray_b[out][s], f4{bc.next_d.x, bc.next_d.y, bc.next_d.z, 0.0f});
            }
            const uint32_t pos = pq + pbase + lane_rank(gm);
            stq(&q.st_a[out][pos], f4{uint_as_float(p.pid), uint_as_float(wf_st(wf_iter(p) + (bounce ? 1u : 0u), bounce, wf_a2(p), ns_new)), uint_as_float(first), bc.cur_refl});
            stq(&q.st_b[out][pos], f4{p.accum.x, p.accum.y, p.accum.z, LOG ? uint_as_float(lg.rays) : 0.0f});
            stq(&q.st_c[out][pos], f4{p.refl.x, p.refl.y, p.refl.z, 0.0f});
            if (PREC) {
                stq(&q.st_d[out][pos], f4{bx.next_o_lo.x, bx.next_o_lo.y, bx.next_o_lo.z, 0.0f});
                stq(&q.st_e[out][pos], f4{bx.next_d_lo.x, bx.next_d_lo.y, bx.next_d_lo.z, 0.0f});
            }
            if (LOG) stq(&q.st_f[out][pos], f4{uint_as_float((uint32_t)lg.ev), uint_as_float((uint32_t)(lg.ev >> 32)), uint_as_float(lg.ev9), uint_as_float(lg.hash)});
        }
    }
    if (CNT) {
        unsigned long long v[3] = {finished, lc.rays, lc.shadow_culled};
        for (int i = 0; i < 3; i++) {
            unsigned long long x = v[i];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            v[i] = x;
        }
        if (lane == 0) { atomicAdd(&cnt->paths, v[0]); atomicAdd(&cnt->rays, v[1]); atomicAdd(&cnt->shadow_culled, v[2]); }
    }
}
