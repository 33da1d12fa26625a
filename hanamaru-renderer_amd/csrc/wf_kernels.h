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
// per step: number of rays, number of live paths, the traversal kernel's queue head (zeroed per launch)
struct WfCounts { uint32_t rays, paths, head, pad; };

struct WfQueues {
    f4 *ray_a[2], *ray_b[2];            // [step & 1][slot]: {o, len}, {d, w}
    WfHitRec *hits;                     // [slot] of the step being worked on
    f4 *st_a[2], *st_b[2], *st_c[2];    // [step & 1][position]: live-path state
    WfCounts *counts;                   // [WF_STEPS + 2]
};

// inclusive prefix sum over the wave's lanes
__device__ __forceinline__ uint32_t wave_scan_inclusive(uint32_t v, uint32_t lane) {
    for (uint32_t off = 1; off < 64u; off <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)v, off);
        if (lane >= off) v += u;
    }
    return v;
}

// one wave per item (tile x sampling of the launch = 64 paths), lane j = (pixel of the tile, sub-sample)
__global__ __launch_bounds__(256) void wf_start_kernel(Scene sc, RenderParams rp, float *recs, WfQueues q) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * 4u, items = rp.tiles_x * rp.tiles_y * rp.num_k;
    for (uint32_t item = blockIdx.x * 4u + (threadIdx.x >> 6); item < items; item += waves) {
        const uint32_t tile = item / rp.num_k, k = item - tile * rp.num_k;
        uint32_t px, py, sub;
        tile_lane_pixel(rp, tile, lane, px, py, sub);
        const bool valid = px < rp.width && py < rp.height;
        Path p;
        p.q = (k << 6) | lane;
        p.tile = tile;
        if (valid) path_start(sc, rp, p, px, py, sub, recs + (size_t)tile * rp.num_k * REC_ITEM_FLOATS);
        const unsigned long long m = wave_ballot(valid);
        if (!m) continue;
        uint32_t rb = 0, pb = 0;
        if (lane == 0) { rb = atomicAdd(&q.counts[1].rays, (uint32_t)__popcll(m)); pb = atomicAdd(&q.counts[1].paths, (uint32_t)__popcll(m)); }
        rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)rb);
        pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)pb);
        if (valid) {
            const uint32_t r = lane_rank(m), slot = rb + r, pos = pb + r;
            q.ray_a[1][slot] = f4{p.ray.o.x, p.ray.o.y, p.ray.o.z, WF_MAIN_RAY};
            q.ray_b[1][slot] = f4{p.ray.d.x, p.ray.d.y, p.ray.d.z, 0.0f};
            q.st_a[1][pos] = f4{uint_as_float(item * 64u + lane), uint_as_float(wf_st(1u, true, (p.q >> 12) & 15u, 0u)), uint_as_float(slot), 1.0f};
            q.st_b[1][pos] = f4{0.0f, 0.0f, 0.0f, 0.0f};
            q.st_c[1][pos] = f4{1.0f, 1.0f, 1.0f, 0.0f};
        }
    }
}

// Persistent waves over the step's ray queue: a lane holds one ray and its walk; lanes whose walk is done write their hit and take the next
// ray of the queue (ballot + prefix rank, one atomic per refill).  The walk is traverse_wave — the megakernel's phase C, unchanged.
template <bool CNT, bool QN>
__global__ __launch_bounds__(256, 8) void wf_traverse_kernel(Scene sc, RenderParams rp, WfQueues q, uint32_t step, Counters *cnt) {
    const uint32_t n = q.counts[step].rays;
    if (!n) return;
    const uint32_t lane = threadIdx.x & 63u;
    const f4 *ra = q.ray_a[step & 1u], *rb = q.ray_b[step & 1u];
    uint32_t *head = &q.counts[step].head;
    LaneCounters lc = {0, 0, 0, 0, 0, 0};
    WaveStats ws = {{0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0}, 0};
    const uint32_t adv_den = rp.adv_den ? rp.adv_den : 2u, leaf_den = rp.leaf_den ? rp.leaf_den : 2u;
    uint32_t tick = 0;
    bool exhausted = false;
    const uint32_t NONE = 0xffffffffu;
    uint32_t slot = NONE;
    TravLane p;
    p.ts.cur = NODE_END; p.ts.leaf = 0; p.ts.leaf2 = 0;
    for (;;) {
        const unsigned long long idle = wave_ballot(slot == NONE);
        if (idle && !exhausted) {
            const uint32_t want = (uint32_t)__popcll(idle);
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(head, want);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            if (t + want >= n) exhausted = true;
            const uint32_t e = t + lane_rank(idle);
            if (slot == NONE && e < n) {
                const f4 a = ra[e], b = rb[e];
                slot = e;
                wf_lane_begin(sc, p, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), a.w);
            }
        }
        const bool active = slot != NONE;
        const uint32_t n_active = (uint32_t)__popcll(wave_ballot(active));
        if (!n_active) {
            if (exhausted) break;
            continue;
        }
        traverse_wave<CNT, QN>(sc, rp, p, active, n_active, exhausted ? 0u : adv_den, leaf_den, lc, ws, tick, 0u);
        if (active && trace_done(p.ts)) {
            q.hits[slot] = wf_hit_pack(p.ts);
            slot = NONE;
        }
    }
    flush_counters<CNT>(cnt, lane, 0u, lc, ws);
}

// One lane per live path of the step.
template <bool CNT>
__global__ __launch_bounds__(256) void wf_shade_kernel(Scene sc, RenderParams rp, float *recs, WfQueues q, uint32_t step, Counters *cnt) {
    const uint32_t n = q.counts[step].paths;
    if (!n) return;
    const uint32_t lane = threadIdx.x & 63u, in = step & 1u, out = in ^ 1u;
    const uint32_t waves = gridDim.x * 4u;
    const uint32_t cull = 7u & ~rp.nee_cull_off;
    LaneCounters lc = {0, 0, 0, 0, 0, 0};
    uint32_t finished = 0;
    for (uint32_t base = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u; base < n; base += waves * 64u) {
        const uint32_t i = base + lane;
        const bool live = i < n;
        WfPath p;
        {
            const uint32_t ii = live ? i : base;
            const f4 a = q.st_a[in][ii], b = q.st_b[in][ii], c = q.st_c[in][ii];
            p.pid = float_as_uint(a.x); p.st = float_as_uint(a.y); p.raybase = float_as_uint(a.z); p.cur_refl = a.w;
            p.accum = v3(b.x, b.y, b.z); p.refl = v3(c.x, c.y, c.z);
        }
        // the shadow rays of the iteration before, in the reference's order (renderer.rs:274)
        const uint32_t ns = live ? wf_shadow_rays(p) : 0u;
        for (uint32_t k = 0; k < ns; k++) {
            const uint32_t s = p.raybase + k;
            const f4 a = q.ray_a[in][s], b = q.ray_b[in][s];
            wf_contribute<CNT>(sc, p, q.hits[s], v3(a.x, a.y, a.z), a.w, v3(b.x, b.y, b.z), b.w, &lc);
        }
        bool fin = true, bounce = false;
        uint32_t ns_new = 0;
        WfBounce bc;
        bc.nee = false;
        const float *rec = recs + wf_rec_base(p.pid);
        if (live && wf_has_main(p)) {
            p.refl = p.refl * p.cur_refl;      // renderer.rs:197, second factor (1 in step 1)
            const uint32_t s = p.raybase + ns;
            const f4 a = q.ray_a[in][s], b = q.ray_b[in][s];
            fin = wf_surface<CNT>(sc, p, rec, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), q.hits[s], bc, &lc);
            if (!fin) {
                if (bc.nee)
                    for (uint32_t k = 0; k < sc.num_emitters; k++) {
                        V3f d; float len;
                        if (wf_nee_ray(sc, bc, k, cull, d, len)) ns_new++;
                        else if (CNT) lc.shadow_culled++;
                    }
                bounce = wf_bounces(p, bc);
                fin = !ns_new && !bounce;
            }
        }
        if (live && fin) {
            *reinterpret_cast<f4 *>(recs + wf_rec_base(p.pid)) = f4{p.accum.x, p.accum.y, p.accum.z, 0.0f};   // quad 0 of the record: accumulate_kernel sums them
            finished++;
        }
        const bool go = live && !fin;
        const unsigned long long gm = wave_ballot(go);
        if (!gm) continue;
        const uint32_t mine = go ? ns_new + (bounce ? 1u : 0u) : 0u;
        const uint32_t incl = wave_scan_inclusive(mine, lane);
        const uint32_t total = (uint32_t)__shfl((int)incl, 63);
        uint32_t rbase = 0, pbase = 0;
        if (lane == 0) { rbase = atomicAdd(&q.counts[step + 1u].rays, total); pbase = atomicAdd(&q.counts[step + 1u].paths, (uint32_t)__popcll(gm)); }
        rbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)rbase);
        pbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)pbase);
        if (go) {
            const uint32_t first = rbase + incl - mine;
            uint32_t s = first;
            if (bc.nee)
                for (uint32_t k = 0; k < sc.num_emitters; k++) {
                    V3f d; float len;
                    if (wf_nee_ray(sc, bc, k, cull, d, len)) {
                        q.ray_a[out][s] = f4{bc.next_o.x, bc.next_o.y, bc.next_o.z, len};
                        q.ray_b[out][s] = f4{d.x, d.y, d.z, wf_nee_weight(sc, bc, k, d, len)};
                        s++;
                    }
                }
            if (bounce) {
                q.ray_a[out][s] = f4{bc.next_o.x, bc.next_o.y, bc.next_o.z, WF_MAIN_RAY};
                q.ray_b[out][s] = f4{bc.next_d.x, bc.next_d.y, bc.next_d.z, 0.0f};
            }
            const uint32_t pos = pbase + lane_rank(gm);
            q.st_a[out][pos] = f4{uint_as_float(p.pid), uint_as_float(wf_st(wf_iter(p) + (bounce ? 1u : 0u), bounce, wf_a2(p), ns_new)), uint_as_float(first), bc.cur_refl};
            q.st_b[out][pos] = f4{p.accum.x, p.accum.y, p.accum.z, 0.0f};
            q.st_c[out][pos] = f4{p.refl.x, p.refl.y, p.refl.z, 0.0f};
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
