// Per-lane code of the SPLIT pipeline (trace_mode 1): the loop body of PathTracingRenderer::calc_pixel (renderer.rs:174-200)
// cut at scene.intersect — a traversal kernel that only walks (wf_kernels.h wf_traverse_kernel: ray in, closest hit out, <= 64 VGPRs) and a
// shading kernel that only shades (wf_shade_kernel), with the path parked in memory between them.  One STEP = one iteration of renderer.rs:174
// for every live path of the launch: the traversal kernel walks the iteration's main ray AND the NEE shadow rays of the iteration before
// (renderer.rs:274-296) in one pass, the shading kernel first adds those shadow rays' contributions in the reference's order, then shades the
// main hit and emits the next rays.  Same arithmetic as path_advance (pt_core.h) — the functions below are its pieces, in its order, on the
// same fp32 values: a launch's accumulator is the same bit for bit whichever pipeline rendered it (test_kernel_variants_render_the_same_bits;
// CPU tier: tests/emu drives these functions against path_advance).
// __host__ __device__ like pt_core.h: the host emulation compiles the very same functions.
#pragma once
#include "pt_core.h"

namespace hr {

// ---- records in HBM (all 16-byte quads, structure-of-arrays: a wave's accesses are contiguous kilobytes)
// ray[slot]   = {o.xyz, len} {d.xyz, w}     len = WF_MAIN_RAY for a main ray, else the shadow ray's length |sample - origin| (nee_setup);
//                                            w = the NEE weight bsdf * G / pdf of renderer.rs:283-292, computed when the ray is emitted
// hit[slot]   = {t, prim | type << 28 (all ones: miss), u, v}            what the walk leaves in TraceState
// state[pos]  = {path id, st, first ray slot, cur_refl} {accum.xyz, -} {refl.xyz, -}   one per LIVE path, position = rank in the step's queue
//               st: bits 0-3 iteration of the main ray in flight, bit 4 a main ray is in flight, bits 8-11 2a (accepted lens attempt, Path::q),
//               bits 12-31 shadow rays in flight (slots raybase .. raybase + n - 1, the main ray behind them)
//               refl: the reflectance BEFORE the sampled bounce's scalar (what the shadow rays' contributions are multiplied by, renderer.rs:295);
//               cur_refl is multiplied in when the step's contributions are in (renderer.rs:197)
static const float WF_MAIN_RAY = -1.0f;
struct alignas(16) WfHitRec { float t; uint32_t pt; float u, v; };
static const uint32_t WF_MISS = 0xffffffffu;

HD WfHitRec wf_hit_pack(const TraceState &ts) {
    WfHitRec h;
    h.t = ts.t; h.pt = ts.prim < 0 ? WF_MISS : ((uint32_t)ts.prim | ((uint32_t)ts.type << 28)); h.u = ts.u; h.v = ts.v;
    return h;
}
HD void wf_hit_unpack(const WfHitRec &h, TraceState &ts) {
    ts.t = h.t; ts.u = h.u; ts.v = h.v;
    ts.prim = h.pt == WF_MISS ? -1 : (int32_t)(h.pt & (MAX_PRIMS_PER_TYPE - 1u));
    ts.type = h.pt == WF_MISS ? 0 : (int32_t)((h.pt >> 28) & 3u);
    ts.cur = NODE_END; ts.leaf = 0; ts.leaf2 = 0;
}

// what the traversal kernel keeps per lane: the ray and the walk, nothing of the path
struct TravLane {
    Ray ray;
    TraceState ts;
    float shadow_len;     // WF_MAIN_RAY: closest-hit query; >= 0: a shadow ray of that length (search limit + early out as in nee_setup / shadow_early_out)
};
// pt_core.h shadow_early_out for a TravLane (a main ray's shadow_len is negative: never true)
HD void shadow_early_out(TravLane &p) {
    if (p.ts.t < p.shadow_len - 0.0201f) { p.ts.cur = NODE_END; p.ts.leaf = 0; p.ts.leaf2 = 0; }
}
HD void wf_lane_begin(const Scene &sc, TravLane &p, V3f o, V3f d, float len) {
    ray_set(p.ray, o, d);
    ray_quantise(sc, p.ray);
    p.shadow_len = len;
    trace_begin(p.ts, len >= 0.0f ? len + 0.03f : T_INF, p.ray.start);   // nee_setup's search limit for a shadow ray
}

struct WfPath { uint32_t pid, st, raybase; float cur_refl; V3f accum, refl; };
HD uint32_t wf_iter(const WfPath &p) { return p.st & 15u; }
HD bool wf_has_main(const WfPath &p) { return (p.st & 16u) != 0u; }
HD uint32_t wf_a2(const WfPath &p) { return (p.st >> 8) & 15u; }
HD uint32_t wf_shadow_rays(const WfPath &p) { return p.st >> 12; }
HD uint32_t wf_st(uint32_t iter, bool has_main, uint32_t a2, uint32_t shadow_rays) { return iter | (has_main ? 16u : 0u) | (a2 << 8) | (shadow_rays << 12); }
// path id = item * 64 + lane of the tile, item = tile * num_k + sampling of the launch: the index of the path's hand-off record
HD size_t wf_rec_base(uint32_t pid) { return (size_t)(pid >> 6) * REC_ITEM_FLOATS + (size_t)(pid & 63u) * 4u; }

// One NEE shadow ray's contribution (renderer.rs:280-292), from what the walk found and what was known when the ray was emitted.
template <bool CNT>
HD void wf_contribute(const Scene &sc, WfPath &p, const WfHitRec &h, V3f o, float len, V3f d, float w, LaneCounters *cn) {
    if (CNT) cn->rays++;
    TraceState ts;
    wf_hit_unpack(h, ts);
    const float dt = ts.t - len;
    if (ts.prim >= 0 && dt * dt < OFFSET_F * 4.0f) {
        const Material mt = sc.materials[hit_element(sc, ts)];
        V3f e = v3(mt.emission);
        if (mt.emission_img >= 0) {
            Ray r;
            r.o = o; r.d = d;
            Surf s;
            hit_surface(sc, r, ts, true, s);
            e = tex_sample(sc, mt.emission_img, e, s.u, s.v);
        }
        p.accum = p.accum + p.refl * (e * w);
    }
}

// the shaded point, as far as the rays that leave it need it
struct WfBounce { V3f next_o, next_d, n, view; float cur_refl, param, roughness, r0, r1; int32_t surface; bool nee; };

// The main ray's result (renderer.rs:175-196 = path_advance's main-ray branch up to the NEE loop): returns true when the path ends here.
// rec = the path's hand-off record (recs + wf_rec_base(pid)): the iteration's two draws, and for a primary ray on a sphere the f64 residuals.
// LOG: the per-path event log of pt_core.h (PathLog), exactly as path_advance<.., LOG> keeps it.
template <bool CNT, bool LOG = false>
HD bool wf_surface(const Scene &sc, WfPath &p, const float *rec, V3f ro, V3f rd, const WfHitRec &h, WfBounce &b, LaneCounters *cn, PathLog *lg = nullptr) {
    if (CNT) cn->rays++;
    if (LOG) lg->rays++;
    TraceState ts;
    wf_hit_unpack(h, ts);
    const uint32_t a2 = wf_a2(p), it = wf_iter(p);
    const f2v r01 = *reinterpret_cast<const f2v *>(rec + rec_slot(0u, a2 + 2u * it));   // renderer.rs:175
    b.r0 = r01[0]; b.r1 = r01[1];
    if (ts.prim < 0) {  // scene.rs:398 + renderer.rs:196,199
        if (LOG) { plog_or(*lg, it, 1u); plog_sky(sc, *lg, rd); }
        p.accum = p.accum + p.refl * sky_sample(sc, rd);
        return true;
    }
    Surf s;
    RayFix fix = no_ray_fix();
    if (ts.type == 1 && it == 1u) ray_fix_load(rec, 0u, a2, fix.o, fix.d);
    Ray r;
    r.o = ro; r.d = rd;
    hit_surface(sc, r, ts, material_needs_uv(sc, hit_element(sc, ts)), s, fix);
    PointMat m;
    material_at(sc, s.elem, s.u, s.v, m);
    b.view = -rd;
    bool transmitted;
    const bool sampled = bsdf_sample(m, b.r0, b.r1, s.pos, b.view, s.n, b.next_o, b.next_d, b.cur_refl, transmitted);
    if (LOG) {
        if (ts.type == 2) plog_hit(*lg, 0x1000 + cuboid_face_of(s.n));
        const Material mt = sc.materials[s.elem];
        plog_quad(sc, *lg, mt.albedo_img, s.u, s.v); plog_quad(sc, *lg, mt.emission_img, s.u, s.v); plog_quad(sc, *lg, mt.roughness_img, s.u, s.v);
        plog_hit(*lg, s.elem);
        if (ts.type == 1) lg->ev9 += 256u;
        if (ts.type == 0) plog_hit(*lg, (int32_t)(sc.tri_face[ts.prim] + 0x9e3779b9u));
        plog_or(*lg, it, sampled ? (2u + (uint32_t)m.surface) | (transmitted ? 8u : 0u) : 7u);
    }
    if (!sampled) return true;  // renderer.rs:190-193
    p.accum = p.accum + p.refl * m.emission;          // renderer.rs:196
    p.refl = p.refl * m.albedo;                       // renderer.rs:183,295 and the first factor of :197
    b.nee = nee_available(m.surface) && sc.num_emitters > 0;
    b.n = s.n; b.param = m.param; b.roughness = m.roughness; b.surface = m.surface;
    return false;
}

// The shadow ray towards emitter k (scene.rs:92-101 + renderer.rs:276-279) = nee_setup, shortcuts included: false = known to add nothing.
HD bool wf_nee_ray(const Scene &sc, const WfBounce &b, uint32_t k, uint32_t cull, V3f &d, float &len) {
    const Emitter em = sc.emitters[k];
    float unit_z = 1.0f - 2.0f * b.r1;
    float a = HR_SQRT(fmaxf(1.0f - unit_z * unit_z, 0.0f));
    float sn_, cs_;
    HR_SINCOS_2PI(b.r0, sn_, cs_);
    V3f sn = v3(a * cs_, a * sn_, unit_z);
    const float ro = em.r + OFFSET_F;
    V3f sp = v3(em.c) + ro * sn;
    V3f sv = sp - b.next_o;
    float sl2 = dot(sv, sv), isl = HR_RSQ(sl2);
    len = sl2 * isl;
    d = sv * isl;
    if (cull) {
        const float slack = 0.0221f + 1e-6f * len;
        const float x = ro * dot(sn, d);
        const bool far_side = (cull & 1u) && x > slack && x * x > 2.0f * (2.0f * em.r * OFFSET_F + OFFSET_F * OFFSET_F) && len > 2.0f * x;
        const float nd = dot(b.n, d);
        const bool ggx_below = (cull & 2u) && b.surface == 3 && signbit(nd);
        if (far_side || ggx_below) return false;
    }
    return true;
}
// emission * THIS is what a visible sample of emitter k adds per unit of reflectance (renderer.rs:283-292): bsdf * G / pdf
HD float wf_nee_weight(const Scene &sc, const WfBounce &b, uint32_t k, V3f d, float len) {
    const Emitter em = sc.emitters[k];
    V3f sp = b.next_o + d * len;
    V3f sn = (sp - v3(em.c)) * HR_RCP(em.r + OFFSET_F);
    float dot_0 = fabsf(dot(b.n, d)), dot_l = fabsf(dot(sn, d));
    float g = (dot_0 * dot_l) * HR_RCP(len * len);
    float inv_pdf = 4.0f * PI_F * em.r * em.r;
    return bsdf_eval(b.surface, b.param, b.roughness, b.view, b.n, d) * g * inv_pdf;
}
// does the path go on with the sampled bounce?  (renderer.rs:197-199 on the reflectance the next step will form)
HD bool wf_bounces(const WfPath &p, const WfBounce &b) { return !(is_zero(p.refl * b.cur_refl) || wf_iter(p) >= 9u); }


// ---------------------------------------------------------------------------------------------
// precise shading in the split pipeline: prec_core.h's shade_hit_f64 (shared with the megakernel's path_advance<.., PREC>) + the queue records
struct WfBounceX { V3f next_o_lo, next_d_lo; };   // the residuals of WfBounce::next_o / next_d (precise shading only)

// wf_surface with the geometry in f64.  (ro, rd) + (fo, fd) = the ray the path follows; h = what the fp32 walk found; rec_lo = the record's twin
// with the draws' residuals (nullptr: none).
// LOG: the per-path event log of pt_core.h (PathLog), as path_advance<.., LOG> keeps it.
template <bool CNT, bool LOG>
HD bool wf_surface_f64(const Scene &sc, WfPath &p, const float *rec, const float *rec_lo, V3f ro, V3f rd, V3f fo, V3f fd, const WfHitRec &h, WfBounce &b, WfBounceX &bx, LaneCounters *cn, PathLog *lg) {
    if (CNT) cn->rays++;
    if (LOG) lg->rays++;
    TraceState ts;
    wf_hit_unpack(h, ts);
    const uint32_t a2 = wf_a2(p), it = wf_iter(p);
    const f2v r01 = *reinterpret_cast<const f2v *>(rec + rec_slot(0u, a2 + 2u * it));   // renderer.rs:175
    b.r0 = r01[0]; b.r1 = r01[1];
    if (ts.prim < 0) {
        if (LOG) { plog_or(*lg, it, 1u); plog_sky(sc, *lg, rd); }
        p.accum = p.accum + p.refl * sky_sample(sc, rd);
        return true;
    }
    PrecHit x;
    double r0 = (double)b.r0, r1 = (double)b.r1;
    if (rec_lo) {   // the draws' residuals (device_scene.h: the records' twin)
        const f2v l01 = *reinterpret_cast<const f2v *>(rec_lo + rec_slot(0u, a2 + 2u * it));
        r0 += (double)l01[0]; r1 += (double)l01[1];
    }
    shade_hit_f64(sc, ro, rd, fo, fd, ts, r0, r1, x);
    b.view = -rd;
    b.cur_refl = x.cur_refl;
    const PointMat &m = x.m;
    const Surf &s = x.s;
    const V3f nf = x.nf;
    const bool sampled = x.sampled, transmitted = x.transmitted;
    const D3 no = x.no, nd = x.nd;
    if (LOG) {
        if (ts.type == 2) plog_hit(*lg, 0x1000 + cuboid_face_of(nf));
        const Material mt = sc.materials[s.elem];
        plog_quad(sc, *lg, mt.albedo_img, s.u, s.v); plog_quad(sc, *lg, mt.emission_img, s.u, s.v); plog_quad(sc, *lg, mt.roughness_img, s.u, s.v);
        plog_hit(*lg, s.elem);
        if (ts.type == 1) lg->ev9 += 256u;
        if (ts.type == 0) plog_hit(*lg, (int32_t)(sc.tri_face[ts.prim] + 0x9e3779b9u));
        plog_or(*lg, it, sampled ? (2u + (uint32_t)m.surface) | (transmitted ? 8u : 0u) : 7u);
    }
    if (!sampled) return true;
    p.accum = p.accum + p.refl * m.emission;
    p.refl = p.refl * m.albedo;
    b.nee = nee_available(m.surface) && sc.num_emitters > 0;
    b.n = nf; b.param = m.param; b.roughness = m.roughness; b.surface = m.surface;
    b.next_o = narrow(no); bx.next_o_lo = residual(no, b.next_o);
    b.next_d = narrow(nd); bx.next_d_lo = residual(nd, b.next_d);
    return false;
}

}  // namespace hr
