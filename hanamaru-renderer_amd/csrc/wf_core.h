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
// PRECISE shading (option shading_precision 1): the geometry of a bounce in the reference's own precision.  The walk stays fp32 — it only
// has to find the right primitive —, but the ray the path really follows is carried as fp32 + residual (o + o_lo, d + d_lo: what rounding
// the f64 ray to the fp32 ray the traversal walks took away), and everything between "this primitive was hit" and "the next ray" is f64:
// the hit distance again from the f64 ray and the f64 primitive (triangle plane from the f64 vertices: Scene::tri_exact; sphere centre and
// radius: sphere_lo; cuboid bounds: cuboid_lo), hit point, normal, mirror / Snell / Fresnel (material.rs:154-199, vector.rs:60-71).  A
// faceted glass body is a billiard and a small sphere multiplies a position error by 2 t / r per bounce: fp32 ray state leaves 100 - 1,000 ppm
// of the paths that took the reference's branches off by more than 1e-3 in such scenes; the megakernel has no registers
// for this (profiles/NOTES.md G), the shading kernel of the split pipeline does.  What stays fp32: the draws themselves (the hand-off
// record holds them rounded once), the directions SAMPLED from them (diffuse lobe, GGX half vector), textures, radiometry.
struct D3 { double x, y, z; };
HD D3 dv(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
HD D3 operator+(D3 a, D3 b) { return dv(a.x + b.x, a.y + b.y, a.z + b.z); }
HD D3 operator-(D3 a, D3 b) { return dv(a.x - b.x, a.y - b.y, a.z - b.z); }
HD D3 operator*(D3 a, double s) { return dv(a.x * s, a.y * s, a.z * s); }
HD D3 operator-(D3 a) { return dv(-a.x, -a.y, -a.z); }
HD double ddot(D3 a, D3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
HD D3 widen(V3f hi, V3f lo) { return dv((double)hi.x + (double)lo.x, (double)hi.y + (double)lo.y, (double)hi.z + (double)lo.z); }
HD V3f narrow(D3 a) { return v3((float)a.x, (float)a.y, (float)a.z); }
HD V3f residual(D3 a, V3f hi) { return v3((float)(a.x - (double)hi.x), (float)(a.y - (double)hi.y), (float)(a.z - (double)hi.z)); }
HD D3 dreflect(D3 v, D3 n) { return v - n * (2.0 * ddot(v, n)); }   // vector.rs:60-62

// material.rs:154-199 in f64.  `in` = direction of the arriving ray.  Returns the new ray and the reflectance scalar; transmitted as in pt_core.h.
HD void sample_refraction_f64(double r0, D3 pos, D3 in, D3 n, double ior, D3 &no, D3 &nd, float &refl, bool &transmitted) {
    transmitted = false;
    const bool incoming = signbit(ddot(in, n));
    const D3 on = incoming ? n : -n;
    const double nnt = incoming ? 1.0 / ior : ior;
    const D3 rdir = dreflect(in, on);
    const double vn = ddot(in, on);
    const double k = 1.0 - nnt * nnt * (1.0 - vn * vn);      // vector.rs:64-71
    if (k < 0.0) { no = pos + on * (double)OFFSET_F; nd = rdir; refl = 1.0f; return; }
    const D3 tdir = in * nnt - on * (nnt * vn + sqrt(k));
    if (tdir.x == 0.0 && tdir.y == 0.0 && tdir.z == 0.0) { no = pos + on * (double)OFFSET_F; nd = rdir; refl = 1.0f; return; }
    const double cos_i = -ddot(in, on), cos_t = -ddot(tdir, on);
    const double a = nnt * cos_i - cos_t, b = nnt * cos_i + cos_t, c = nnt * cos_t - cos_i, d = nnt * cos_t + cos_i;
    const double fr = 0.5 * (a * a / (b * b) + c * c / (d * d));
    if (r0 <= fr) { no = pos + on * (double)OFFSET_F; nd = rdir; refl = 1.0f; }
    else { no = pos - on * (double)OFFSET_F; nd = tdir; refl = (float)(nnt * nnt); transmitted = true; }
}

// sin / cos of 2 pi r for r in [0, 1), to f64 accuracy: quadrant + Taylor polynomials on [-pi/4, pi/4] (no f64 transcendental hardware;
// the fp32 v_sin_f32 / v_cos_f32 the megakernel uses are off by up to ~1e-6 — more than the rounding of the draw itself).  The same code
// on the host: the emulation and the kernels agree.
HD void sincos_2pi_f64(double r, double &sn, double &cs) {
    const double q = floor(r * 4.0 + 0.5);
    const double x = (r - q * 0.25) * 6.28318530717958647692, x2 = x * x;
    const double sx = x * fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, -1.0 / 1307674368000.0, 1.0 / 6227020800.0), -1.0 / 39916800.0), 1.0 / 362880.0), -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
    const double cx = fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, 1.0 / 20922789888000.0, -1.0 / 87178291200.0), 1.0 / 479001600.0), -1.0 / 3628800.0), 1.0 / 40320.0), -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
    const int k = (int)q & 3;
    sn = k == 0 ? sx : k == 1 ? cx : k == 2 ? -sx : -cx;
    cs = k == 0 ? cx : k == 1 ? -sx : k == 2 ? -cx : sx;
}
HD D3 dcross(D3 a, D3 b) { return dv(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HD void tangent_basis_f64(D3 n, D3 &t, D3 &b) {  // material.rs:202-211
    const D3 up = fabs(n.x) > 1e-4 ? dv(0, 1, 0) : dv(1, 0, 0);
    const D3 c = dcross(up, n);
    t = c * (1.0 / sqrt(ddot(c, c)));
    b = dcross(n, t);
}
HD D3 sample_diffuse_f64(double r0, double r1, D3 n) {  // material.rs:227-248
    D3 t, b;
    tangent_basis_f64(n, t, b);
    double sn, cs;
    sincos_2pi_f64(r0, sn, cs);
    return (t * cs + b * sn) * sqrt(r1) + n * sqrt(1.0 - r1);
}
HD D3 sample_ggx_half_f64(double r0, double r1, D3 n, double alpha2) {  // material.rs:260-269
    D3 t, b;
    tangent_basis_f64(n, t, b);
    double sn, cs;
    sincos_2pi_f64(r0, sn, cs);
    const double cos_theta = sqrt((1.0 - r1) / (1.0 + (alpha2 - 1.0) * r1));
    const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    return t * (sin_theta * cs) + b * (sin_theta * sn) + n * cos_theta;
}

struct WfBounceX { V3f next_o_lo, next_d_lo; };   // the residuals of WfBounce::next_o / next_d (precise shading only)

// wf_surface with the geometry in f64.  (ro, rd) + (fo, fd) = the ray the path follows; h = what the fp32 walk found.
// LOG: the per-path event log of pt_core.h (PathLog), as path_advance<.., LOG> keeps it.
template <bool CNT, bool LOG>
HD bool wf_surface_f64(const Scene &sc, WfPath &p, const float *rec, V3f ro, V3f rd, V3f fo, V3f fd, const WfHitRec &h, WfBounce &b, WfBounceX &bx, LaneCounters *cn, PathLog *lg) {
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
    const D3 o = widen(ro, fo), d = widen(rd, fd);
    D3 pos, n;
    Surf s;
    s.u = ts.u; s.v = ts.v;
    const bool want_uv = material_needs_uv(sc, hit_element(sc, ts));
    if (ts.type == 0) {            // bvh.rs:266-290: the plane of the f64 triangle
        s.elem = sc.tri_shade[ts.prim].element;
        const TriX tx = sc.tri_exact[sc.tri_face[ts.prim]];
        n = dv(tx.n[0], tx.n[1], tx.n[2]);
        const D3 v0 = dv(tx.v0[0], tx.v0[1], tx.v0[2]);
        const double t = -ddot(n, o - v0) / ddot(n, d);
        pos = o + d * t;
    } else if (ts.type == 1) {     // scene.rs:58-66, the root again from the f64 ray and the f64 sphere
        const f4 sp = sc.spheres[ts.prim], lo = sc.sphere_lo[ts.prim];
        s.elem = sc.sphere_elem[ts.prim];
        const D3 c = dv((double)sp.x + (double)lo.x, (double)sp.y + (double)lo.y, (double)sp.z + (double)lo.z);
        const double cr = (double)sp.w + (double)lo.w;
        const D3 a = o - c;
        const double idd = 1.0 / ddot(d, d);     // (a sampled direction is a unit vector to fp32 only)
        const double bq = ddot(a, d) * idd;
        const D3 q = a - d * bq;
        const double disc = (cr * cr - ddot(q, q)) * idd;
        const double t = -bq - sqrt(disc > 0.0 ? disc : 0.0);
        const D3 nn = a + d * t;
        n = nn * (1.0 / sqrt(ddot(nn, nn)));
        pos = c + nn;
        if (want_uv) {  // scene.rs:67-71
            const V3f nf = narrow(n);
            s.v = 1.0f - acosf(fminf(fmaxf(nf.y, -1.0f), 1.0f)) * (1.0f / PI_F);
            float sg = signbit(nf.z) ? -1.0f : 1.0f;
            s.u = 0.5f - sg * acosf(fminf(fmaxf(nf.x * HR_RSQ(nf.x * nf.x + nf.z * nf.z), -1.0f), 1.0f)) * (1.0f / PI2_F);
        }
    } else {                       // bvh.rs:20-39 + scene.rs:152-182 on the f64 box
        const f4 mnf = sc.cuboids[2 * ts.prim], mxf = sc.cuboids[2 * ts.prim + 1];
        s.elem = float_as_int(mnf.w);
        const f4 mnl = sc.cuboid_lo[2 * s.elem], mxl = sc.cuboid_lo[2 * s.elem + 1];
        const D3 mn = dv((double)mnf.x + (double)mnl.x, (double)mnf.y + (double)mnl.y, (double)mnf.z + (double)mnl.z);
        const D3 mx = dv((double)mxf.x + (double)mxl.x, (double)mxf.y + (double)mxl.y, (double)mxf.z + (double)mxl.z);
        const double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z;
        const double t1 = (mn.x - o.x) * ix, t2 = (mx.x - o.x) * ix, t3 = (mn.y - o.y) * iy, t4 = (mx.y - o.y) * iy, t5 = (mn.z - o.z) * iz, t6 = (mx.z - o.z) * iz;
        const double tmin = fmax(fmax(fmin(t1, t2), fmin(t3, t4)), fmin(t5, t6)), tmax = fmin(fmin(fmax(t1, t2), fmax(t3, t4)), fmax(t5, t6));
        // the fp32 walk decided that the box is hit; should the f64 slabs disagree at a grazing edge, the walk's distance stands
        const double dist = (tmin <= tmax && !signbit(tmax)) ? (signbit(tmin) ? tmax : tmin) : (double)ts.t;
        pos = o + d * dist;
        const D3 uvw = dv((pos.x - mn.x) / (mx.x - mn.x), (pos.y - mn.y) / (mx.y - mn.y), (pos.z - mn.z) / (mx.z - mn.z));
        const double E = 1e-4;   // config.rs:7
        int face;
        if (fabs(pos.y - mx.y) < E) face = 0; else if (fabs(pos.y - mn.y) < E) face = 1; else if (fabs(pos.x - mn.x) < E) face = 2;
        else if (fabs(pos.x - mx.x) < E) face = 3; else if (fabs(pos.z - mn.z) < E) face = 4; else if (fabs(pos.z - mx.z) < E) face = 5;
        else {   // (cannot happen for a hit found in f64; the nearest face, as hit_surface)
            const double dy1 = fabs(pos.y - mx.y), dy0 = fabs(pos.y - mn.y), dx0 = fabs(pos.x - mn.x), dx1 = fabs(pos.x - mx.x), dz0 = fabs(pos.z - mn.z), dz1 = fabs(pos.z - mx.z);
            const double best = fmin(fmin(fmin(dy1, dy0), fmin(dx0, dx1)), fmin(dz0, dz1));
            face = best == dy1 ? 0 : best == dy0 ? 1 : best == dx0 ? 2 : best == dx1 ? 3 : best == dz0 ? 4 : 5;
        }
        if (face == 0) { n = dv(0, 1, 0); s.u = (float)uvw.x; s.v = (float)(1.0 - uvw.z); }
        else if (face == 1) { n = dv(0, -1, 0); s.u = (float)uvw.x; s.v = (float)(1.0 - uvw.z); }
        else if (face == 2) { n = dv(-1, 0, 0); s.u = (float)uvw.z; s.v = (float)uvw.y; }
        else if (face == 3) { n = dv(1, 0, 0); s.u = (float)uvw.z; s.v = (float)uvw.y; }
        else if (face == 4) { n = dv(0, 0, -1); s.u = (float)uvw.x; s.v = (float)uvw.y; }
        else { n = dv(0, 0, 1); s.u = (float)uvw.x; s.v = (float)uvw.y; }
    }
    const V3f nf = narrow(n);
    PointMat m;
    material_at(sc, s.elem, s.u, s.v, m);
    b.view = -rd;
    bool transmitted = false, sampled = true;
    D3 no, nd;
    switch (m.surface) {       // material.rs:91-151
        case 0: no = pos + n * (double)OFFSET_F; nd = sample_diffuse_f64((double)b.r0, (double)b.r1, n); b.cur_refl = 1.0f; break;
        case 1: no = pos + n * (double)OFFSET_F; nd = dreflect(d, n); b.cur_refl = 1.0f; break;
        case 2: sample_refraction_f64((double)b.r0, pos, d, n, (double)m.param, no, nd, b.cur_refl, transmitted); break;
        case 3: {
            const float alpha2 = m.roughness * m.roughness;
            const D3 hh = sample_ggx_half_f64((double)b.r0, (double)b.r1, n, (double)m.roughness * (double)m.roughness);
            nd = dreflect(d, hh);
            const double lnd = ddot(nd, n);
            if (signbit(lnd)) { sampled = false; break; }
            const float ln = (float)lnd, vn = (float)-ddot(d, n), vh = (float)-ddot(d, hh), hn = (float)ddot(hh, n);
            b.cur_refl = f_schlick(vh, m.param) * saturatef(g_smith_joint(ln, vn, alpha2) * vh * HR_RCP(hn * vn));
            no = pos + n * (double)OFFSET_F;
            break;
        }
        default: {
            const D3 hh = sample_ggx_half_f64((double)b.r0, (double)b.r1, n, (double)m.roughness * (double)m.roughness);
            sample_refraction_f64((double)b.r0, pos, d, hh, (double)m.param, no, nd, b.cur_refl, transmitted);
            break;
        }
    }
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
