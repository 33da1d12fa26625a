// fp32 per-lane path-tracing core: everything below PathTracingRenderer::calc_pixel (renderer.rs:163-203).
// Written as __host__ __device__ inline functions over the POD device scene so the identical code runs
// in the HIP kernels (hr_api.hip) and in the host emulation used by the CPU-only tests (tests/emu).
//
// Structure: a path is a small state machine (`Path`) driven by three calls
//     path_start()   — camera ray (camera.rs:83-96), iteration 1
//     trace_step()   — ONE node visit of the stackless threaded BVH traversal (replaces bvh.rs:213-263)
//     path_advance() — when the current ray is finished: shade / NEE / next ray (renderer.rs:174-200)
// so that a wavefront can keep all 64 lanes in the traversal loop and refill finished lanes.
#pragma once
#include <math.h>
#if defined(HR_PATH_VERBOSE)
#include <stdio.h>
#include <stdlib.h>
#endif

#include "device_scene.h"
#include "isaac_core.h"

namespace hr {

// Transcendentals.  On the device the hardware forms are used: v_exp_f32(y * v_log_f32(x)) for x^y and
// v_sin_f32 / v_cos_f32, which take their argument in REVOLUTIONS — exactly the 2*pi*r0 the samplers need,
// so no range reduction is involved.  The host emulation uses libm.
// Reciprocal / square root use the raw 1-ulp hardware instructions (the compiler's default expansions add
// 4-5 range-scaling instructions each for denormal inputs, which do not occur here).
#if defined(__HIP_DEVICE_COMPILE__)
#define HR_POWF(x, y) __builtin_amdgcn_exp2f((y) * __builtin_amdgcn_logf(x))
#define HR_SINCOS_2PI(r, s, c) do { (s) = __builtin_amdgcn_sinf(r); (c) = __builtin_amdgcn_cosf(r); } while (0)
#define HR_RCP(x) __builtin_amdgcn_rcpf(x)
#define HR_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define HR_RSQ(x) __builtin_amdgcn_rsqf(x)
#else
#define HR_POWF(x, y) powf((x), (y))
#define HR_SINCOS_2PI(r, s, c) do { float ph_ = 6.28318530717958647692f * (r); (s) = sinf(ph_); (c) = cosf(ph_); } while (0)
#define HR_RCP(x) (1.0f / (x))
#define HR_SQRT(x) sqrtf(x)
#define HR_RSQ(x) (1.0f / sqrtf(x))
#endif

struct V3f { float x, y, z; };
HD V3f v3(float x, float y, float z) { V3f r; r.x = x; r.y = y; r.z = z; return r; }
HD V3f v3(const float *p) { return v3(p[0], p[1], p[2]); }
HD V3f operator+(V3f a, V3f b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
HD V3f operator-(V3f a, V3f b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
HD V3f operator*(V3f a, V3f b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
HD V3f operator*(V3f a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
HD V3f operator*(float s, V3f a) { return v3(a.x * s, a.y * s, a.z * s); }
HD V3f operator-(V3f a) { return v3(-a.x, -a.y, -a.z); }
// explicit FMAs: the contraction is then the same in every inlined copy, so a primitive test returns the same bits
// wherever the primitive sits in a leaf (closest hits do not depend on the tree)
HD float dot(V3f a, V3f b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
HD V3f cross(V3f a, V3f b) { return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))); }
HD V3f normalize(V3f a) { return a * HR_RSQ(dot(a, a)); }
HD V3f reflect(V3f v, V3f n) { return v - (2.0f * dot(v, n)) * n; }  // vector.rs:60-62
HD bool is_zero(V3f a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
HD float saturatef(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

static const float PI_F = 3.14159265358979323846f;
static const float PI2_F = 6.28318530717958647692f;
static const float EPS_F = 1e-4f;      // config.rs:7-8
static const float OFFSET_F = 1e-4f;
static const float T_INF = 3.0e38f;    // config.rs:9 INF = 1e100 (f64); fp32 stand-in

typedef float f2v __attribute__((vector_size(8)));   // one packed-fp32 operand (v_pk_add_f32 / v_pk_mul_f32 on gfx950)
struct Ray {
    V3f o, d, inv;
    uint32_t oct;     // direction octant (bit k set = component k negative)
    uint32_t start;   // where a walk of this ray begins: byte offset of its octant's copy in qnodes[] (walks on the 16-byte records), else 0
    // for the quantised nodes (device_scene.h QNode): distance to grid plane q = q * qinv + qc  (qinv = qstep * inv, qc = (qmin - o) * inv)
    V3f qinv, qc;
};
HD void ray_set(Ray &r, V3f o, V3f d) {
    r.o = o; r.d = d;
    r.inv = v3(HR_RCP(d.x), HR_RCP(d.y), HR_RCP(d.z));  // bvh.rs:21-25 (±inf for zero components)
    // by SIGN BIT (so that -0.0, whose reciprocal is -inf, counts as negative): the near / far planes of the per-octant
    // nodes must agree with the sign of r.inv
    r.oct = (signbit(d.x) ? 1u : 0u) | (signbit(d.y) ? 2u : 0u) | (signbit(d.z) ? 4u : 0u);
}

// A direction component of exactly zero has inv = +-inf, and q * inf + (-inf) is NaN for EVERY plane of that axis: the axis would
// drop out of the box test altogether (still conservative, but such a ray would walk most of the tree).  Clamped to +-1e30 the
// FMA keeps the sign of (plane - origin) down to differences of ~2e-6, far below the half step the planes are padded by.
HD float clamp_inv(float v) { return fminf(fmaxf(v, -1e30f), 1e30f); }
HD void ray_quantise(const Scene &sc, Ray &r) {
    r.start = sc.qnodes ? qnode_offset((int)r.oct, sc.num_nodes + 1u, 0u) : 0u;
    const V3f inv = v3(clamp_inv(r.inv.x), clamp_inv(r.inv.y), clamp_inv(r.inv.z));
    r.qinv = v3(sc.qstep[0] * inv.x, sc.qstep[1] * inv.y, sc.qstep[2] * inv.z);
    r.qc = v3((sc.qmin[0] - r.o.x) * inv.x, (sc.qmin[1] - r.o.y) * inv.y, (sc.qmin[2] - r.o.z) * inv.z);
}

struct TraceState {
    uint32_t cur;     // node to visit next, NODE_END = finished
    float t;          // closest distance so far
    int32_t prim;     // leaf-ordered primitive index of the closest hit, -1 = none
    int32_t type;     // 0 tri, 1 sphere, 2 cuboid
    float u, v;       // barycentrics (triangles)
    uint32_t leaf;    // pending leaf word (node visited, primitives not yet tested), 0 = none
    uint32_t leaf2;   // a second pending leaf, found while `leaf` was still waiting for the leaf phase (trace kernel only)
};
// start: 0 for walks on the 32-byte records (node index), Ray::start for walks on the 16-byte records (byte offset)
HD void trace_begin(TraceState &ts, float tmax, uint32_t start = 0u) { ts.cur = start; ts.t = tmax; ts.prim = -1; ts.type = 0; ts.u = ts.v = 0.0f; ts.leaf = 0; ts.leaf2 = 0; }
HD bool trace_done(const TraceState &ts) { return ts.cur == NODE_END && ts.leaf == 0; }

struct LaneCounters { uint32_t rays, node_tests, tri_tests, sphere_tests, cuboid_tests, shadow_culled; };

// bvh.rs:266-290 (two-sided, accepts t == best, rejects det == 0) on the derived record TriT (device_scene.h): the plane first, then the
// barycentrics as two dot products with the hit position relative to v0.  Straight-line: the lanes of a wave test different triangles,
// an early return saves nothing unless ALL of them take it, and the branches cost scalar slots the seed kernel's waves want.
//   `live` = this lane really has a triangle to test (the second slot of an odd-sized leaf repeats the first).
// Acceptance: t >= 0, u >= 0, v >= 0 (one min3 and one compare), u + v <= 1, t <= closest so far.  u <= 1 (bvh.rs:277) follows from
// v >= 0 and u + v <= 1 (rounding is monotonic).  nu . d == 0 needs no compare of its own: 1 / 0 = inf makes t infinite or NaN, the
// position p then has infinite or NaN components exactly where d is non-zero, and a dot product with it is +-inf or NaN (0 x inf),
// never finite: u + v <= 1 fails (a NaN fails every compare; min ignores a NaN operand, the sum does not).
template <bool CNT>
HD void tri_test(const TriT &tr, const Ray &r, TraceState &ts, int32_t index, bool live, LaneCounters *cn) {
    if (CNT) cn->tri_tests += live ? 1u : 0u;
    const V3f n = v3(tr.n), a = v3(tr.ax, tr.ay, tr.az), b = v3(tr.bx, tr.by, tr.bz);
    const V3f dd = r.o - v3(tr.v0);
    const float t = -dot(n, dd) * HR_RCP(dot(n, r.d));
    const V3f p = v3(fmaf(t, r.d.x, dd.x), fmaf(t, r.d.y, dd.y), fmaf(t, r.d.z, dd.z));
    const float u = dot(a, p), v = dot(b, p);   // (packed FMAs for the two pairs of dot products: measured, 2 % slower — the operand pairs cost moves)
    const bool ok = live & (fminf(fminf(t, u), v) >= 0.0f) & (u + v <= 1.0f) & (t <= ts.t);
    ts.t = ok ? t : ts.t; ts.prim = ok ? index : ts.prim; ts.type = ok ? 0 : ts.type; ts.u = ok ? u : ts.u; ts.v = ok ? v : ts.v;
}
// scene.rs:58-78 (outer root only), in f64 on the fp32 ray and sphere.  Same roots as the reference's b^2 - c form, with the
// discriminant taken from the perpendicular offset of the centre (r^2 - |a - b d|^2: no cancellation when the origin is many radii
// away).  Two things make this the one primitive test that is not fp32:
//   * the fp32 direction is a unit vector only to ~2e-7 (v_rsq_f32, v_sin / v_cos), and unlike the triangle and slab tests — exact
//     for any direction length — the quadratic takes |d| = 1 for granted: the hit distance would be off by (|d|^2 - 1) t, ten times the
//     rounding of the position itself, with a bias that showed in the image MEAN of sphere scenes (4e-5).  The roots are therefore
//     those of |a + t d|^2 = r^2 with dd = |d|^2 divided out (1 / dd = 2 - dd to 1e-14);
//   * the hit / miss decision at a silhouette and the distance of a grazing hit (t = -b - sqrt(disc), disc -> 0) lose most of their
//     fp32 digits; in f64 the only error left is the fp32 ray's own.
// Measured on the sphere-only scene (256x144x2 against the f64 oracle): channels within 1e-3 0.9907 -> 0.9958, image mean
// 3.437471 -> 3.437606 (oracle 3.437607).  sqrt: v_rsq_f32 seed + one Newton step in f64 (1e-14), no f64 division or square root.
HD double hr_sqrt_f64(double x) {
    const double y0 = (double)HR_RSQ(fmaxf((float)x, 1e-30f));
    const double y1 = y0 * fma(-0.5 * x, y0 * y0, 1.5);
    return x * y1;
}
// returns the hit distance, or a negative value for a miss.  NOT inlined on the device: its ~20 live f64 values would otherwise
// raise the register pressure of the whole leaf phase (measured: 8 -> 27 spilled VGPRs, trace kernel +7.8 %)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline))
#else
inline
#endif
float sphere_root(float sx, float sy, float sz, float sw, float ox, float oy, float oz, float ddx, float ddy, float ddz) {
    const double ax = (double)ox - (double)sx, ay = (double)oy - (double)sy, az = (double)oz - (double)sz;
    const double dx = ddx, dy = ddy, dz = ddz;
    const double idd = 2.0 - fma(dx, dx, fma(dy, dy, dz * dz));
    const double b = fma(ax, dx, fma(ay, dy, az * dz)) * idd;
    const double px = fma(-b, dx, ax), py = fma(-b, dy, ay), pz = fma(-b, dz, az);
    const double d = fma((double)sw, (double)sw, -fma(px, px, fma(py, py, pz * pz))) * idd;
    if (!(d > 0.0)) return -1.0f;
    return (float)(-b - hr_sqrt_f64(d));
}
template <bool CNT>
HD void sphere_test(const f4 &s, const Ray &r, TraceState &ts, int32_t index, LaneCounters *cn) {
    if (CNT) cn->sphere_tests++;
    const float t = sphere_root(s.x, s.y, s.z, s.w, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z);
    if (t > 0.0f && t < ts.t) { ts.t = t; ts.prim = index; ts.type = 1; }
}
// bvh.rs:20-39 + scene.rs:152-158 (hit part): slab test of the cuboid itself; the reference's `distance` is tmin if
// sign-positive else tmax.  Scalars only — a small private array here gets promoted to LDS by the compiler, and a trace kernel
// that owns even a few KiB of LDS cannot share a CU with the seed kernel, which owns all of it.  The reciprocals (bvh.rs:21-25)
// are formed here so that the walk on the quantised nodes need not keep them.
template <bool CNT>
HD void cuboid_test(const f4 &mn, const f4 &mx, const Ray &r, TraceState &ts, int32_t index, LaneCounters *cn) {
    if (CNT) cn->cuboid_tests++;
    const float ix = HR_RCP(r.d.x), iy = HR_RCP(r.d.y), iz = HR_RCP(r.d.z);
    float t1 = (mn.x - r.o.x) * ix, t2 = (mx.x - r.o.x) * ix;
    float t3 = (mn.y - r.o.y) * iy, t4 = (mx.y - r.o.y) * iy;
    float t5 = (mn.z - r.o.z) * iz, t6 = (mx.z - r.o.z) * iz;
    float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    if (!(tmin <= tmax && !signbit(tmax))) return;
    float dist = signbit(tmin) ? tmax : tmin;
    if (dist < ts.t) { ts.t = dist; ts.prim = index; ts.type = 2; }
}

// The traversal is split in two so that a wavefront can run them as separate, well-filled phases
// ("while-while"): trace_node() is ONE box test of the stackless threaded walk and parks a hit leaf in
// ts.leaf; trace_leaf() tests that leaf's primitives.  Nodes whose entry distance exceeds the closest hit
// so far are skipped (the reference visits them, bvh.rs:214,240 — the closest hit is the same).
// SPEC (trace kernel): a lane whose first leaf is still parked keeps walking until it finds a second one ("postponed
// leaf": fuller box AND leaf phases); the box it tests meanwhile is culled against a closest hit that does not yet include
// the parked leaf's primitives, which can only add node visits, never lose a hit.  Leaves are still tested in walk order.
// slab test of bvh.rs:20-39 with the planes already sorted along the ray (near / far per octant, device_scene.h): entry
// distance = max of the near terms, exit distance = min of the far terms.  A NaN term (origin exactly on a plane the ray
// runs parallel to) is ignored by max3 / min3, which keeps the test conservative.  Packed pairs: three subtracts, three
// multiplies.
HD bool node_hit(const Node &nd, const Ray &r, float tbest) {
    const f2v oxy = {r.o.x, r.o.y}, ozz = {r.o.z, r.o.z}, ixy = {r.inv.x, r.inv.y}, izz = {r.inv.z, r.inv.z};
    const f2v nxy = {nd.nearx, nd.neary}, fxy = {nd.farx, nd.fary}, zz = {nd.nearz, nd.farz};
    const f2v tn = (nxy - oxy) * ixy, tf = (fxy - oxy) * ixy, tz = (zz - ozz) * izz;
    const float entry = fmaxf(fmaxf(fmaxf(tn[0], tn[1]), tz[0]), 0.0f);      // as in trace_qnode: one compare
    const float exit_ = fminf(fminf(fminf(tf[0], tf[1]), tz[1]), tbest);
    return entry <= exit_;
}
// What a visit does to the walk.  `link` is the record's link word (inner: miss successor, leaf: leaf word), `next` the successor
// in preorder (inner: the near child, leaf: the node behind it).  Written for the instruction count of the box phase's loop: one
// compare for leaf-ness, one select for the walk, two for the parked leaves.
// SPEC (trace kernel): a lane walks on with ONE leaf parked and stops at the second.  The two slots are a shift register — the
// newest leaf in ts.leaf, the one before it in ts.leaf2 — so that parking is two selects on one mask; the leaf phase tests the OLDER
// one first (trace_leaf_next): leaves are tested in walk order.
template <bool SPEC>
HD void node_advance(TraceState &ts, bool hit, uint32_t link, uint32_t next) {
    const bool leaf = node_word_is_leaf(link);
    ts.cur = (leaf || hit) ? next : link;
    const bool found = leaf && hit;
    if (SPEC) {
        ts.leaf2 = found ? ts.leaf : ts.leaf2;
        ts.leaf = found ? link : ts.leaf;
    } else {
        ts.leaf = found ? link : 0u;
    }
}
// SPEC walk: may this lane take another node?  (not finished, and at most one leaf parked)
HD bool trace_can_walk(const TraceState &ts) { return ts.leaf2 == 0u && ts.cur != NODE_END; }
template <bool CNT, bool SPEC = false>
HD void trace_node(const Scene &sc, const Ray &r, TraceState &ts, LaneCounters *cn) {
    const Node nd = sc.nodes[(size_t)r.oct * sc.num_nodes + ts.cur];
    if (CNT) cn->node_tests++;
    // 32-byte records carry both successors: a = hit (inner: near child; leaf: leaf word), b = miss / leaf done
    const bool leaf = node_word_is_leaf(nd.a);
    node_advance<SPEC>(ts, node_hit(nd, r, ts.t), leaf ? nd.a : nd.b, leaf ? nd.b : nd.a);
}
// One visit on the 16-byte nodes: a single 16-byte load; the six planes are grid coordinates, exact in fp32, and the box test is
// six FMAs on the ray's precomputed (qinv, qc).  The grid planes lie at least one step outside the true box and the fp32 error
// of q * qinv + qc is ~0.004 steps, so the test can only over-report hits (more visits), never lose one.  Same visits in the same order as
// trace_node() on the 32-byte records of the same tree, plus the few extra ones the fatter boxes let through.
template <bool CNT, bool SPEC = false>
HD void trace_qnode(const Scene &sc, const Ray &r, TraceState &ts, LaneCounters *cn) {
    const QNode nd = *reinterpret_cast<const QNode *>(reinterpret_cast<const char *>(sc.qnodes) + ts.cur);   // ts.cur: byte offset, octant copy included
    if (CNT) cn->node_tests++;
    const f2v nxy = {(float)(nd.xy_near & 0xffffu), (float)(nd.xy_near >> 16)}, fxy = {(float)(nd.xy_far & 0xffffu), (float)(nd.xy_far >> 16)};
    const f2v zz = {(float)(nd.z_nf & 0xffffu), (float)(nd.z_nf >> 16)};
    const f2v ixy = {r.qinv.x, r.qinv.y}, izz = {r.qinv.z, r.qinv.z}, cxy = {r.qc.x, r.qc.y}, czz = {r.qc.z, r.qc.z};
    const f2v tn = nxy * ixy + cxy, tf = fxy * ixy + cxy, tz = zz * izz + czz;
    // entry distance clamped at 0, exit distance clamped at the closest hit so far: ONE compare decides (three compares joined by two
    // scalar ANDs before).  A box that ends exactly at the origin (exit = -0.0) now counts as hit — one more visit, never a lost one.
    const float entry = fmaxf(fmaxf(fmaxf(tn[0], tn[1]), tz[0]), 0.0f);
    const float exit_ = fminf(fminf(fminf(tf[0], tf[1]), tz[1]), ts.t);
    const bool hit = entry <= exit_;
    node_advance<SPEC>(ts, hit, nd.link, ts.cur + 16u);
}
template <bool CNT>
HD void trace_leaf(const Scene &sc, const Ray &r, TraceState &ts, LaneCounters *cn) {
    const uint32_t type = leaf_type(ts.leaf), count = leaf_count(ts.leaf), first = leaf_first(ts.leaf);
    ts.leaf = 0;
    if (type == 0) {
        // two triangles per round so their loads are in flight together
        for (uint32_t k = 0; k < count; k += 2) {
            const TriT ta = sc.tris[first + k];
            const bool two = k + 1 < count;
            const TriT tb = sc.tris[first + (two ? k + 1 : k)];
            tri_test<CNT>(ta, r, ts, (int32_t)(first + k), true, cn);
            tri_test<CNT>(tb, r, ts, (int32_t)(first + k + 1), two, cn);
        }
    } else if (type == 1) {
        for (uint32_t k = 0; k < count; k++) sphere_test<CNT>(sc.spheres[first + k], r, ts, (int32_t)(first + k), cn);
    } else {
        for (uint32_t k = 0; k < count; k++)
            cuboid_test<CNT>(sc.cuboids[2 * (first + k)], sc.cuboids[2 * (first + k) + 1], r, ts, (int32_t)(first + k), cn);
    }
}
// SPEC walk: test the OLDER of the parked leaves (ts.leaf2 when two are parked, else ts.leaf) and free its slot
template <bool CNT>
HD void trace_leaf_next(const Scene &sc, const Ray &r, TraceState &ts, LaneCounters *cn) {
    const bool two = ts.leaf2 != 0u;
    const uint32_t newest = ts.leaf;
    ts.leaf = two ? ts.leaf2 : ts.leaf;
    trace_leaf<CNT>(sc, r, ts, cn);          // tests ts.leaf, clears it
    ts.leaf = two ? newest : 0u;
    ts.leaf2 = 0u;
}
// scalar convenience (host emulation, debug kernel): one visit = node + its leaf
template <bool CNT>
HD void trace_step(const Scene &sc, const Ray &r, TraceState &ts, LaneCounters *cn) {
    trace_node<CNT>(sc, r, ts, cn);
    if (ts.leaf) trace_leaf<CNT>(sc, r, ts, cn);
}

// ---------------------------------------------------------------------------------------------
// surface attributes of the closest hit
struct Surf { V3f pos, n; float u, v; int32_t elem; };

HD int32_t float_as_int(float f) { union { float f; int32_t i; } c; c.f = f; return c.i; }

// Sphere hits in the reference's precision.  hr_rsqrt_f64: v_rsq_f32 seed + one Newton step in f64 (1e-14), as hr_sqrt_f64.
HD double hr_rsqrt_f64(double x) {
    const double y0 = (double)HR_RSQ(fmaxf((float)x, 1e-30f));
    return y0 * fma(-0.5 * x, y0 * y0, 1.5);
}
// Hit point and normal of a sphere hit, in f64 (scene.rs:58-66): the root is found again from the f64 ray — the closest-hit search's fp32
// distance (sphere_test) only chose the sphere — so that neither the rounding of t (|t| 6e-8 along the ray) nor, for a primary ray, the
// fp32 rounding of the camera ray (3e-8 x distance) reaches the normal, where a small sphere multiplies a position error by 1 / radius
// and every further bounce off a small sphere multiplies it again (r = 0.1 at distance 5: x 100 per bounce — measured on the sphere
// scenes by the per-path accounting of round 4: 1,800 ppm of the paths that took the reference's branches were off by more than 1e-3
// before, 540 ppm after; what is left is the fp32 rounding of the sampled directions and of the draws themselves).
struct SphHit { float px, py, pz, nx, ny, nz; };
// (cx, cy, cz, cr): the sphere in f64 — its fp32 record plus what rounding it took away (Scene::sphere_lo).  The reference holds centres in
// f64; an fp32 centre is off by 3e-8 |c|, and the normal of an r = 0.1 sphere by ten times that — measured (round 4, per-path accounting):
// it was the LARGEST term left in the sphere scenes' same-branch error (563 -> 95 ppm of the paths beyond 1e-3 in `spheres`).
HD SphHit sphere_surface_f64(double cx, double cy, double cz, double cr, double ox, double oy, double oz, double dx, double dy, double dz) {
    const double ax = ox - cx, ay = oy - cy, az = oz - cz;
    const double idd = 2.0 - fma(dx, dx, fma(dy, dy, dz * dz));
    const double b = fma(ax, dx, fma(ay, dy, az * dz)) * idd;
    const double qx = fma(-b, dx, ax), qy = fma(-b, dy, ay), qz = fma(-b, dz, az);
    const double disc = fma(cr, cr, -fma(qx, qx, fma(qy, qy, qz * qz))) * idd;
    const double t = -b - hr_sqrt_f64(disc > 0.0 ? disc : 0.0);     // (a grazing hit the f64 ray just misses: the tangent point)
    const double nx = fma(t, dx, ax), ny = fma(t, dy, ay), nz = fma(t, dz, az);
    const double il = hr_rsqrt_f64(fma(nx, nx, fma(ny, ny, nz * nz)));
    SphHit h;
    h.px = (float)(cx + nx); h.py = (float)(cy + ny); h.pz = (float)(cz + nz);
    h.nx = (float)(nx * il); h.ny = (float)(ny * il); h.nz = (float)(nz * il);
    return h;
}
// The ray is the fp32 ray plus a correction (do, dd): zero for any ray but a path's first — there it is what the fp32 rounding of the f64
// camera ray took away (path_start computes the camera ray in f64 and parks the two residuals in dead slots of the path's record); the
// sphere is its fp32 record plus (lx, ly, lz, lw) = Scene::sphere_lo.  Inlined (unlike sphere_root, which sits in the leaf phase): as a
// called function its 20 arguments and the call's clobbers cost the shading code three spilled VGPRs and 1.5 % of the kernel (same-box A/B,
// round 4); inlined, the f64 block is one spill and the kernel 12.9 -> 12.75 ms alone.
HD SphHit sphere_surface(float sx, float sy, float sz, float sw, float lx, float ly, float lz, float lw, float ox, float oy, float oz, float dx, float dy, float dz,
                                  float dox, float doy, float doz, float ddx, float ddy, float ddz) {
    return sphere_surface_f64((double)sx + (double)lx, (double)sy + (double)ly, (double)sz + (double)lz, (double)sw + (double)lw, (double)ox + (double)dox,
                              (double)oy + (double)doy, (double)oz + (double)doz, (double)dx + (double)ddx, (double)dy + (double)ddy, (double)dz + (double)ddz);
}
// the residuals of a path's first ray (zero: any other ray)
struct RayFix { V3f o, d; };
HD RayFix no_ray_fix() { RayFix f; f.o = v3(0, 0, 0); f.d = v3(0, 0, 0); return f; }
// where path_start parks them: six slots of the path's record that no iteration reads — a path whose accepted lens attempt is a consumes draws
// 2a + 2 .. 2a + 19 (renderer.rs:175 for iterations 1 .. 9), the slots before are rejected lens attempts, the slots from 2a + 20 on spare
// (a <= LENS_FAST - 1 = 4, REC_DRAWS = 28: at least 10 dead slots; the path's radiance goes to slots 0 .. 3 when it ENDS).  Pairs of slots
// are 8 contiguous bytes (rec_slot), the first is even.
// The six slots are consecutive (modulo REC_DRAWS) from the even slot 2a + 20: a quad (16 contiguous bytes) and a pair (8), in the order the
// alignment of the first slot dictates — two stores / two loads per path, two 32-byte sectors of its record.
HD uint32_t ray_fix_slot(uint32_t twice_a, uint32_t k) { uint32_t s = twice_a + 20u + 2u * k; return s >= (uint32_t)REC_DRAWS ? s - (uint32_t)REC_DRAWS : s; }
HD void ray_fix_store(float *recs, uint32_t lb, uint32_t twice_a, V3f fo, V3f fd) {
    if ((twice_a & 2u) == 0u) {   // 2a + 20 is a multiple of 4: quad, then pair
        *reinterpret_cast<f4 *>(recs + rec_slot(lb, ray_fix_slot(twice_a, 0u))) = f4{fo.x, fo.y, fo.z, fd.x};
        *reinterpret_cast<f2v *>(recs + rec_slot(lb, ray_fix_slot(twice_a, 2u))) = f2v{fd.y, fd.z};
    } else {                      // pair, then quad
        *reinterpret_cast<f2v *>(recs + rec_slot(lb, ray_fix_slot(twice_a, 0u))) = f2v{fo.x, fo.y};
        *reinterpret_cast<f4 *>(recs + rec_slot(lb, ray_fix_slot(twice_a, 1u))) = f4{fo.z, fd.x, fd.y, fd.z};
    }
}
HD void ray_fix_load(const float *recs, uint32_t lb, uint32_t twice_a, V3f &fo, V3f &fd) {
    if ((twice_a & 2u) == 0u) {
        const f4 q = *reinterpret_cast<const f4 *>(recs + rec_slot(lb, ray_fix_slot(twice_a, 0u)));
        const f2v r = *reinterpret_cast<const f2v *>(recs + rec_slot(lb, ray_fix_slot(twice_a, 2u)));
        fo = v3(q.x, q.y, q.z); fd = v3(q.w, r[0], r[1]);
    } else {
        const f2v r = *reinterpret_cast<const f2v *>(recs + rec_slot(lb, ray_fix_slot(twice_a, 0u)));
        const f4 q = *reinterpret_cast<const f4 *>(recs + rec_slot(lb, ray_fix_slot(twice_a, 1u)));
        fo = v3(r[0], r[1], q.x); fd = v3(q.y, q.z, q.w);
    }
}

HD void hit_surface(const Scene &sc, const Ray &r, const TraceState &ts, bool want_uv, Surf &s, const RayFix &fix) {
    s.pos = r.o + r.d * ts.t;
    s.u = ts.u; s.v = ts.v;
    if (ts.type == 0) {
        const TriS tr = sc.tri_shade[ts.prim];
        s.n = v3(tr.n);  // bvh.rs:286 (never flipped), normalised in f64 by tri_derive()
        s.elem = tr.element;
    } else if (ts.type == 1) {
        const f4 sp = sc.spheres[ts.prim];
        s.elem = sc.sphere_elem[ts.prim];
        {
            const f4 lo = sc.sphere_lo[ts.prim];
            const SphHit h = sphere_surface(sp.x, sp.y, sp.z, sp.w, lo.x, lo.y, lo.z, lo.w, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, fix.o.x, fix.o.y, fix.o.z, fix.d.x, fix.d.y, fix.d.z);
            s.pos = v3(h.px, h.py, h.pz);
            s.n = v3(h.nx, h.ny, h.nz);
        }
        if (want_uv) {  // scene.rs:67-71
            s.v = 1.0f - acosf(fminf(fmaxf(s.n.y, -1.0f), 1.0f)) * (1.0f / PI_F);
            float sg = signbit(s.n.z) ? -1.0f : 1.0f;
            s.u = 0.5f - sg * acosf(fminf(fmaxf(s.n.x * HR_RSQ(s.n.x * s.n.x + s.n.z * s.n.z), -1.0f), 1.0f)) * (1.0f / PI2_F);
        }
    } else {
        const f4 mn = sc.cuboids[2 * ts.prim], mx = sc.cuboids[2 * ts.prim + 1];
        s.elem = float_as_int(mn.w);
        V3f uvw = v3((s.pos.x - mn.x) * HR_RCP(mx.x - mn.x), (s.pos.y - mn.y) * HR_RCP(mx.y - mn.y), (s.pos.z - mn.z) * HR_RCP(mx.z - mn.z));
        // scene.rs:160-182 face cascade Y+, Y-, X-, X+, Z-, Z+
        s.n = v3(0.f, 0.f, 0.f);
        if (fabsf(s.pos.y - mx.y) < EPS_F) { s.n = v3(0, 1, 0); s.u = uvw.x; s.v = 1.0f - uvw.z; }
        else if (fabsf(s.pos.y - mn.y) < EPS_F) { s.n = v3(0, -1, 0); s.u = uvw.x; s.v = 1.0f - uvw.z; }
        else if (fabsf(s.pos.x - mn.x) < EPS_F) { s.n = v3(-1, 0, 0); s.u = uvw.z; s.v = uvw.y; }
        else if (fabsf(s.pos.x - mx.x) < EPS_F) { s.n = v3(1, 0, 0); s.u = uvw.z; s.v = uvw.y; }
        else if (fabsf(s.pos.z - mn.z) < EPS_F) { s.n = v3(0, 0, -1); s.u = uvw.x; s.v = uvw.y; }
        else if (fabsf(s.pos.z - mx.z) < EPS_F) { s.n = v3(0, 0, 1); s.u = uvw.x; s.v = uvw.y; }
        else {
            // fp32 only: far from the origin the hit position is off its face by more than EPS (|pos| * 6e-8 > 1e-4 from
            // |pos| ~ 1700; the f64 reference always finds a face) — take the nearest face instead of a zero normal
            float dy1 = fabsf(s.pos.y - mx.y), dy0 = fabsf(s.pos.y - mn.y), dx0 = fabsf(s.pos.x - mn.x), dx1 = fabsf(s.pos.x - mx.x);
            float dz0 = fabsf(s.pos.z - mn.z), dz1 = fabsf(s.pos.z - mx.z);
            float best = fminf(fminf(fminf(dy1, dy0), fminf(dx0, dx1)), fminf(dz0, dz1));
            if (best == dy1) { s.n = v3(0, 1, 0); s.u = uvw.x; s.v = 1.0f - uvw.z; }
            else if (best == dy0) { s.n = v3(0, -1, 0); s.u = uvw.x; s.v = 1.0f - uvw.z; }
            else if (best == dx0) { s.n = v3(-1, 0, 0); s.u = uvw.z; s.v = uvw.y; }
            else if (best == dx1) { s.n = v3(1, 0, 0); s.u = uvw.z; s.v = uvw.y; }
            else if (best == dz0) { s.n = v3(0, 0, -1); s.u = uvw.x; s.v = uvw.y; }
            else { s.n = v3(0, 0, 1); s.u = uvw.x; s.v = uvw.y; }
        }
    }
}

HD void hit_surface(const Scene &sc, const Ray &r, const TraceState &ts, bool want_uv, Surf &s) { hit_surface(sc, r, ts, want_uv, s, no_ray_fix()); }

// ---------------------------------------------------------------------------------------------
// textures — texture.rs:29-63, color.rs:18-36
HD uint32_t f32_as_u32_sat(float v) {  // Rust `as u32`
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967040.0f) return 4294967295u;
    return (uint32_t)v;
}
HD V3f texel(const Scene &sc, const ImageRef &im, uint32_t x, uint32_t y) {
    x = x > im.width - 1 ? im.width - 1 : x;
    uint32_t yy = im.height - y - 1u;  // wrapping
    yy = yy > im.height - 1 ? im.height - 1 : yy;
    uint32_t p = sc.texels[im.offset + yy * im.width + x];
    const float k = 1.0f / 255.0f;
    return v3((float)(p & 255u) * k, (float)((p >> 8) & 255u) * k, (float)((p >> 16) & 255u) * k);
}
HD float gamma_to_linear(float v) { return HR_POWF(v, 2.2f); }
HD V3f sample_bilinear(const Scene &sc, int32_t image, float u, float v) {
    const ImageRef im = sc.images[image];
    float x = u * (float)im.width, y = v * (float)im.height;
    float x1 = floorf(x), y1 = floorf(y);
    float x2 = x1 + 1.0f, y2 = y1 + 1.0f;
    uint32_t ix1 = f32_as_u32_sat(x1), ix2 = f32_as_u32_sat(x2), iy1 = f32_as_u32_sat(y1), iy2 = f32_as_u32_sat(y2);
    V3f p11 = texel(sc, im, ix1, iy1), p12 = texel(sc, im, ix1, iy2), p21 = texel(sc, im, ix2, iy1), p22 = texel(sc, im, ix2, iy2);
    V3f g = p11 * ((x2 - x) * (y2 - y)) + p21 * ((x - x1) * (y2 - y)) + p12 * ((x2 - x) * (y - y1)) + p22 * ((x - x1) * (y - y1));
    return v3(gamma_to_linear(g.x), gamma_to_linear(g.y), gamma_to_linear(g.z));
}
HD V3f tex_sample(const Scene &sc, int32_t image, V3f tint, float u, float v) {  // texture.rs:108-114
    if (image >= 0) return sample_bilinear(sc, image, u, v) * tint;
    return tint;
}
HD V3f sky_sample(const Scene &sc, V3f d) {  // scene.rs:295-319
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face; float u, v;
    if (ax > ay && ax > az) {
        float i = HR_RCP(d.x);
        if (!signbit(d.x)) { face = 0; u = -d.z * i; v = d.y * i; } else { face = 1; u = -d.z * i; v = -d.y * i; }
    } else if (ay > ax && ay > az) {
        float i = HR_RCP(d.y);
        if (!signbit(d.y)) { face = 2; u = d.x * i; v = -d.z * i; } else { face = 3; u = -d.x * i; v = -d.z * i; }
    } else {
        float i = HR_RCP(d.z);
        if (!signbit(d.z)) { face = 4; u = d.x * i; v = d.y * i; } else { face = 5; u = d.x * i; v = -d.y * i; }
    }
    u = 0.5f * (u + 1.0f); v = 0.5f * (v + 1.0f);
    if (sc.sky_quads) {
        // the footprint of sample_bilinear() as one 16-byte load (Scene::sky_quads).  x and y are clamped at 0: |u| <= 1 holds exactly in
        // the reference's f64, here the approximate reciprocal can leave it by an ulp, and a corner of -1 has another footprint than 0
        const float x = fmaxf(u * (float)sc.sky_w, 0.0f), y = fmaxf(v * (float)sc.sky_h, 0.0f);
        const float x1 = floorf(x), y1 = floorf(y), x2 = x1 + 1.0f, y2 = y1 + 1.0f;
        const uint32_t ix = f32_as_u32_sat(x1), iy = f32_as_u32_sat(y1);
        const uint32_t cx = ix > sc.sky_w ? sc.sky_w : ix, cy = iy > sc.sky_h ? sc.sky_h : iy;   // (beyond the last corner every texel is the clamped one, as at it)
        const uint32_t *q = sc.sky_quads + (((size_t)face * (sc.sky_h + 1u) + cy) * (sc.sky_w + 2u) + cx) * 2u;   // column pairs cx and cx + 1
        struct alignas(8) Quad { uint32_t p11, p12, p21, p22; };
        const Quad t = *reinterpret_cast<const Quad *>(q);
        const float k = 1.0f / 255.0f;
        const V3f p11 = v3((float)(t.p11 & 255u) * k, (float)((t.p11 >> 8) & 255u) * k, (float)((t.p11 >> 16) & 255u) * k);
        const V3f p12 = v3((float)(t.p12 & 255u) * k, (float)((t.p12 >> 8) & 255u) * k, (float)((t.p12 >> 16) & 255u) * k);
        const V3f p21 = v3((float)(t.p21 & 255u) * k, (float)((t.p21 >> 8) & 255u) * k, (float)((t.p21 >> 16) & 255u) * k);
        const V3f p22 = v3((float)(t.p22 & 255u) * k, (float)((t.p22 >> 8) & 255u) * k, (float)((t.p22 >> 16) & 255u) * k);
        const V3f g = p11 * ((x2 - x) * (y2 - y)) + p21 * ((x - x1) * (y2 - y)) + p12 * ((x2 - x) * (y - y1)) + p22 * ((x - x1) * (y - y1));
        return v3(sc.sky_intensity) * v3(gamma_to_linear(g.x), gamma_to_linear(g.y), gamma_to_linear(g.z));
    }
    return v3(sc.sky_intensity) * sample_bilinear(sc, sc.sky_image[face], u, v);
}

// ---------------------------------------------------------------------------------------------
// BSDFs — material.rs
struct PointMat { int32_t surface; float param; V3f albedo, emission; float roughness; };

HD void material_at(const Scene &sc, int32_t elem, float u, float v, PointMat &m) {  // scene.rs:389-396
    const Material mt = sc.materials[elem];
    m.surface = mt.surface; m.param = mt.param;
    m.albedo = tex_sample(sc, mt.albedo_img, v3(mt.albedo), u, v);
    m.emission = tex_sample(sc, mt.emission_img, v3(mt.emission), u, v);
    m.roughness = (mt.roughness_img >= 0) ? sample_bilinear(sc, mt.roughness_img, u, v).x * mt.roughness : mt.roughness;
}
HD bool material_needs_uv(const Scene &sc, int32_t elem) {
    const Material &mt = sc.materials[elem];
    return mt.albedo_img >= 0 || mt.emission_img >= 0 || mt.roughness_img >= 0;
}
HD bool nee_available(int32_t surface) { return surface == 0 || surface == 3; }  // material.rs:42-51

HD void tangent_basis(V3f n, V3f &t, V3f &b) {  // material.rs:202-211
    V3f up = fabsf(n.x) > EPS_F ? v3(0, 1, 0) : v3(1, 0, 0);
    t = normalize(cross(up, n));
    b = cross(n, t);
}
HD V3f sample_diffuse(float r0, float r1, V3f n) {  // material.rs:227-248
    V3f t, b;
    tangent_basis(n, t, b);
    float sn, cs;
    HR_SINCOS_2PI(r0, sn, cs);
    return (t * cs + b * sn) * HR_SQRT(r1) + n * HR_SQRT(fmaxf(1.0f - r1, 0.0f));
}
HD V3f sample_ggx_half(float r0, float r1, V3f n, float alpha2) {  // material.rs:260-269
    V3f t, b;
    tangent_basis(n, t, b);
    float sn, cs;
    HR_SINCOS_2PI(r0, sn, cs);
    // cos^2 = (1 - r1) / den, and sin^2 = 1 - cos^2 = alpha2 r1 / den taken from the same quotient instead of by subtraction: for a
    // near-mirror GGX (alpha2 -> 0) cos^2 is 1 - O(alpha2), and 1 - cos^2 in fp32 would be rounding noise of ~1e-7, i.e. a half
    // vector tilted by ~4e-4 where the f64 reference (material.rs:260-269, sqrt(1 - cos^2) with 1e-16 noise) tilts it by 1e-8
    const float iden = HR_RCP(1.0f + (alpha2 - 1.0f) * r1);
    float cos_theta = HR_SQRT(fmaxf((1.0f - r1) * iden, 0.0f));
    float sin_theta = HR_SQRT(fmaxf(alpha2 * r1 * iden, 0.0f));
    return t * (sin_theta * cs) + b * (sin_theta * sn) + n * cos_theta;
}
HD float smith_lambda(float xn, float alpha2) { float a = HR_RCP(xn * xn) - 1.0f; return 0.5f * HR_SQRT(1.0f + alpha2 * a) - 0.5f; }
HD float g_smith_joint(float ln, float vn, float alpha2) { return HR_RCP(1.0f + smith_lambda(ln, alpha2) + smith_lambda(vn, alpha2)); }
HD float f_schlick(float vh, float f0) { float x = 1.0f - vh, x2 = x * x; return f0 + (1.0f - f0) * (x * (x2 * x2)); }

HD float bsdf_eval(int32_t surface, float param, float roughness, V3f view, V3f n, V3f light) {  // material.rs:53-89
    if (surface == 0) return 1.0f / PI_F;
    float alpha2 = roughness * roughness;
    V3f h = normalize(light + view);
    float ln = dot(light, n);
    // material.rs:64-67 returns 0 for a NEGATIVE l.n; l.n = +0 EXACTLY goes on to 0 / 0 there as well — in f64 it never happens, in fp32 it does
    // (a shadow ray toward an emitter sample at the very height of a horizontal face: one path in ~10^10, a NaN pixel in a 1,024-sampling
    // frame of cornell_mini).  The limit of bsdf x (l.n) for l.n -> 0+ is 0 (the Smith term vanishes): 0 is what a grazing sample adds.
    if (!(ln > 0.0f)) return 0.0f;
    float vn = dot(view, n), vh = dot(view, h), hn = dot(h, n);
    float tmp = 1.0f - (1.0f - alpha2) * hn * hn;
    float d = alpha2 * HR_RCP(PI_F * tmp * tmp);
    return d * g_smith_joint(ln, vn, alpha2) * f_schlick(vh, param) * HR_RCP(4.0f * ln * vn);
}
// material.rs:154-199
// `transmitted`: true when the ray leaves along the refracted direction (false: Fresnel reflection or total internal reflection) —
// only the per-path event log reads it (PathLog below); dead in every other instantiation
HD void sample_refraction(float r0, V3f pos, V3f view, V3f n, float ior, V3f &no, V3f &nd, float &refl, bool &transmitted) {
    transmitted = false;
    bool incoming = signbit(dot(view, n));
    V3f on = incoming ? n : -n;
    float nnt = incoming ? HR_RCP(ior) : ior;
    V3f rdir = reflect(view, on);
    float vn = dot(view, on);
    float k = 1.0f - nnt * nnt * (1.0f - vn * vn);  // vector.rs:64-71
    if (k < 0.0f) { no = pos + OFFSET_F * on; nd = rdir; refl = 1.0f; return; }
    V3f tdir = nnt * view - (nnt * vn + HR_SQRT(k)) * on;
    if (is_zero(tdir)) { no = pos + OFFSET_F * on; nd = rdir; refl = 1.0f; return; }
    float cos_i = dot(view, -on), cos_t = dot(tdir, -on);
    float a = nnt * cos_i - cos_t, b = nnt * cos_i + cos_t, c = nnt * cos_t - cos_i, d = nnt * cos_t + cos_i;
    float fr = 0.5f * (a * a * HR_RCP(b * b) + c * c * HR_RCP(d * d));
    if (r0 <= fr) { no = pos + OFFSET_F * on; nd = rdir; refl = 1.0f; }
    else { no = pos - OFFSET_F * on; nd = tdir; refl = nnt * nnt; transmitted = true; }
}
// material.rs:91-151; returns false for "sampled below the horizon" (None)
HD bool bsdf_sample(const PointMat &m, float r0, float r1, V3f pos, V3f view, V3f n, V3f &no, V3f &nd, float &refl, bool &transmitted) {
    V3f in = -view;
    transmitted = false;
    switch (m.surface) {
        case 0: no = pos + n * OFFSET_F; nd = sample_diffuse(r0, r1, n); refl = 1.0f; return true;
        case 1: no = pos + n * OFFSET_F; nd = reflect(in, n); refl = 1.0f; return true;
        case 2: sample_refraction(r0, pos, in, n, m.param, no, nd, refl, transmitted); return true;
        case 3: {
            float alpha2 = m.roughness * m.roughness;
            V3f h = sample_ggx_half(r0, r1, n, alpha2);
            V3f l = reflect(in, h);
            float ln = dot(l, n);
            if (signbit(ln)) return false;
            float vn = dot(view, n), vh = dot(view, h), hn = dot(h, n);
            refl = f_schlick(vh, m.param) * saturatef(g_smith_joint(ln, vn, alpha2) * vh * HR_RCP(hn * vn));
            no = pos + n * OFFSET_F; nd = l;
            return true;
        }
        default: {
            V3f h = sample_ggx_half(r0, r1, n, m.roughness * m.roughness);
            sample_refraction(r0, pos, in, h, m.param, no, nd, refl, transmitted);
            return true;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the path state machine
// Kept small on purpose: the trace kernel holds one Path per lane for its whole life, and every register here is one the
// shading code cannot use (the kernel is compiled for 5 waves per SIMD = 96 VGPRs).
struct Path {
    uint32_t q;           // bits 0-5: lane of the tile (pixel, sub-sample), 6-11: sampling inside the launch's batch, 12-15: 2 * a (a = accepted
                          // lens attempt: iteration i reads draws 2a + 2i, 2a + 2i + 1); 0xffffffff = lane idle
    uint32_t tile;        // the 4x4-pixel tile the path belongs to (a wave works on several tiles over its life)
    uint32_t st;          // bits 0-3: iteration 1..9 (renderer.rs:174), 4: phase (0 = main ray in flight, 1 = shadow ray), 5-7: surface type of
                          // the shaded point, 8-31: emitter the shadow ray aims at
    Ray ray;
    TraceState ts;
    V3f accum;
    V3f refl;             // reflectance so far; while shadow rays are in flight: refl * albedo (what one emitter's emission * bsdf * G / pdf
                          // is multiplied by, renderer.rs:183,295) — multiplied by cur_refl afterwards (renderer.rs:197)
    // valid while a shadow ray is in flight (the bounce ray waits in next_o / next_d)
    V3f next_o, next_d;
    float cur_refl;       // reflectance of the sampled bounce (PointMaterial::sample's scalar)
    V3f view, n; float param, roughness;
    float shadow_len;     // |sample point - shadow origin|
    float r0, r1;
};
static const uint32_t PATH_IDLE = 0xffffffffu;
HD uint32_t path_iter(const Path &p) { return p.st & 15u; }
HD bool path_in_shadow_phase(const Path &p) { return (p.st & 16u) != 0u; }
HD int32_t path_surface(const Path &p) { return (int32_t)((p.st >> 5) & 7u); }
HD uint32_t path_emitter(const Path &p) { return p.st >> 8; }
// lane base of the path's hand-off record inside the tile's block of records (device_scene.h rec_slot)
HD uint32_t path_draw_base(const Path &p) { return ((p.q >> 6) & 63u) * REC_ITEM_FLOATS + (p.q & 63u) * 4u; }

// camera.rs:83-96 with the lens rejection loop already resolved by the seed kernel (record head: attempt a, lens x, lens y).
// In: p.q = slot of the path (bits 0-11); `recs` = the tile's block of records.
// The ray is computed in f64 (the reference's camera is f64: device_scene.h CameraD) and rounded ONCE: the fp32 ray the traversal walks is the
// best fp32 ray there is, and what the rounding took away — (o64 - o32, d64 - d32), six floats — is parked in dead slots of the path's own
// record (ray_fix_slot) for the one consumer that needs it: a primary ray that hits a sphere (hit_surface).  ~60 f64 operations per path.
HD void path_start(const Scene &sc, const RenderParams &rp, Path &p, uint32_t px, uint32_t py, uint32_t sub, float *recs) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) CameraD *CamConstPtr;   // constant address space: the 20 doubles are scalar loads
    const CameraD c = *(CamConstPtr)(unsigned long long)sc.camd;
#else
    const CameraD &c = *sc.camd;
#endif
    const uint32_t lb = path_draw_base(p);
    const f4 head = *reinterpret_cast<const f4 *>(recs + rec_slot(lb, REC_HEAD));
    const uint32_t a2 = 2u * float_as_uint(head.x);
    p.q = (p.q & 0xfffu) | (a2 << 12);
    const double fx = (double)px, fy = (double)(rp.height - py);
    const double sx = (double)(sub & 1u) * 0.5 - 0.5, sy = (double)(sub >> 1) * 0.5 - 0.5;
    const double m = (double)(rp.width < rp.height ? rp.width : rp.height);
    double im = (double)HR_RCP((float)m);                       // one f64 reciprocal (v_rcp_f32 seed + two Newton steps) instead of two f64 divisions
    im = im * (2.0 - m * im); im = im * (2.0 - m * im);
    const double ncx = ((fx + sx) * 2.0 - (double)rp.width) * im, ncy = ((fy + sy) * 2.0 - (double)rp.height) * im;   // renderer.rs:53-54
    const double lx = (double)head.y * c.lens_radius, ly = (double)head.z * c.lens_radius;
    const double lpx = c.right[0] * lx + c.up[0] * ly, lpy = c.right[1] * lx + c.up[1] * ly, lpz = c.right[2] * lx + c.up[2] * ly;
    const double vx = ncx * c.phr[0] + ncy * c.phu[0] + c.focus_distance * c.forward[0] - lpx;
    const double vy = ncx * c.phr[1] + ncy * c.phu[1] + c.focus_distance * c.forward[1] - lpy;
    const double vz = ncx * c.phr[2] + ncy * c.phu[2] + c.focus_distance * c.forward[2] - lpz;
    const double il = hr_rsqrt_f64(fma(vx, vx, fma(vy, vy, vz * vz)));
    const double o64x = c.eye[0] + lpx, o64y = c.eye[1] + lpy, o64z = c.eye[2] + lpz, d64x = vx * il, d64y = vy * il, d64z = vz * il;
    const V3f o = v3((float)o64x, (float)o64y, (float)o64z), d = v3((float)d64x, (float)d64y, (float)d64z);
    ray_fix_store(recs, lb, a2, v3((float)(o64x - (double)o.x), (float)(o64y - (double)o.y), (float)(o64z - (double)o.z)),
                  v3((float)(d64x - (double)d.x), (float)(d64y - (double)d.y), (float)(d64z - (double)d.z)));
    ray_set(p.ray, o, d);
    ray_quantise(sc, p.ray);
    p.st = 1u;            // iteration 1, main ray
    p.accum = v3(0, 0, 0); p.refl = v3(1, 1, 1);
    trace_begin(p.ts, T_INF, p.ray.start);
}

// scene.rs:92-101 + renderer.rs:276-279: shadow ray toward the sample point on emitter path_emitter(p).
// Returns false when the ray need not be traced at all, because its contribution is known to be exactly zero (round 5; `cull` = 0
// switches the shortcuts off: debug option nee_cull 0, the A/B and the bit-equality test; bit 0 = (1), bit 1 = (2); bit 2 is reserved — a third
// shortcut, "the shaded sphere itself in the way", is exact too and was dropped: one more live register cost more than its rays gave — the
// logging instantiation keeps (2) off, because the log records the visibility verdict of renderer.rs:280 also where the BSDF is zero):
//   (1) the sample point lies on the FAR side of the emitter.  sample_on_surface draws uniformly over the whole sphere (scene.rs:92-101), so
//       for more than half of the samples the shadow ray runs through the emitter itself before it reaches the sample: the closest hit of
//       renderer.rs:279 is then the emitter's near side (or something nearer still), `approximately` (vector.rs:89-91) fails, nothing is
//       added.  With sn the unit normal at the sample and d the ray direction, x = (r + OFFSET) (sn . d) is the distance from the sample
//       back to the ray's point of closest approach to the centre; the ray is inside the sphere of radius r over at least the last x of
//       its way whenever x^2 > 2 r OFFSET + OFFSET^2 (it enters the sphere at all).  The walk would stop at the first hit more than 0.0201 in
//       front of the sample (shadow_early_out); x > 0.0201 + slack therefore decides the same thing without walking.  Slack: the fp32 ray
//       misses the sample point by ~2e-7 of its length — 2e-3 + 1e-6 L on x and a factor 2 on the entry condition are far more than that.
//       Marginal samples (a chord of about 0.02: within 1.3 degrees of the silhouette of an r = 1 emitter) are traced as before.
//   (2) GGX, emitter below the shaded point's horizon: material.rs:64-67 returns 0 (`l_dot_n.is_sign_negative()`), the contribution is
//       emission * 0 — nothing is added whatever the shadow ray finds (bsdf_eval tests the same sign bit on the same two vectors).
// A culled ray is one the reference traces and then discards; the accumulator is the same to the bit (test_nee_culls_do_not_change_a_bit).
HD bool nee_setup(const Scene &sc, Path &p, uint32_t cull) {
    const Emitter em = sc.emitters[path_emitter(p)];
    float unit_z = 1.0f - 2.0f * p.r1;
    float a = HR_SQRT(fmaxf(1.0f - unit_z * unit_z, 0.0f));
    float sn_, cs_;
    HR_SINCOS_2PI(p.r0, sn_, cs_);
    V3f sn = v3(a * cs_, a * sn_, unit_z);
    const float ro = em.r + OFFSET_F;
    V3f sp = v3(em.c) + ro * sn;
    V3f sv = sp - p.next_o;
    float sl2 = dot(sv, sv), isl = HR_RSQ(sl2);
    const float len = sl2 * isl;
    const V3f d = sv * isl;
    if (cull) {
        const float slack = 0.0221f + 1e-6f * len;
        const float x = ro * dot(sn, d);
        const bool far_side = (cull & 1u) && x > slack && x * x > 2.0f * (2.0f * em.r * OFFSET_F + OFFSET_F * OFFSET_F) && len > 2.0f * x;   // (len > 2 x: the origin lies in front of the entry point, which is at most 2 x before the sample)
        const float nd = dot(p.n, d);
        const bool ggx_below = (cull & 2u) && path_surface(p) == 3 && signbit(nd);
        if (far_side || ggx_below) return false;
    }
    p.shadow_len = len;
    ray_set(p.ray, p.next_o, d);
    ray_quantise(sc, p.ray);
    // a closest hit beyond the sample point can never pass the proximity test (vector.rs:89-91: |dp|^2 < 4e-4),
    // so the search is limited to the sample distance + 0.03 (the reference does an unbounded closest-hit query)
    trace_begin(p.ts, p.shadow_len + 0.03f, p.ray.start);
    p.st |= 16u;
    return true;
}

// A shadow ray only contributes when its closest hit lies within 0.02 of the sample point (renderer.rs:280-282 with
// vector.rs:89-91: |dp|^2 < 4e-4).  As soon as ANY hit is farther than that in front of the sample point the closest hit
// is too, so the rest of the walk cannot change the outcome: stop (called after every leaf).
HD void shadow_early_out(Path &p) {
    if (path_in_shadow_phase(p) && p.ts.t < p.shadow_len - 0.0201f) { p.ts.cur = NODE_END; p.ts.leaf = 0; p.ts.leaf2 = 0; }
}

HD int32_t hit_element(const Scene &sc, const TraceState &ts) {
    return (ts.type == 0) ? sc.tri_shade[ts.prim].element : (ts.type == 1 ? sc.sphere_elem[ts.prim] : float_as_int(sc.cuboids[2 * ts.prim].w));
}

}  // namespace hr
#include "prec_core.h"
namespace hr {

// Per-path event log (hr_debug_path_log: the parity accounting of tests/test_gpu_parity.py and profiles/r04_parity_report.json).  The
// oracle keeps the same log (oracle.cpp PathLog): two paths "took the same branches" when their logs are equal.
//   one byte per iteration i = 1..9 (renderer.rs:174), byte i - 1 of ev (i <= 8) / ev9:
//     bits 0-2  0 = iteration not reached, 1 = the ray missed (sky), 2 + surface type = hit and sampled (2 Diffuse, 3 Specular, 4 Refraction,
//               5 GGX, 6 GGXRefraction), 7 = hit, PointMaterial::sample returned None (GGX half vector below the horizon, material.rs:119-121)
//     bit  3    Refraction / GGXRefraction: the ray was transmitted (0: Fresnel reflection or total internal reflection, material.rs:163-199)
//     bits 4-7  NEE (Diffuse / GGX only): bit 4 + (k mod 4) set when the shadow ray towards emitter k passed the visibility test of renderer.rs:280
//   hash: FNV-style hash over the element indices — and, for meshes, the input triangle indices — the main rays hit, in order (another
//         sphere of the same material, or the neighbouring triangle with another normal, is another branch)
//   rays: scene.intersect calls of the path (main + shadow rays)
//   ev9 bits 8-15: how many of the path's main rays hit a SPHERE (the one primitive that multiplies a position error by 1 / radius)
// Round 5 — every discrete decision of a main ray is in the log: besides the elements and triangles, `hash` takes the FACE of a cuboid hit
// (scene.rs:160-182's cascade: another face is another normal) and the cube-map FACE of the sky lookup that ends a path (scene.rs:295-319:
// every face is a texture of its own, clamped at its border — a flip at a seam is a jump); and ev9 bits 16-31 carry a 16-bit sum over the
// TEXEL QUADS (integer corner x1, y1 of texture.rs:29-49) of every image the main rays sampled — surface textures and the sky.  Bilinear
// interpolation is continuous across quad borders, so another quad is NOT another branch: paths are "same" by events + hash, and the quad sum
// says how many of them interpolated between other texels (tests/path_parity.py `other_texel_quad`).  All of it is computed by LOG-only helpers
// beside the production code (plog_sky, plog_quad, cuboid_face_of), from the same fp32 values.
struct PathLog { unsigned long long ev; uint32_t ev9, hash, rays; };
HD void plog_reset(PathLog &l) { l.ev = 0ull; l.ev9 = 0u; l.hash = 0x811c9dc5u; l.rays = 0u; }
HD void plog_or(PathLog &l, uint32_t iter, uint32_t bits) {
    if (iter <= 8u) l.ev |= (unsigned long long)bits << ((iter - 1u) * 8u);
    else l.ev9 |= bits;
}
HD void plog_hit(PathLog &l, int32_t elem) { l.hash = (l.hash ^ (uint32_t)(elem + 1)) * 0x01000193u; }
HD uint32_t quad_hash16(uint32_t ix, uint32_t iy) { return ((ix * 0x9E3779B1u) ^ (iy * 0x85EBCA77u)) >> 16; }
HD void plog_quad_add(PathLog &l, uint32_t ix, uint32_t iy) { l.ev9 = (l.ev9 & 0xffffu) | ((((l.ev9 >> 16) + quad_hash16(ix, iy)) & 0xffffu) << 16); }
// the corner (x1, y1) sample_bilinear() starts from (texture.rs:30-33), from the same fp32 products
HD void plog_quad(const Scene &sc, PathLog &l, int32_t image, float u, float v) {
    if (image < 0) return;
    const ImageRef im = sc.images[image];
    plog_quad_add(l, f32_as_u32_sat(floorf(u * (float)im.width)), f32_as_u32_sat(floorf(v * (float)im.height)));
}
// face and corner of sky_sample()'s lookup: the same comparisons and quotients (scene.rs:295-319)
HD void plog_sky(const Scene &sc, PathLog &l, V3f d) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face; float u, v;
    if (ax > ay && ax > az) {
        float i = HR_RCP(d.x);
        if (!signbit(d.x)) { face = 0; u = -d.z * i; v = d.y * i; } else { face = 1; u = -d.z * i; v = -d.y * i; }
    } else if (ay > ax && ay > az) {
        float i = HR_RCP(d.y);
        if (!signbit(d.y)) { face = 2; u = d.x * i; v = -d.z * i; } else { face = 3; u = -d.x * i; v = -d.z * i; }
    } else {
        float i = HR_RCP(d.z);
        if (!signbit(d.z)) { face = 4; u = d.x * i; v = d.y * i; } else { face = 5; u = d.x * i; v = -d.y * i; }
    }
    u = 0.5f * (u + 1.0f); v = 0.5f * (v + 1.0f);
    plog_hit(l, 0x2000 + face);
    uint32_t w = sc.sky_w, h = sc.sky_h;
    if (!sc.sky_quads) { const ImageRef im = sc.images[sc.sky_image[face]]; w = im.width; h = im.height; }
    plog_quad_add(l, f32_as_u32_sat(floorf(u * (float)w)), f32_as_u32_sat(floorf(v * (float)h)));
}
// which face of scene.rs:160-182's cascade a cuboid hit took, from the normal hit_surface() chose
HD int32_t cuboid_face_of(V3f n) { return n.y > 0.0f ? 0 : n.y < 0.0f ? 1 : n.x < 0.0f ? 2 : n.x > 0.0f ? 3 : n.z < 0.0f ? 4 : n.z > 0.0f ? 5 : 6; }

// returns true when the path is finished (accum final)
// RR / rr_start: Russian roulette from that iteration on (off = the reference's estimator: renderer.rs:174-200 has none).  A template
// parameter: the roulette's few instructions in the default kernel cost it 2 % next to the seed kernel (measured), so the default
// instantiation does not contain them.
// The roulette's uniform variates come from an integer hash of (tile, lane, sampling, iteration) — NOT from the path's ISAAC-64
// stream, whose draws the reference estimator has all spoken for (and whose seed kernel is not to be touched: a spare word in the
// hand-off record was tried and cost the hand-scheduled consumer wave 1 % by its changed register allocation).
HD float rr_uniform(uint32_t tile, uint32_t q, uint32_t salt, uint32_t iter) {
    uint32_t x = tile * 0x9E3779B1u ^ (q & 0xfffu) * 0x85EBCA77u ^ salt * 0xC2B2AE3Du ^ iter * 0x27D4EB2Fu;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}
// PREC (option precise_shading): the geometry of the bounce in f64 (prec_core.h shade_hit_f64).  The ray's residuals — what rounding the f64 ray
// to the fp32 ray the traversal walks took away — do not live in registers: they are parked in the six dead slots of the path's own record that
// path_start uses for the camera ray's (ray_fix_slot), read when the main ray's hit is shaded and overwritten with the next ray's.
template <bool CNT, bool RR = false, bool LOG = false, bool PREC = false>
HD bool path_advance(const Scene &sc, const RenderParams &rp, Path &p, const float *recs, LaneCounters *cn, uint32_t rr_start = 0u, uint32_t rr_salt = 0u, PathLog *lg = nullptr) {
    if (CNT) cn->rays++;
    if (LOG) lg->rays++;
    const bool hit = p.ts.prim >= 0;
#if defined(HR_PATH_VERBOSE) && !defined(__HIP_DEVICE_COMPILE__)   // host emulation only: one line per finished ray of the path HR_V selects (tests/emu)
    if (getenv("HR_V")) fprintf(stderr, "HIP it %u shadow %d hit %d t %.9g o %.9g %.9g %.9g d %.9g %.9g %.9g accum %.9g %.9g %.9g refl %.9g %.9g %.9g\n", path_iter(p), (int)path_in_shadow_phase(p), (int)hit, p.ts.t, p.ray.o.x, p.ray.o.y, p.ray.o.z, p.ray.d.x, p.ray.d.y, p.ray.d.z, p.accum.x, p.accum.y, p.accum.z, p.refl.x, p.refl.y, p.refl.z);
#endif
    if (!path_in_shadow_phase(p)) {
        const f2v r01 = *reinterpret_cast<const f2v *>(recs + rec_slot(path_draw_base(p), ((p.q >> 12) & 15u) + 2u * path_iter(p)));   // renderer.rs:175
        p.r0 = r01[0]; p.r1 = r01[1];
        if (!hit) {  // scene.rs:398 + renderer.rs:196,199
            if (LOG) { plog_or(*lg, path_iter(p), 1u); plog_sky(sc, *lg, p.ray.d); }
            p.accum = p.accum + p.refl * sky_sample(sc, p.ray.d);
            return true;
        }
        if (PREC) {
            const uint32_t lb = path_draw_base(p), a2 = (p.q >> 12) & 15u;
            V3f fo, fd;
            ray_fix_load(recs, lb, a2, fo, fd);
            double r0 = (double)p.r0, r1 = (double)p.r1;
            if (rp.rec_lo_off) {    // the draws' residuals (device_scene.h: the records' twin)
                const f2v l01 = *reinterpret_cast<const f2v *>(recs + rp.rec_lo_off + rec_slot(lb, a2 + 2u * path_iter(p)));
                r0 += (double)l01[0]; r1 += (double)l01[1];
            }
            PrecHit x;
            shade_hit_f64(sc, p.ray.o, p.ray.d, fo, fd, p.ts, r0, r1, x);
            p.view = -p.ray.d;
            if (LOG) {
                if (p.ts.type == 2) plog_hit(*lg, 0x1000 + cuboid_face_of(x.nf));
                const Material mt = sc.materials[x.s.elem];
                plog_quad(sc, *lg, mt.albedo_img, x.s.u, x.s.v); plog_quad(sc, *lg, mt.emission_img, x.s.u, x.s.v); plog_quad(sc, *lg, mt.roughness_img, x.s.u, x.s.v);
                plog_hit(*lg, x.s.elem);
                if (p.ts.type == 1) lg->ev9 += 256u;
                if (p.ts.type == 0) plog_hit(*lg, (int32_t)(sc.tri_face[p.ts.prim] + 0x9e3779b9u));
                plog_or(*lg, path_iter(p), x.sampled ? (2u + (uint32_t)x.m.surface) | (x.transmitted ? 8u : 0u) : 7u);
            }
            if (!x.sampled) return true;
            p.accum = p.accum + p.refl * x.m.emission;
            p.refl = p.refl * x.m.albedo;
            p.next_o = narrow(x.no); p.next_d = narrow(x.nd); p.cur_refl = x.cur_refl;
            ray_fix_store(const_cast<float *>(recs), lb, a2, residual(x.no, p.next_o), residual(x.nd, p.next_d));
            if (!(nee_available(x.m.surface) && sc.num_emitters > 0)) goto bounce;
            p.n = x.nf; p.param = x.m.param; p.roughness = x.m.roughness;
            p.st = (p.st & 15u) | ((uint32_t)x.m.surface << 5);
        } else {
        Surf s;
        RayFix fix = no_ray_fix();
        if (p.ts.type == 1 && path_iter(p) == 1u) {   // a primary ray on a sphere: hit point and normal from the f64 camera ray (path_start's residuals)
            const uint32_t lb = path_draw_base(p), a2 = (p.q >> 12) & 15u;
            ray_fix_load(recs, lb, a2, fix.o, fix.d);
        }
        hit_surface(sc, p.ray, p.ts, material_needs_uv(sc, hit_element(sc, p.ts)), s, fix);
        PointMat m;
        material_at(sc, s.elem, s.u, s.v, m);
        p.view = -p.ray.d;
        bool transmitted;
        const bool sampled = bsdf_sample(m, p.r0, p.r1, s.pos, p.view, s.n, p.next_o, p.next_d, p.cur_refl, transmitted);
        if (LOG) {
            if (p.ts.type == 2) plog_hit(*lg, 0x1000 + cuboid_face_of(s.n));
            const Material mt = sc.materials[s.elem];
            plog_quad(sc, *lg, mt.albedo_img, s.u, s.v); plog_quad(sc, *lg, mt.emission_img, s.u, s.v); plog_quad(sc, *lg, mt.roughness_img, s.u, s.v);
        }
        if (LOG) { plog_hit(*lg, s.elem); if (p.ts.type == 1) lg->ev9 += 256u; if (p.ts.type == 0) plog_hit(*lg, (int32_t)(sc.tri_face[p.ts.prim] + 0x9e3779b9u)); plog_or(*lg, path_iter(p), sampled ? (2u + (uint32_t)m.surface) | (transmitted ? 8u : 0u) : 7u); }
        if (!sampled) return true;  // renderer.rs:190-193
        p.accum = p.accum + p.refl * m.emission;          // renderer.rs:196
        p.refl = p.refl * m.albedo;                       // renderer.rs:183,295 (NEE scale) and the first factor of :197
        if (!(nee_available(m.surface) && sc.num_emitters > 0)) goto bounce;
        p.n = s.n; p.param = m.param; p.roughness = m.roughness;
        p.st = (p.st & 15u) | ((uint32_t)m.surface << 5);   // emitter 0; the phase bit is set by nee_setup
        }
    } else {
        // renderer.rs:280-292.  Hit point and sample point lie on the same ray: |hit - sample| = |t - shadow_len|
        float dt = p.ts.t - p.shadow_len;
        if (hit && dt * dt < OFFSET_F * 4.0f) {
            const Material mt = sc.materials[hit_element(sc, p.ts)];
            V3f e = v3(mt.emission);
            if (mt.emission_img >= 0) {
                Surf s;
                hit_surface(sc, p.ray, p.ts, true, s);
                e = tex_sample(sc, mt.emission_img, e, s.u, s.v);
            }
            const Emitter em = sc.emitters[path_emitter(p)];
            V3f sp = p.ray.o + p.ray.d * p.shadow_len;
            V3f sn = (sp - v3(em.c)) * HR_RCP(em.r + OFFSET_F);
            float dot_0 = fabsf(dot(p.n, p.ray.d)), dot_l = fabsf(dot(sn, p.ray.d));
            float g = (dot_0 * dot_l) * HR_RCP(p.shadow_len * p.shadow_len);
            float inv_pdf = 4.0f * PI_F * em.r * em.r;
            float w = bsdf_eval(path_surface(p), p.param, p.roughness, p.view, p.n, p.ray.d) * g * inv_pdf;
            p.accum = p.accum + p.refl * (e * w);
            if (LOG) plog_or(*lg, path_iter(p), 16u << (path_emitter(p) & 3u));
        }
        p.st = (p.st & ~16u) + 256u;     // next emitter
    }
    // the next emitter whose shadow ray has to be traced (renderer.rs:274: `for emission in emissions`, in order); the ones nee_setup
    // culls are rays the reference traces and discards — they count as rays of the path in the log, not in the kernel's counters
    for (; path_emitter(p) < sc.num_emitters; p.st += 256u) {
        if (nee_setup(sc, p, (LOG ? 5u : 7u) & ~rp.nee_cull_off)) return false;
        if (CNT) cn->shadow_culled++;
        if (LOG) lg->rays++;
    }
    p.st &= 15u;      // back to the main ray
bounce:
    // renderer.rs:197-199 (a miss returned above)
    p.refl = p.refl * p.cur_refl;
    if (is_zero(p.refl) || path_iter(p) >= 9u) return true;
    if (RR && path_iter(p) + 1u >= rr_start) {
        // NOT the reference's estimator (option "russian_roulette", off by default).  The path enters iteration i + 1 with probability
        // q = max(reflectance), capped at 1, and its reflectance is divided by q: E[weight] = 1 whatever the path did so far.
        const float q = fminf(fmaxf(fmaxf(p.refl.x, p.refl.y), p.refl.z), 1.0f);
        if (!(rr_uniform(p.tile, p.q, rr_salt, path_iter(p)) < q)) return true;
        p.refl = p.refl * HR_RCP(q);
    }
    p.st++;
    ray_set(p.ray, p.next_o, p.next_d);
    ray_quantise(sc, p.ray);
    trace_begin(p.ts, T_INF, p.ray.start);
    return false;
}

// ---------------------------------------------------------------------------------------------
// DebugRenderer::calc_pixel (renderer.rs:116-139): pinhole ray (camera.rs:98-107), no RNG.
// mode 0 Shading, 1 Normal, 2 Depth, 3 FocalPlane.  Also serves as a traversal-only workload (hr_render_debug walks the production
// traversal: debug_render_kernel in trace_kernel.h).  In pieces, so that the kernel and the host emulation share everything but the walk.
HD void debug_camera_ray(const Scene &sc, const RenderParams &rp, uint32_t px, uint32_t py, uint32_t sub, Ray &ray) {
    float fx = (float)px, fy = (float)(rp.height - py);
    float ox = (float)(sub & 1u) * 0.5f - 0.5f, oy = (float)(sub >> 1) * 0.5f - 0.5f;
    float m = (float)(rp.width < rp.height ? rp.width : rp.height);
    float ncx = ((fx + ox) * 2.0f - (float)rp.width) * HR_RCP(m), ncy = ((fy + oy) * 2.0f - (float)rp.height) * HR_RCP(m);
    const CameraF &c = sc.cam;
    ray_set(ray, v3(c.eye), normalize(ncx * v3(c.phr) + ncy * v3(c.phu) + c.focus_distance * v3(c.forward)));
}
// The primary ray's result.  Returns true when the pixel still needs the shadow ray `sh` of the Shading mode (renderer.rs:121-131):
// the pixel is then val + lit * (shadow ray hit something ? 0.5 : 1); otherwise it is val.
HD bool debug_primary(const Scene &sc, const Ray &ray, const TraceState &ts, int mode, V3f &val, V3f &lit, Ray &sh) {
    if (ts.prim < 0) { val = sky_sample(sc, ray.d); return false; }
    Surf s;
    hit_surface(sc, ray, ts, material_needs_uv(sc, hit_element(sc, ts)), s);
    if (mode == 1) { val = s.n; return false; }
    if (mode == 2) { float v = 0.5f * ts.t * HR_RCP(sc.cam.focus_distance); val = v3(v, v, v); return false; }
    if (mode == 3) { float v = fabsf(ts.t - sc.cam.focus_distance); val = v3(v, v, v); return false; }
    PointMat pm;
    material_at(sc, s.elem, s.u, s.v, pm);
    const V3f light = normalize(v3(1.0f, 2.0f, -1.0f));
    ray_set(sh, s.pos + s.n * OFFSET_F, light);
    val = pm.emission;
    lit = pm.albedo * fmaxf(dot(s.n, light), 0.0f);
    return true;
}
// scalar form (host emulation): one node + its leaf per step
template <bool CNT>
HD V3f debug_pixel(const Scene &sc, const RenderParams &rp, uint32_t px, uint32_t py, uint32_t sub, int mode, LaneCounters *cn) {
    Ray ray, sh;
    debug_camera_ray(sc, rp, px, py, sub, ray);
    TraceState ts;
    trace_begin(ts, T_INF);
    while (ts.cur != NODE_END) trace_step<CNT>(sc, ray, ts, cn);
    if (CNT) cn->rays++;
    V3f val, lit;
    if (!debug_primary(sc, ray, ts, mode, val, lit, sh)) return val;
    TraceState st;
    trace_begin(st, T_INF);
    while (st.cur != NODE_END) trace_step<CNT>(sc, sh, st, cn);
    if (CNT) cn->rays++;
    return val + lit * (st.prim >= 0 ? 0.5f : 1.0f);
}

}  // namespace hr
