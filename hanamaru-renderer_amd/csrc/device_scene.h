// Device-side scene layout (fp32, SoA-ish, all in HBM; the geometry part is ~1 MB and lives in L2).
//
//   qnodes[o][]  16 B   the record the trace kernel walks (QNode below): six 16-bit grid planes + one link, one copy per ray-direction
//                       octant o, each in its own near-first preorder; rtcamp6_v3_1: 8 x 13,950 x 16 B = 1.8 MB, L2-resident
//   nodes[o][]   32 B   the same trees with fp32 near / far planes and explicit hit / miss links (Node below): debug kernels and the
//                       quant_nodes = 0 variant of the trace kernel.  One record = two 16-byte loads
//   tris[]       48 B   leaf-ordered TriT: unit normal, the two barycentric gradients, v0 — what the intersection test consumes
//   tri_shade[]  16 B   leaf-ordered TriS: unit normal + element id — what shading a triangle hit consumes
//                       (both derived in f64 by tri_derive() from the geometry record Tri: v0, e1 = v1-v0, e2 = v2-v0 — edges formed in
//                       f64, then rounded — + element id; Tri is what the builders read and is not kept after the upload of a host-built tree)
//   spheres[]    16 B   centre, radius            (+ sphere_elem[])
//   cuboids[]    32 B   min, max                  (+ element id in .w of the first float4)
//   materials[]  64 B   per element
//   texels[]      4 B   RGBA8, all images back to back; images[] = {offset, width, height}
//   sky_quads[]   8 B   the six skybox faces once more, as column pairs: a bilinear footprint is 16 contiguous bytes (Scene::sky_quads below)
//
// "Algorithmic bytes"  are counted as 32 B per node test, 36 B per triangle test,
// 16 B per sphere test, 24 B per cuboid test — the information content, not the padded layout.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

namespace hr {

struct alignas(16) f4 { float x, y, z, w; };

// The box is stored per ray-direction octant as NEAR and FAR planes (near = min where the ray travels in +axis, max where it
// travels in -axis), in the pairs the slab test consumes with packed fp32 math (v_pk_add_f32 / v_pk_mul_f32):
// {near.x, near.y} {far.x, far.y} | {near.z, far.z} a b — two 16-byte loads.  Entry distance = max3 of the near terms, exit
// distance = min3 of the far terms: no per-axis min / max (bvh.rs:26-33 needs them because it does not know the direction).
struct alignas(16) Node {
    float nearx, neary, farx, fary;
    float nearz, farz;
    uint32_t a;       // box hit -> inner: next node; leaf: the leaf word (leaf_word() below)
    uint32_t b;       // box missed (or leaf done) -> next node in this octant's order, NODE_END = finished
};
HD void node_set_box(Node &n, const float *mn, const float *mx, int octant) {
    n.nearx = (octant & 1) ? mx[0] : mn[0]; n.farx = (octant & 1) ? mn[0] : mx[0];
    n.neary = (octant & 2) ? mx[1] : mn[1]; n.fary = (octant & 2) ? mn[1] : mx[1];
    n.nearz = (octant & 4) ? mx[2] : mn[2]; n.farz = (octant & 4) ? mn[2] : mx[2];
}
// links: successors (node indices, or byte offsets on the 16-byte records) are below 2^31 - 1, NODE_END = 2^31 - 1 ends the walk, and a
// word with bit 31 set is a leaf word — one compare tells a leaf word from a successor.
//   leaf word = 1 << 31 | type << 28 (0 triangle, 1 sphere, 2 cuboid) | count << 24 (1 .. 15 = option max_leaf's range) | first (24 bits):
//   `count` primitives of one type from index `first` of that type's leaf-ordered array — 2^24 = 16.7 M references per type (until round 4
//   the word was (type + 1) << 28 | count << 20 | first, 8 count bits and 20 index bits: scenes stopped at 2^20 primitives).
static const uint32_t NODE_END = 0x7fffffffu;
static const uint32_t LEAF_FLAG = 0x80000000u;
static const uint32_t LEAF_FIRST_BITS = 24u, LEAF_MAX_COUNT = 15u;
static const uint32_t MAX_PRIMS_PER_TYPE = 1u << LEAF_FIRST_BITS;
HD bool node_word_is_leaf(uint32_t a) { return a >= LEAF_FLAG; }
HD uint32_t leaf_word(uint32_t type, uint32_t count, uint32_t first) { return LEAF_FLAG | (type << 28) | (count << LEAF_FIRST_BITS) | first; }
HD uint32_t leaf_type(uint32_t w) { return (w >> 28) & 3u; }
HD uint32_t leaf_count(uint32_t w) { return (w >> LEAF_FIRST_BITS) & LEAF_MAX_COUNT; }
HD uint32_t leaf_first(uint32_t w) { return w & (MAX_PRIMS_PER_TYPE - 1u); }

// The trace kernel's node: 16 bytes, ONE load per visit.  The six planes are 16-bit coordinates on a grid over the scene's box
// (plane = qmin + q * qstep per axis), rounded outward by at least one step — a box that only grows can add node visits, never
// lose a hit.  The links are implicit in the layout: every octant's copy is stored in its own near-first preorder, so an inner
// node's near child is record cur + 1 and `link` is its miss successor; a leaf's `link` is the leaf word and its successor is
// cur + 1 (as byte offsets: + 16).  Record N of every copy is a sentinel that nothing hits and whose miss successor is NODE_END.
struct alignas(16) QNode {
    uint32_t xy_near;   // near.x | near.y << 16
    uint32_t xy_far;    // far.x  | far.y  << 16
    uint32_t z_nf;      // near.z | far.z  << 16
    uint32_t link;      // inner: successor when the box is missed, as its BYTE OFFSET in qnodes[] (or NODE_END); leaf: the leaf word
};
// The walk's position on the 16-byte records is a byte offset into the whole qnodes[8][N + 1] array (octant copy included): a visit's
// address is base + offset with no arithmetic, the near child / the node behind a leaf is offset + 16, and a miss takes `link` as
// it stands.  Offsets stay below 2^31 (leaf words have bit 31 set): N + 1 <= 2^24 records per octant copy (2 GiB of records; a tree over
// 2^24 references with leaves of 4 has ~10^7 nodes).
HD uint32_t qnode_offset(int octant, uint32_t records_per_copy, uint32_t index) { return ((uint32_t)octant * records_per_copy + index) * 16u; }
HD uint32_t qnode_link(int octant, uint32_t records_per_copy, uint32_t successor_index) {
    return successor_index == NODE_END ? NODE_END : qnode_offset(octant, records_per_copy, successor_index);
}
// grid of the 16-bit planes: the root box, a little enlarged, in 65,532 steps.  Outward rounding leaves a margin for the kernel's
// fp32 arithmetic: distance = q * (qstep * inv) + (qmin - o) * inv carries three roundings of ~6e-8 relative on values up to
// 65,535 steps — 0.004 step each; QMARGIN = 0.05 step covers them with room.
HD void qframe_from_box(const float *root_mn, const float *root_mx, float *qmin, float *qstep) {
    for (int a = 0; a < 3; a++) {
        double ext = (double)root_mx[a] - (double)root_mn[a];
        if (!(ext > 1e-6)) ext = 1e-6;
        double gmin = (double)root_mn[a] - 1e-6 * ext;
        float fmin_ = (float)gmin;
        fmin_ -= (fmin_ < 0 ? -fmin_ : fmin_) * 1e-6f + 1e-30f;         // the fp32 frame may only sit lower / be coarser
        qmin[a] = fmin_;
        qstep[a] = (float)(ext * (1.0 + 4e-6) / 65532.0) * (1.0f + 1e-6f);
    }
}
HD uint32_t q_low(float v, float qmin, float qstep) {    // largest grid coordinate whose plane is safely <= v
    double q = floor(((double)v - (double)qmin) / (double)qstep - 0.05);
    return (uint32_t)(q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q));
}
HD uint32_t q_high(float v, float qmin, float qstep) {   // smallest grid coordinate whose plane is safely >= v
    double q = ceil(((double)v - (double)qmin) / (double)qstep + 0.05);
    return (uint32_t)(q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q));
}
HD QNode qnode_make(const float *mn, const float *mx, int octant, const float *qmin, const float *qstep, uint32_t link) {
    uint32_t nearq[3], farq[3];
    for (int a = 0; a < 3; a++) {
        uint32_t lo = q_low(mn[a], qmin[a], qstep[a]), hi = q_high(mx[a], qmin[a], qstep[a]);
        nearq[a] = ((octant >> a) & 1) ? hi : lo;
        farq[a] = ((octant >> a) & 1) ? lo : hi;
    }
    QNode q;
    q.xy_near = nearq[0] | (nearq[1] << 16);
    q.xy_far = farq[0] | (farq[1] << 16);
    q.z_nf = nearq[2] | (farq[2] << 16);
    q.link = link;
    return q;
}
// record N of every octant's copy: entered at the far end on every axis (entry distance > exit distance), its miss successor ends the walk
HD QNode qnode_sentinel(int octant) {
    QNode q;
    uint32_t nearq[3], farq[3];
    for (int a = 0; a < 3; a++) { nearq[a] = ((octant >> a) & 1) ? 0u : 65535u; farq[a] = ((octant >> a) & 1) ? 65535u : 0u; }
    q.xy_near = nearq[0] | (nearq[1] << 16);
    q.xy_far = farq[0] | (farq[1] << 16);
    q.z_nf = nearq[2] | (farq[2] << 16);
    q.link = NODE_END;
    return q;
}


struct alignas(16) Tri {
    float v0[3]; float e1x;
    float e1y, e1z, e2x, e2y;
    float e2z; int32_t element; uint32_t face; float pad1;   // face: index of the input triangle (over all meshes, in element order) — only the per-path event log reads it
};

// The records the kernels read for a triangle, derived from Tri (the fp32 triangle v0, v0 + e1, v0 + e2 IS the geometry: everything
// below is computed in f64 from those nine floats and rounded once).  With n = e1 x e2:
//   test:  t = -(nu . dd) / (nu . d) for dd = o - v0 (nu = n / |n|: the quotient does not depend on the length of n, the unit vector
//          keeps both dot products well inside the fp32 range for any triangle size), p = dd + t d, u = a . p, v = b . p with
//          a = (e2 x n) / |n|^2, b = (n x e1) / |n|^2  (a . e1 = 1, a . e2 = 0, b . e1 = 0, b . e2 = 1: the gradients of the barycentrics).
//          Same t as Cramer's rule on (e1, e2, -d) (bvh.rs:266-290); u and v come out of 6 FMAs instead of a cross product and two dots.
//   shade: the normal normalize(e1 x e2) of bvh.rs:286 (never flipped) and the element id in ONE 16-byte load.
struct alignas(16) TriT {
    float n[3]; float ax;
    float ay, az, bx, by;
    float bz; float v0[3];
};
struct alignas(16) TriS { float n[3]; int32_t element; };
// the plane of an INPUT triangle in the reference's precision (unit normal of the f64 vertices, bvh.rs:286; v0): precise shading only (wf_core.h)
struct alignas(16) TriX { double n[3], v0[3]; };
HD void tri_derive(const Tri &t, TriT &tt, TriS &ts) {
    const double e1[3] = {t.e1x, t.e1y, t.e1z}, e2[3] = {t.e2x, t.e2y, t.e2z};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    // a degenerate triangle (nn == 0) gets n = a = b = 0: nu . d == 0 is the reference's det == 0, which rejects (bvh.rs:271)
    const double inn = nn > 0.0 ? 1.0 / nn : 0.0, il = nn > 0.0 ? 1.0 / sqrt(nn) : 0.0;
    const double a[3] = {(e2[1] * n[2] - e2[2] * n[1]) * inn, (e2[2] * n[0] - e2[0] * n[2]) * inn, (e2[0] * n[1] - e2[1] * n[0]) * inn};
    const double b[3] = {(n[1] * e1[2] - n[2] * e1[1]) * inn, (n[2] * e1[0] - n[0] * e1[2]) * inn, (n[0] * e1[1] - n[1] * e1[0]) * inn};
    for (int k = 0; k < 3; k++) { tt.n[k] = (float)(n[k] * il); tt.v0[k] = t.v0[k]; ts.n[k] = (float)(n[k] * il); }
    tt.ax = (float)a[0]; tt.ay = (float)a[1]; tt.az = (float)a[2];
    tt.bx = (float)b[0]; tt.by = (float)b[1]; tt.bz = (float)b[2];
    ts.element = t.element;
}

struct alignas(16) Material {
    int32_t surface; float param; int32_t albedo_img, emission_img;
    float albedo[3]; int32_t roughness_img;
    float emission[3]; float roughness;  // roughness tint .x (material.roughness.sample(uv).x, scene.rs:395)
    int32_t nee_albedo_zero; int32_t pad[3];
};

struct ImageRef { uint32_t offset, width, height, pad; };

struct Emitter { float c[3]; float r; int32_t element; int32_t pad[3]; };

struct CameraF {
    float eye[3], right[3], up[3], forward[3], phr[3], phu[3];
    float lens_radius, focus_distance;
    int32_t lens_shape;
};

// The camera once more in the reference's own precision (camera.rs:7-29 is f64): read only when a PRIMARY ray hits a sphere — the hit point
// and normal are then taken from the exact camera ray (pt_core.h camera_ray_f64 / sphere_surface), not from its fp32 rounding.
struct CameraD {
    double eye[3], right[3], up[3], forward[3], phr[3], phu[3];
    double lens_radius, focus_distance;
};

struct Scene {
    const QNode *qnodes;     // [8][num_nodes + 1], octant-major (host- and device-built trees alike); nullptr with option quant_nodes = 0
    float qmin[3], qstep[3]; // the grid of the quantised planes
    const Node *nodes;       // [8][num_nodes], octant-major
    const TriT *tris; const TriS *tri_shade;   // leaf-ordered, one pair per triangle reference
    const uint32_t *tri_face;                  // leaf-ordered: index of the input triangle each reference belongs to (hr_debug_path_log only)
    const f4 *spheres; const int32_t *sphere_elem;
    const f4 *sphere_lo;     // per sphere: what rounding its f64 centre and radius to `spheres[]` took away (c - (float)c, r - (float)r) — hit_surface only
    const f4 *cuboids;       // 2 per cuboid: {min, element-as-int-bits}, {max, 0}
    const f4 *cuboid_lo;     // [2 * ELEMENT id]: what rounding a cuboid's f64 bounds to `cuboids[]` took away — precise shading only
    const TriX *tri_exact;   // [input triangle] (tri_face[] leads there from a leaf-ordered reference) — precise shading only
    const Material *materials;
    const uint32_t *texels;
    const ImageRef *images;
    const Emitter *emitters;
    uint32_t num_nodes, num_tris, num_spheres, num_cuboids, num_elements, num_emitters;
    int32_t sky_image[6];
    float sky_intensity[3];
    // The skybox again as bilinear FOOTPRINTS (nullptr when its faces differ in size): [face][iy1 = 0 .. h][x = 0 .. w + 1] x 2 RGBA8 words,
    // the column pair {(x, iy1), (x, iy2)} as texture.rs:29-49 would fetch it (clamps and the flipped row included) — the four texels
    // of the footprint of corner (ix1, iy1) are the 16 contiguous bytes of pairs ix1 and ix1 + 1: ONE load instead of an image descriptor
    // and four scattered texels.  A path ends with a sky lookup in a random direction: 7 % of the trace kernel's time went into
    // those loads.  2 x the texels (50 MB for six 1024^2 faces).
    const uint32_t *sky_quads;
    uint32_t sky_w, sky_h;
    CameraF cam;
    const CameraD *camd;     // the camera in f64 (one small buffer in HBM)
};

// The priority governor's state, in device memory (hr_api.hip governor_kernel): the two kernels of a launch stamp their
// first start and last end here, a one-thread kernel behind every trace kernel turns the stamps of the launch that has just finished into
// the level the NEXT kernels start at, and the kernels read that level when they start — a launch is enqueued many launches before it
// runs, so what the host could put into its arguments would be decided far too early.
struct GovDev {
    unsigned long long t0[2][2], t1[2][2];   // [0 seed | 1 trace][slot = launch parity]: min of the starts / max of the ends (s_memrealtime ticks)
    unsigned long long prev_t0, prev_t1;     // the trace kernel of the launch before (what this launch's seed kernel ran beside)
    uint32_t lvl[2][2];                      // the level each of the four kernels read
    int32_t level;                           // what kernels that start now run at (0 .. 4, hr_api.hip GOV_LEVELS)
    int32_t fixed;                           // >= 0: option trace_boost pins the level
    float known[5];                          // smoothed max(seed, trace) ticks per level, 0 = not tried yet
    uint32_t decisions, moves;               // launches the governor judged / times it changed the level
    // The wave budget (round 5): how many of the trace kernel's persistent workgroups stay (blockIdx.x < budget; 0 = all).  Where the
    // trace kernel is the faster kernel of the pair by a margin, every wave it does not need is a wave that does not take issue slots
    // from the seed kernel beside it: the headline runs 3 workgroups per CU instead of 4 and the seed kernel 4 % faster.
    uint32_t budget;                         // what trace kernels that start now obey
    uint32_t bud[2];                         // the budget the trace kernel of each slot runs with, + 1 (0 = unset): fixed by its first wave, obeyed by all
    uint32_t budget_lo, budget_hi, budget_step;   // smallest budget, largest budget below "all", step (workgroups; set by the host from the CU count)
    uint32_t budget_moves;
    float thr_down, thr_up;                  // the control law's thresholds on trace time / seed time (gov_budget_next; set by the host per pipeline)
};
// the trace kernel's priority mask of a level (bits 0-3: which of every four box phases run at priority 1, bit 4: the leaf phase too)
HD uint32_t gov_trace_mask(int32_t level) { return level >= 4 ? 0x1fu : level == 3 ? 0xfu : 0u; }
// The wave budget's control law (governor_kernel; also compiled for the host: tests/test_host_layer.py drives it against the measured plant).
// B = the budget a judged launch ran with (0 = every workgroup), ratio = its trace kernel's time / its seed kernel's time at level 0.  One step
// down while the trace kernel has more than 12 % to spare, one step up when it comes within 3 % of the seed kernel, "all" above budget_hi.
// The split pipeline (wf_kernels.h) has other thresholds: its traversal workgroups cost the seed kernel beside them more than they gain their own
// kernel, and the pair is best where the two sides take the same time (measured on the headline with precise shading, workgroups 768 / 1,024 /
// 1,280 / 1,536 / all: seed 25.4 / 25.6 / 26.2 / 26.9 / 26.9 ms, trace side 26.4 / 25.2 / 24.5 / 24.5 / 24.6 ms) — down below 0.96, up above 1.02.
HD uint32_t gov_budget_next(uint32_t B, float ratio, uint32_t lo, uint32_t hi, uint32_t step, float down = 0.88f, float up = 0.97f) {
    if (!step) return B;
    if (ratio > up && B) return B + step > hi ? 0u : B + step;
    if (ratio < down) return !B ? hi : (B >= lo + step ? B - step : B);
    return B;
}
// the seed kernel's producer priorities of a level: bits 0-1 even groups, bits 2-3 odd groups
HD uint32_t gov_producer_prio(int32_t level, uint32_t init_prio) { return level <= 0 ? (init_prio | init_prio << 2) : level == 1 ? init_prio : 0u; }

// One render launch covers `num_k` samplings (sampling_begin + k*stride) of the whole image.
struct RenderParams {
    uint32_t width, height;
    uint32_t tiles_x, tiles_y;        // 4x4-pixel tiles
    uint32_t sampling_begin, stride, num_k;
    uint32_t adv_den;                 // trace kernel: leave the traversal loop when 1/adv_den of the live lanes are done
    uint32_t leaf_den;                // trace kernel: run the leaf phase when 1/leaf_den of the traversing lanes parked a leaf
    uint32_t pad[3];                  // seed kernel: [0] s_setprio of the consumer waves, [1] of the producer waves at level 0 (with gov == nullptr: the packed even | odd priorities themselves), [2] debug_skip bits
    uint32_t trace_boost;             // trace kernel, gov == nullptr only: its priority mask (gov_trace_mask)
    uint32_t node_unroll;             // trace kernel: node fetches per pass of the box-phase loop (1 or 2)
    uint32_t kchunk;                  // trace kernel: samplings per work unit (0 = 4)
    uint32_t ovf_cap;                 // seed kernels: entries of each consumer wave's fix-up list (sized per launch by hr_api.hip)
    uint32_t rr_start;                // trace kernel: Russian roulette from this iteration on (0 = off, the default: the reference has none)
    uint32_t gov_slot;                // parity of the launch: which slot of gov-> its two kernels stamp
    uint32_t nee_cull_off;            // trace kernel: which of nee_setup's two shortcuts are switched OFF (bit 0 far side, 1 GGX below the horizon; bit 2 reserved, nothing reads it); 7 = trace every NEE shadow ray (debug option nee_cull: the A/B and the bit-equality test)
    uint32_t tail_div;                // trace kernel: the last tiles / tail_div tiles of a launch are handed out one sampling at a time (0 = none): finer work units where the launch runs dry
    uint32_t wg_budget;               // trace kernel: workgroups with blockIdx.x >= wg_budget leave at once (0 = all stay) — debug option trace_budget
    GovDev *gov;                      // nullptr: no governor (debug kernels, host emulation) — trace_boost / pad[1] as given
    uint64_t rec_lo_off;              // precise shading: the records' second half, in floats from the first — slot k of a path holds what rounding draw k to fp32
                                      // took away (isaac_core.h draw_lo_f32), same layout; 0: the launch carries no residuals (fp32 draws)
};

// lane j of tile `tile` -> pixel and sub-sample (tile = 4x4 pixels x 4 sub-samples = 64 paths per sampling)
HD void tile_lane_pixel(const RenderParams &rp, uint32_t tile, uint32_t j, uint32_t &px, uint32_t &py, uint32_t &sub) {
    uint32_t tx = tile % rp.tiles_x, ty = tile / rp.tiles_x;
    uint32_t pix = j >> 2;
    sub = j & 3u;
    px = tx * 4u + (pix & 3u);
    py = ty * 4u + (pix >> 2);
}

// Hand-off from the seed kernel to the trace kernel: a 128-byte record of REC_FLOATS fp32 slots per path,
//   slots 0 .. REC_DRAWS-1   the path's first draws as the fp32 values the trace kernel computes with (draw k = k-th next_f64, rounded once)
//   slot  REC_HEAD           index a of the accepted lens attempt (uint bits), then lens x, lens y (2 u - 1, camera.rs:69-70), spare
// Precise shading (RenderParams::rec_lo_off != 0): a second record array of the same layout behind the first holds the draws' residuals —
// f64 draw = (double)slot + (double)residual slot to 2^-49 — written by the same stores' twins, read by shading two floats at a time.
// stored per item (= tile x sampling: the 64 paths of one wave-sized tile) as [quad = slot / 4][64 lanes][4 floats]: the seed kernel's
// lanes (consecutive paths) write 16 bytes each into one contiguous row per store instruction, the trace kernel's lanes read their
// two draws of an iteration with one 8-byte load and the head with one 16-byte load, both coalesced across the lanes of a tile.
// A path consumes draws 2*a, 2*a+1 (lens, taken from the head) and 2*(a+i), 2*(a+i)+1 for iteration i = 1..9, so a record covers
// a <= LENS_FAST-1 rejections; the rare path that needs more (probability (1 - pi/4)^LENS_FAST = 4.6e-4 for the round lens) is
// queued for the fix-up kernel, which re-derives it with a window of ISAAC_TAIL outputs and rewrites the record rebased to a = 0.
static const int ISAAC_TAIL = 64;     // raw-output window of the debug and fix-up kernels
static const int REC_FLOATS = 32;
static const int REC_DRAWS = 28;
static const int REC_HEAD = 28;
static const int LENS_FAST = 5;       // 2 * (LENS_FAST - 1 + 9) + 1 < REC_DRAWS
static const int DRAWS_PER_PATH = 20; // what a path can consume at most (debug API)
static const uint32_t REC_ITEM_FLOATS = 64u * REC_FLOATS;   // one item = 8 KiB
// float index of `slot` of the path whose lane base is item * REC_ITEM_FLOATS + lane * 4
// seed_seg_kernel: the init sweep of 32 blocks as three runs of SEG_NBLK blocks starting at blocks 0, SEG_B1, SEG_B2 (the last two
// overlap by one block, which both write with the same values: every run has the same length, no lane is masked)
static const int SEG_B1 = 11, SEG_B2 = 21, SEG_NBLK = 11;
HD uint32_t rec_slot(uint32_t lane_base, uint32_t slot) { return lane_base + (slot >> 2) * 256u + (slot & 3u); }

struct Counters {
    unsigned long long paths, rays, node_tests, tri_tests, sphere_tests, cuboid_tests, rng_overflow, shadow_culled;   // shadow_culled: NEE shadow rays known to add nothing before they are traced (pt_core.h nee_setup)
    // wave-level phase statistics of the trace kernel (counters build): invocations and lanes served
    unsigned long long shade_calls, shade_lanes, box_passes, box_lanes, leaf_calls, leaf_lanes, outer_iters, pad2;
    // wave-cycles (s_memtime deltas, summed over the waves) spent in: A shade, B refill, C box phase, C leaf phase
    unsigned long long phase_cycles[4];
    // seed kernel, option seed_prof: cycles of the consumer waves per phase (seed_kernels.h), [7] = groups processed
    unsigned long long seed_phase[8];
};

}  // namespace hr
