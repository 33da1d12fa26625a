// BVH build on the GPU, per-thread bodies.  Two ways to get the hierarchy over the Morton-sorted
// primitives, one way to finish it:
//   builder 1  LBVH   (Karras 2012): one thread per internal node derives its children from the sorted keys alone
//   builder 2  PLOC   (Meister & Bittner 2018): parallel locally-ordered clustering — every cluster looks for its nearest
//              neighbour (smallest surface area of the union) within +-PLOC_RADIUS positions of the Morton order, mutual pairs
//              merge, the array is compacted, until one cluster is left.  An agglomerative build: near-SAH quality.
//   finish     bottom-up fit (boxes, primitive counts per type, collapse of <= max_leaf same-type primitives into leaves, emitted
//              sizes, near / far axis), then every node finds its own place by walking up to the root: the per-type rank of a
//              leaf's first primitive (primitives end up in left-first depth-first order), and — per ray-direction octant — its
//              index in that octant's near-first preorder.  The tree is emitted in both of the trace kernel's formats
//              (device_scene.h): 32-byte fp32 records with explicit links and 16-byte quantised records with implicit ones.
//
// The reference builds its trees on the CPU with a full sort per level (bvh.rs:107-211).  Closest-hit results do not depend on
// the tree, so these builders are interchangeable with the host SAH builder (bvh_build.cpp).  HD functions: csrc/gpu_bvh.h wraps
// them in kernels, tests/emu runs them sequentially on the host.
#pragma once
#include <math.h>

#include "device_scene.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define LBVH_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LBVH_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define LBVH_LD(p) (*(p))
#define LBVH_ST(p, v) (*(p) = (v))
#endif

namespace hr {
namespace lbvh {

static const uint32_t NO_PARENT = 0xffffffffu;   // parent[] of the root (the arrays are preset with bytes of 0xff)

typedef unsigned long long mkey_t;
typedef unsigned long long u64t;
// sort key = type (2 bits) | Morton code | input index: up to 2^20 primitives 20 index bits and a 42-bit code (14 bits per axis), beyond
// that (up to 2^24) 24 index bits and a 36-bit code (12 per axis) — Prims::index_bits says which
static const int KEY_INDEX_BITS_SMALL = 20, KEY_INDEX_BITS_LARGE = 24;
HD int key_index_bits_for(uint64_t primitives) { return primitives <= (1ull << KEY_INDEX_BITS_SMALL) ? KEY_INDEX_BITS_SMALL : KEY_INDEX_BITS_LARGE; }
HD int key_axis_bits(int index_bits) { return (62 - index_bits) / 3; }
static const int PLOC_RADIUS = 16;      // nearest-neighbour search window: +-16 positions (8: 1 % more node tests per ray, 0.4 ms less build time at 15 k primitives)

struct Prims {   // input order: triangles (or their split references), then spheres, then cuboids
    const Tri *tris; uint32_t num_tris;   // num_tris = triangle-type primitives the builder sees (= references when ref_tri is set)
    const uint32_t *ref_tri;              // early split clipping: reference i is (part of) triangle ref_tri[i] with the box ref_box[6 i ..]; nullptr = none
    const float *ref_box;
    const f4 *spheres; uint32_t num_spheres;
    const f4 *cuboids; uint32_t num_cuboids;
    float smin[3], sinv[3];  // scene bounds -> [0,1)^3
    int index_bits;          // key_index_bits_for(all primitives the builder sees)
};
HD uint32_t type_offset(const Prims &p, uint32_t type) { return type == 0 ? 0u : type == 1 ? p.num_tris : p.num_tris + p.num_spheres; }

// 2n-1 nodes: internal [0, n-1) with the root at 0, leaf k at n-1+k (k = position in sorted order)
struct Work {
    uint32_t *parent;             // per node
    uint32_t *left, *right;       // per internal node (node indices)
    uint32_t *flags;              // per internal node: arrival counter of the bottom-up fit (zeroed)
    float *bmin, *bmax;           // per node, 3 floats each
    uint32_t *info;               // per node: INFO_* bits | primitives in the subtree
    u64t *tc;                     // per node: primitives per type in the subtree: triangles bits 0-23, spheres 24-43, cuboids 44-63
    uint32_t *size;               // per node: records the subtree is emitted as (1 for a collapsed subtree)
    uint32_t *axis_low;           // per internal node: near/far axis | (lower child is the right one ? 4 : 0)
    uint32_t *word;               // per node: the leaf word (device_scene.h) of a leaf top, set by finish_node; else untouched
};
static const uint32_t INFO_COLLAPSED = 1u << 31;   // <= max_leaf primitives of one type: the subtree is (part of) a leaf
static const uint32_t INFO_UNIFORM = 1u << 30;     // all primitives of one type
static const uint32_t INFO_COUNT = (1u << 24) - 1u;
HD uint32_t info_type(uint32_t info) { return (info >> 28) & 3u; }
// (triangle references < 2^24, spheres and cuboids < 2^20 each: flatten_scene's limits)
HD uint32_t tc_shift(uint32_t t) { return t == 0u ? 0u : t == 1u ? 24u : 44u; }
HD uint32_t tc_of(u64t tc, uint32_t t) { return (uint32_t)(tc >> tc_shift(t)) & (t == 0u ? 0xffffffu : 0xfffffu); }

HD void prim_box(const Prims &p, uint32_t i, float *mn, float *mx, uint32_t &type) {
    if (i < p.num_tris && p.ref_box) {
        type = 0;
        for (int a = 0; a < 3; a++) { mn[a] = p.ref_box[6 * (size_t)i + a]; mx[a] = p.ref_box[6 * (size_t)i + 3 + a]; }
    } else if (i < p.num_tris) {
        type = 0;
        const Tri t = p.tris[i];
        float v1[3] = {t.v0[0] + t.e1x, t.v0[1] + t.e1y, t.v0[2] + t.e1z}, v2[3] = {t.v0[0] + t.e2x, t.v0[1] + t.e2y, t.v0[2] + t.e2z};
        for (int a = 0; a < 3; a++) { mn[a] = fminf(fminf(t.v0[a], v1[a]), v2[a]); mx[a] = fmaxf(fmaxf(t.v0[a], v1[a]), v2[a]); }
    } else if (i < p.num_tris + p.num_spheres) {
        type = 1;
        const f4 s = p.spheres[i - p.num_tris];
        mn[0] = s.x - s.w; mn[1] = s.y - s.w; mn[2] = s.z - s.w; mx[0] = s.x + s.w; mx[1] = s.y + s.w; mx[2] = s.z + s.w;
    } else {
        type = 2;
        uint32_t l = i - p.num_tris - p.num_spheres;
        const f4 a = p.cuboids[2 * l], b = p.cuboids[2 * l + 1];
        mn[0] = a.x; mn[1] = a.y; mn[2] = a.z; mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
    }
}
HD mkey_t spread3(mkey_t x) {  // 21 bits -> every third bit
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
// key = type (2 bits) | Morton code | input index: unique, and the types form contiguous runs
HD mkey_t prim_key(const Prims &p, uint32_t i) {
    float mn[3], mx[3];
    uint32_t type;
    prim_box(p, i, mn, mx, type);
    mkey_t q[3];
    const int axis_bits = key_axis_bits(p.index_bits);
    const float top = (float)((1 << axis_bits) - 1);
    for (int a = 0; a < 3; a++) {
        float c = (0.5f * (mn[a] + mx[a]) - p.smin[a]) * p.sinv[a];
        q[a] = (mkey_t)fminf(fmaxf(c * (top + 1.0f), 0.0f), top);
    }
    mkey_t m = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    return ((mkey_t)type << 62) | (m << p.index_bits) | i;
}
HD uint32_t key_index(const Prims &p, mkey_t k) { return (uint32_t)(k & ((1ull << p.index_bits) - 1ull)); }

HD int clz64(mkey_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)v);
#else
    return __builtin_clzll(v);
#endif
}
HD int delta(const mkey_t *k, int n, int i, int j) { return (j < 0 || j >= n) ? -1 : clz64(k[i] ^ k[j]); }

// ---- builder 1: Karras 2012 §4, internal node i of n-1
HD void hierarchy_node(const mkey_t *keys, int n, int i, const Work &w) {
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    int gamma = i + s * d + (d < 0 ? -1 : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    uint32_t lc = (lo == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
    uint32_t rc = (hi == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    w.left[i] = lc; w.right[i] = rc;
    w.parent[lc] = (uint32_t)i; w.parent[rc] = (uint32_t)i;
    if (i == 0) w.parent[0] = NO_PARENT;
}

// ---- early split clipping on the device (Ernst & Greiner 2007; the host builder's form is flatten.cpp split_refs): a long thin
// triangle whose box is much bigger than the part of the triangle inside it is cut at the middle of the box's longest extent, the
// triangle clipped to each half, until the boxes fit (or `depth` cuts).  The builders then see one primitive per piece; a leaf holds
// the whole triangle's record for each of its pieces (closest hits do not change: the same triangle tested twice gives the same t).
// The recursion is an explicit stack of polygons; arithmetic in f64 on the fp32 triangle the kernel tests (v0, v0 + e1, v0 + e2),
// boxes rounded outward to fp32.  Two passes over the same code: out == nullptr counts the pieces, else writes their boxes.
struct SplitParams { double ratio_max, min_sa; int depth; };
static const int SPLIT_MAX_VERTS = 12, SPLIT_MAX_DEPTH = 6;
struct SplitPoly { double v[SPLIT_MAX_VERTS][3]; int n, depth; };
HD void split_clip(const SplitPoly &in, int axis, double pos, bool keep_low, SplitPoly &out) {
    out.n = 0;
    for (int i = 0; i < in.n; i++) {
        const double *a = in.v[i], *b = in.v[(i + 1) % in.n];
        const bool ia = keep_low ? a[axis] <= pos : a[axis] >= pos, ib = keep_low ? b[axis] <= pos : b[axis] >= pos;
        if (ia && out.n < SPLIT_MAX_VERTS) { for (int k = 0; k < 3; k++) out.v[out.n][k] = a[k]; out.n++; }
        if (ia != ib && out.n < SPLIT_MAX_VERTS) {
            const double t = (pos - a[axis]) / (b[axis] - a[axis]);
            for (int k = 0; k < 3; k++) out.v[out.n][k] = a[k] + t * (b[k] - a[k]);
            out.v[out.n][axis] = pos;
            out.n++;
        }
    }
}
HD float f32_below(double v) { float f = (float)v; return (double)f > v ? nextafterf(f, -INFINITY) : f; }
HD float f32_above(double v) { float f = (float)v; return (double)f < v ? nextafterf(f, INFINITY) : f; }
HD uint32_t split_tri(const Tri &t, const SplitParams &sp, float *out) {
    SplitPoly stack[SPLIT_MAX_DEPTH + 2];
    int top = 0;
    SplitPoly &r = stack[0];
    r.n = 3; r.depth = sp.depth < SPLIT_MAX_DEPTH ? sp.depth : SPLIT_MAX_DEPTH;
    for (int k = 0; k < 3; k++) r.v[0][k] = (double)t.v0[k];
    r.v[1][0] = (double)t.v0[0] + (double)t.e1x; r.v[1][1] = (double)t.v0[1] + (double)t.e1y; r.v[1][2] = (double)t.v0[2] + (double)t.e1z;
    r.v[2][0] = (double)t.v0[0] + (double)t.e2x; r.v[2][1] = (double)t.v0[1] + (double)t.e2y; r.v[2][2] = (double)t.v0[2] + (double)t.e2z;
    top = 1;
    uint32_t count = 0;
    while (top > 0) {
        const SplitPoly cur = stack[--top];
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
        for (int i = 0; i < cur.n; i++) for (int a = 0; a < 3; a++) { mn[a] = fmin(mn[a], cur.v[i][a]); mx[a] = fmax(mx[a], cur.v[i][a]); }
        const double d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        const double sa = 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
        double ax = 0, ay = 0, az = 0;
        for (int i = 1; i + 1 < cur.n; i++) {
            double e1[3], e2[3];
            for (int a = 0; a < 3; a++) { e1[a] = cur.v[i][a] - cur.v[0][a]; e2[a] = cur.v[i + 1][a] - cur.v[0][a]; }
            ax += e1[1] * e2[2] - e1[2] * e2[1]; ay += e1[2] * e2[0] - e1[0] * e2[2]; az += e1[0] * e2[1] - e1[1] * e2[0];
        }
        const double area = 0.5 * sqrt(ax * ax + ay * ay + az * az);
        const int axis = d[0] > d[1] ? (d[0] > d[2] ? 0 : 2) : (d[1] > d[2] ? 1 : 2);
        if (cur.depth <= 0 || cur.n < 3 || !(sa > sp.ratio_max * 4.0 * area) || !(sa > sp.min_sa) || !(d[axis] > 1e-6)) {
            if (out) for (int a = 0; a < 3; a++) { out[6 * (size_t)count + a] = f32_below(mn[a]); out[6 * (size_t)count + 3 + a] = f32_above(mx[a]); }
            count++;
            continue;
        }
        const double pos = 0.5 * (mn[axis] + mx[axis]);
        SplitPoly lo, hi;
        split_clip(cur, axis, pos, true, lo);
        split_clip(cur, axis, pos, false, hi);
        lo.depth = hi.depth = cur.depth - 1;
        if (hi.n >= 3) stack[top++] = hi;     // the low part is taken up first (the order of the host's recursion)
        if (lo.n >= 3) stack[top++] = lo;
    }
    return count;
}

// ---- leaf node of sorted position k: box, type, counts
HD void fit_leaf(const Prims &p, const mkey_t *keys, int n, int k, const Work &w) {
    float mn[3], mx[3];
    uint32_t type;
    prim_box(p, key_index(p, keys[k]), mn, mx, type);
    uint32_t node = (uint32_t)(n - 1 + k);
    for (int a = 0; a < 3; a++) { LBVH_ST(&w.bmin[node * 3 + a], mn[a]); LBVH_ST(&w.bmax[node * 3 + a], mx[a]); }
    LBVH_ST(&w.info[node], INFO_COLLAPSED | INFO_UNIFORM | (type << 28) | 1u);
    LBVH_ST(&w.tc[node], (u64t)1 << tc_shift(type));
    LBVH_ST(&w.size[node], 1u);
}
// internal node whose two subtrees are complete
HD void fit_inner(int n, uint32_t cur, uint32_t max_leaf, const Work &w) {
    uint32_t l = w.left[cur], r = w.right[cur];
    float cl[3], cr[3];
    for (int a = 0; a < 3; a++) {
        float lmn = LBVH_LD(&w.bmin[l * 3 + a]), lmx = LBVH_LD(&w.bmax[l * 3 + a]);
        float rmn = LBVH_LD(&w.bmin[r * 3 + a]), rmx = LBVH_LD(&w.bmax[r * 3 + a]);
        LBVH_ST(&w.bmin[cur * 3 + a], fminf(lmn, rmn));
        LBVH_ST(&w.bmax[cur * 3 + a], fmaxf(lmx, rmx));
        cl[a] = lmn + lmx; cr[a] = rmn + rmx;
    }
    // near/far order: the axis on which the children's centres differ most
    int axis = 0;
    float best = fabsf(cl[0] - cr[0]);
    for (int a = 1; a < 3; a++) if (fabsf(cl[a] - cr[a]) > best) { best = fabsf(cl[a] - cr[a]); axis = a; }
    w.axis_low[cur] = (uint32_t)axis | (cr[axis] < cl[axis] ? 4u : 0u);
    // collapse: a subtree of <= max_leaf primitives of ONE type becomes a leaf
    const uint32_t il = LBVH_LD(&w.info[l]), ir = LBVH_LD(&w.info[r]);
    const uint32_t cnt = (il & INFO_COUNT) + (ir & INFO_COUNT);
    const bool uniform = (il & INFO_UNIFORM) && (ir & INFO_UNIFORM) && info_type(il) == info_type(ir);
    // (a leaf wherever <= max_leaf primitives of one type meet; pricing the collapse like the host builder's leaf rule — one leaf of n
    // primitives against a node visit plus two child leaves — was measured on both device builders: fewer triangle tests, more node
    // tests, 0 to -1.3 % in Mpaths/s)
    const bool collapsed = uniform && cnt <= max_leaf;
    LBVH_ST(&w.info[cur], (collapsed ? INFO_COLLAPSED : 0u) | (uniform ? INFO_UNIFORM : 0u) | (info_type(il) << 28) | cnt);
    LBVH_ST(&w.tc[cur], LBVH_LD(&w.tc[l]) + LBVH_LD(&w.tc[r]));
    LBVH_ST(&w.size[cur], collapsed ? 1u : 1u + LBVH_LD(&w.size[l]) + LBVH_LD(&w.size[r]));
}

// ---- tree rotations (Kensler 2008) during the bottom-up fit: before an inner node is fitted — its two subtrees are complete and no
// other thread touches them — one of its children may change places with a grandchild on the other side when that shrinks the box
// of the inner node the grandchild leaves behind (the only box of the tree that changes).  Bottom-up builders (Morton splits,
// merges of Morton neighbours) leave such local mistakes; the host builder's top-down SAH splits do not.
HD float union_area3(const Work &w, uint32_t a, uint32_t b) {
    float d[3];
    for (int k = 0; k < 3; k++) d[k] = fmaxf(LBVH_LD(&w.bmax[a * 3 + k]), LBVH_LD(&w.bmax[b * 3 + k])) - fminf(LBVH_LD(&w.bmin[a * 3 + k]), LBVH_LD(&w.bmin[b * 3 + k]));
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
}
HD float node_half_area(const Work &w, uint32_t a) {
    float d[3];
    for (int k = 0; k < 3; k++) d[k] = LBVH_LD(&w.bmax[a * 3 + k]) - LBVH_LD(&w.bmin[a * 3 + k]);
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
}
HD void fit_inner(int n, uint32_t cur, uint32_t max_leaf, const Work &w);
HD void rotate_children(int n, uint32_t cur, uint32_t max_leaf, const Work &w) {
    const uint32_t c[2] = {w.left[cur], w.right[cur]};
    float best = 0.0f;
    int best_side = -1, best_grand = 0;
    for (int side = 0; side < 2; side++) {
        const uint32_t x = c[side], other = c[1 - side];           // a grandchild below x may change places with `other`
        if (x >= (uint32_t)(n - 1) || (LBVH_LD(&w.info[x]) & INFO_COLLAPSED)) continue;   // x is a leaf of the emitted tree
        const uint32_t g[2] = {w.left[x], w.right[x]};
        const float ax = node_half_area(w, x);
        for (int k = 0; k < 2; k++) {
            const float gain = ax - union_area3(w, other, g[1 - k]);   // x would keep g[1 - k] and get `other`
            if (gain > best) { best = gain; best_side = side; best_grand = k; }
        }
    }
    if (best_side < 0) return;
    const uint32_t x = c[best_side], other = c[1 - best_side];
    const uint32_t g = best_grand ? w.right[x] : w.left[x];
    if (best_grand) w.right[x] = other; else w.left[x] = other;
    if (best_side == 0) w.right[cur] = g; else w.left[cur] = g;
    w.parent[other] = x; w.parent[g] = cur;
    fit_inner(n, x, max_leaf, w);
}

// ---- builder 2: PLOC.  `cl` = the current clusters (node ids) in Morton order, m of them
HD float union_area(const Work &w, uint32_t a, uint32_t b) {
    float d[3];
    for (int k = 0; k < 3; k++) d[k] = fmaxf(w.bmax[a * 3 + k], w.bmax[b * 3 + k]) - fminf(w.bmin[a * 3 + k], w.bmin[b * 3 + k]);
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
}
HD uint32_t ploc_nearest(const Work &w, const uint32_t *cl, uint32_t m, uint32_t i) {
    uint32_t lo = i > (uint32_t)PLOC_RADIUS ? i - PLOC_RADIUS : 0u, hi = i + PLOC_RADIUS < m - 1u ? i + PLOC_RADIUS : m - 1u;
    uint32_t best = i == lo ? i + 1u : lo;
    float ba = 3.0e38f;
    for (uint32_t j = lo; j <= hi; j++) {
        if (j == i) continue;
        float a = union_area(w, cl[i], cl[j]);
        if (a < ba) { ba = a; best = j; }   // ties: the lower position — both partners of a pair then agree
    }
    return best;
}
// 1 = cluster i merges with nn[i] and the new node takes its place, 2 = it is the absorbed partner, 0 = it stays
HD int ploc_role(const uint32_t *nn, uint32_t i) {
    uint32_t j = nn[i];
    if (nn[j] != i) return 0;
    return i < j ? 1 : 2;
}
static const uint32_t PLOC_TOP_CLUSTERS = 8192;   // the merges stop at <= this many clusters; bvh_build.cpp's build_top_tree joins them top-down
HD void ploc_make_node(const Work &w, uint32_t id, uint32_t l, uint32_t r) {
    w.left[id] = l; w.right[id] = r;
    w.parent[l] = id; w.parent[r] = id;
    w.info[id] = (w.info[l] & INFO_COUNT) + (w.info[r] & INFO_COUNT);   // primitives in the cluster, for the top-down build (the fit rewrites info)
    for (int a = 0; a < 3; a++) {
        w.bmin[id * 3 + a] = fminf(w.bmin[l * 3 + a], w.bmin[r * 3 + a]);
        w.bmax[id * 3 + a] = fmaxf(w.bmax[l * 3 + a], w.bmax[r * 3 + a]);
    }
}

// SAH cost of the emitted tree as the host builder prices it (bvh_build.cpp): every emitted node pays its box area, a leaf 1.5 more
// per primitive; relative to the root's area.  One node's share (0 for nodes inside a collapsed subtree).
HD bool is_collapsed(const Work &w, uint32_t node);
HD bool is_emitted(const Work &w, uint32_t node);
HD bool is_leaf_top(const Work &w, uint32_t node);
HD float node_area(const Work &w, uint32_t node) {
    const float dx = w.bmax[node * 3] - w.bmin[node * 3], dy = w.bmax[node * 3 + 1] - w.bmin[node * 3 + 1], dz = w.bmax[node * 3 + 2] - w.bmin[node * 3 + 2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}
HD float sah_share(const Work &w, uint32_t node) {
    if (!is_emitted(w, node)) return 0.0f;
    return node_area(w, node) * (1.0f + (is_leaf_top(w, node) ? 1.5f * (float)(w.info[node] & INFO_COUNT) : 0.0f));
}

// the top tree (bvh_build.h build_top_tree) joins the clusters the merges left: its inner node i is internal node i of the hierarchy
// (the merges handed their ids out downwards and left exactly 0 .. m - 2), a child value < 0 is cluster ~v
HD void top_apply(const Work &w, uint32_t i, const int32_t *tl, const int32_t *tr, const uint32_t *clusters) {
    const uint32_t l = tl[i] >= 0 ? (uint32_t)tl[i] : clusters[~tl[i]], r = tr[i] >= 0 ? (uint32_t)tr[i] : clusters[~tr[i]];
    w.left[i] = l; w.right[i] = r;
    w.parent[l] = i; w.parent[r] = i;
    if (i == 0) w.parent[0] = NO_PARENT;
}

// ---- finish: every node finds its own place by walking up
HD bool is_collapsed(const Work &w, uint32_t node) { return (w.info[node] & INFO_COLLAPSED) != 0u; }
// a leaf of the emitted tree: collapsed, and not inside a bigger collapsed subtree
HD bool is_leaf_top(const Work &w, uint32_t node) {
    if (!is_collapsed(w, node)) return false;
    uint32_t par = w.parent[node];
    return par == NO_PARENT || !is_collapsed(w, par);
}
HD bool is_emitted(const Work &w, uint32_t node) { return !is_collapsed(w, node) || is_leaf_top(w, node); }
// primitives of type t that precede the subtree of `node` in left-first depth-first order
HD uint32_t type_rank(const Work &w, uint32_t node, uint32_t t) {
    uint32_t r = 0, c = node;
    for (uint32_t par = w.parent[c]; par != NO_PARENT; c = par, par = w.parent[c])
        if (c == w.right[par]) r += tc_of(w.tc[w.left[par]], t);
    return r;
}
HD bool near_is_right(const Work &w, uint32_t inner, int o) {
    uint32_t al = w.axis_low[inner];
    bool neg = (o >> (al & 3u)) & 1;           // ray travels toward -axis: the higher-coordinate child is nearer
    return neg != (bool)(al & 4u);
}
// index of an emitted node in octant o's near-first preorder
HD uint32_t preorder_index(const Work &w, uint32_t node, int o) {
    uint32_t idx = 0, c = node;
    for (uint32_t par = w.parent[c]; par != NO_PARENT; c = par, par = w.parent[c]) {
        uint32_t nearc = near_is_right(w, par, o) ? w.right[par] : w.left[par];
        idx += (c == nearc) ? 1u : 1u + w.size[nearc];
    }
    return idx;
}
// per node: leaf word of a leaf top; per primitive leaf (node >= n-1): where its primitive goes in the per-type arrays
HD void finish_node(const Prims &p, int n, uint32_t node, const Work &w, uint32_t *prim_pos) {
    const uint32_t info = w.info[node], t = info_type(info);
    if (is_leaf_top(w, node)) w.word[node] = leaf_word(t, info & INFO_COUNT, type_rank(w, node, t));
    if (node >= (uint32_t)(n - 1)) prim_pos[node - (uint32_t)(n - 1)] = type_offset(p, t) + type_rank(w, node, t);
}

HD float pad_down(float v) { return nextafterf(nextafterf(v, -INFINITY), -INFINITY); }
HD float pad_up(float v) { return nextafterf(nextafterf(v, INFINITY), INFINITY); }

// records of node `node` for ray-direction octant o, written at its preorder index; total = w.size[0] records per octant
HD void emit_node(uint32_t node, int o, const Work &w, const float *qmin, const float *qstep, Node *nodes, QNode *qnodes) {
    if (!is_emitted(w, node)) return;
    const uint32_t total = w.size[0], idx = preorder_index(w, node, o);
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = pad_down(w.bmin[node * 3 + a]); mx[a] = pad_up(w.bmax[node * 3 + a]); }  // slack like bvh_build.cpp
    const bool leaf = is_leaf_top(w, node);
    uint32_t after = idx + w.size[node];
    if (after == total) after = NODE_END;
    Node nd;
    node_set_box(nd, mn, mx, o);
    nd.a = leaf ? w.word[node] : idx + 1u;
    nd.b = after;
    nodes[(size_t)o * total + idx] = nd;
    qnodes[(size_t)o * (total + 1u) + idx] = qnode_make(mn, mx, o, qmin, qstep, leaf ? w.word[node] : qnode_link(o, total + 1u, after));
}

}  // namespace lbvh
}  // namespace hr
