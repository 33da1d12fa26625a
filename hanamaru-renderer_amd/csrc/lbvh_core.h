// LBVH build, per-thread bodies (BVH construction on the GPU).  Karras 2012: Morton keys of the
// primitive centroids -> radix sort -> one thread per internal node for the hierarchy -> bottom-up box fit with
// arrival counters; subtrees of <= max_leaf same-type primitives collapse into leaves; the result is emitted in the
// trace kernel's format (device_scene.h): nodes[8][2n-1] with per-octant hit / miss successors, root at index 0.
//
// The reference builds its trees on the CPU with a full sort per level (bvh.rs:107-211).  Closest-hit results do not
// depend on the tree, so this builder is interchangeable with the host SAH builder (bvh_build.cpp).  HD functions:
// csrc/gpu_bvh.h wraps them in kernels, tests/emu runs them sequentially on the host.
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

typedef unsigned long long mkey_t;
static const int KEY_INDEX_BITS = 20;   // flatten_scene admits < 2^20 primitives
static const int KEY_AXIS_BITS = 14;    // 42-bit Morton code

struct Prims {   // input order: triangles, then spheres, then cuboids
    const Tri *tris; uint32_t num_tris;
    const f4 *spheres; uint32_t num_spheres;
    const f4 *cuboids; uint32_t num_cuboids;
    float smin[3], sinv[3];  // scene bounds -> [0,1)^3
};
HD uint32_t type_offset(const Prims &p, uint32_t type) { return type == 0 ? 0u : type == 1 ? p.num_tris : p.num_tris + p.num_spheres; }

struct Work {          // 2n-1 nodes: internal [0, n-1), leaf k at n-1+k (k = position in sorted order)
    uint32_t *parent;             // per node
    uint32_t *left, *right;       // per internal node (node indices)
    uint32_t *first, *last;       // per internal node: covered sorted range
    uint32_t *flags;              // per internal node: arrival counter (zeroed)
    float *bmin, *bmax;           // per node, 3 floats each
    uint32_t *word;               // per node: leaf word (device_scene.h Node::a) or 0 = inner
    uint32_t *axis_low;           // per internal node: near/far axis | (lower child is the right one ? 4 : 0)
};

HD void prim_box(const Prims &p, uint32_t i, float *mn, float *mx, uint32_t &type) {
    if (i < p.num_tris) {
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
// key = type (2 bits) | 42-bit Morton code | input index (20 bits): unique, and the types form contiguous runs
HD mkey_t prim_key(const Prims &p, uint32_t i) {
    float mn[3], mx[3];
    uint32_t type;
    prim_box(p, i, mn, mx, type);
    mkey_t q[3];
    const float top = (float)((1 << KEY_AXIS_BITS) - 1);
    for (int a = 0; a < 3; a++) {
        float c = (0.5f * (mn[a] + mx[a]) - p.smin[a]) * p.sinv[a];
        q[a] = (mkey_t)fminf(fmaxf(c * (top + 1.0f), 0.0f), top);
    }
    mkey_t m = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    return ((mkey_t)type << 62) | (m << KEY_INDEX_BITS) | i;
}
HD uint32_t key_index(mkey_t k) { return (uint32_t)(k & ((1u << KEY_INDEX_BITS) - 1)); }

HD int clz64(mkey_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)v);
#else
    return __builtin_clzll(v);
#endif
}
HD int delta(const mkey_t *k, int n, int i, int j) { return (j < 0 || j >= n) ? -1 : clz64(k[i] ^ k[j]); }

// Karras 2012 §4, internal node i of n-1
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
    w.first[i] = (uint32_t)lo; w.last[i] = (uint32_t)hi;
    w.parent[lc] = (uint32_t)i; w.parent[rc] = (uint32_t)i;
    if (i == 0) w.parent[0] = NODE_END;
}

// leaf node of sorted position k: box + one-primitive leaf word.  The primitive arrays are re-stored per type in sorted
// order, so the rank within the type is the sorted position minus the start of the type's run.
HD void fit_leaf(const Prims &p, const mkey_t *keys, int n, int k, const Work &w) {
    float mn[3], mx[3];
    uint32_t type;
    prim_box(p, key_index(keys[k]), mn, mx, type);
    uint32_t node = (uint32_t)(n - 1 + k);
    for (int a = 0; a < 3; a++) { LBVH_ST(&w.bmin[node * 3 + a], mn[a]); LBVH_ST(&w.bmax[node * 3 + a], mx[a]); }
    LBVH_ST(&w.word[node], ((type + 1u) << 28) | (1u << 20) | ((uint32_t)k - type_offset(p, type)));
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
    uint32_t f = w.first[cur], la = w.last[cur], cnt = la - f + 1;
    uint32_t wl = LBVH_LD(&w.word[n - 1 + f]), wr = LBVH_LD(&w.word[n - 1 + la]);
    uint32_t word = 0;
    if (cnt <= max_leaf && (wl >> 28) == (wr >> 28)) word = (wl & 0xf0000000u) | (cnt << 20) | (wl & 0xfffffu);
    LBVH_ST(&w.word[cur], word);
}

HD float pad_down(float v) { return nextafterf(nextafterf(v, -INFINITY), -INFINITY); }
HD float pad_up(float v) { return nextafterf(nextafterf(v, INFINITY), INFINITY); }

// record of node i for ray-direction octant o.  Nodes below a collapsed ancestor are unreachable; their records are inert.
HD Node emit_node(int n, int i, int o, const Work &w) {
    Node nd;
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = pad_down(w.bmin[i * 3 + a]); mx[a] = pad_up(w.bmax[i * 3 + a]); }  // slack like bvh_build.cpp
    node_set_box(nd, mn, mx, o);
    uint32_t word = w.word[i];
    if (word) nd.a = word;
    else {
        uint32_t al = w.axis_low[i];
        bool neg = (o >> (al & 3u)) & 1;           // ray travels toward -axis: the higher-coordinate child is nearer
        bool near_right = neg != (bool)(al & 4u);
        nd.a = near_right ? w.right[i] : w.left[i];
    }
    // miss / leaf done: climb until this subtree is the NEAR child of an ancestor -> that ancestor's far child
    uint32_t cur = (uint32_t)i, miss = NODE_END;
    for (;;) {
        uint32_t par = w.parent[cur];
        if (par == NODE_END) break;
        uint32_t al = w.axis_low[par];
        bool neg = (o >> (al & 3u)) & 1;
        bool near_right = neg != (bool)(al & 4u);
        uint32_t nearc = near_right ? w.right[par] : w.left[par], farc = near_right ? w.left[par] : w.right[par];
        if (cur == nearc) { miss = farc; break; }
        cur = par;
    }
    nd.b = miss;
    return nd;
}

}  // namespace lbvh
}  // namespace hr
