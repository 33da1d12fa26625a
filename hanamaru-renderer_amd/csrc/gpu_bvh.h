// GPU BVH build: kernels around csrc/lbvh_core.h + the hipCUB radix sort.  Included by hr_api.hip.
//   (split clipping ->) keys -> radix sort -> leaf boxes -> hierarchy (LBVH: one thread per internal node | PLOC: iterations of
//   nearest-neighbour search / merge / compaction down to a few thousand clusters, joined top-down by the host's binned SAH)
//   -> bottom-up fit with tree rotations -> finish (leaf words, primitive order) -> frame of the
//   quantised planes -> emit (both record formats, per-octant preorder) -> gather the primitives into leaf order
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "lbvh_core.h"

namespace hr {
namespace lbvh {

__global__ void key_kernel(Prims p, int n, mkey_t *keys) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = prim_key(p, (uint32_t)i);
}
// early split clipping: pieces per triangle, then (after an exclusive scan of the counts) their boxes and owners
__global__ void split_count_kernel(const Tri *tris, uint32_t num_tris, SplitParams sp, uint32_t *counts) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < num_tris) counts[i] = split_tri(tris[i], sp, nullptr);
}
__global__ void split_emit_kernel(const Tri *tris, uint32_t num_tris, SplitParams sp, const uint32_t *offsets, uint32_t *ref_tri, float *ref_box) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_tris) return;
    const uint32_t first = offsets[i];
    const uint32_t cnt = split_tri(tris[i], sp, ref_box + 6 * (size_t)first);
    for (uint32_t k = 0; k < cnt; k++) ref_tri[first + k] = i;
}
__global__ void leaf_kernel(Prims p, const mkey_t *keys, int n, Work w) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) fit_leaf(p, keys, n, k, w);
}
__global__ void hierarchy_kernel(const mkey_t *keys, int n, Work w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n - 1) hierarchy_node(keys, n, i, w);
}

// PLOC, the bottom of the tree.  One iteration = four launches on the build stream (round 2 ran all iterations in ONE workgroup:
// 3 - 5 ms at 10^4 primitives, unusable at 10^6):
//   ploc_nn_kernel     every cluster's nearest neighbour within +-PLOC_RADIUS (reads the boxes the previous iteration's merges wrote)
//   ploc_role_kernel   role per cluster -> packed counters {keeps a slot, makes a node}
//   hipcub ExclusiveSum over the packed counters: the position of every survivor and the rank of every merge, in Morton order
//   ploc_merge_kernel  merges + compaction into the other cluster array; the last cluster's thread writes the next iteration's
//                      {cluster count, next free node id} into the OTHER half of a two-slot state (no kernel reads what it writes)
// The cluster count lives on the device; the host reads it back every few iterations only to stop and to shrink the grids.
// The merges stop at <= PLOC_TOP_CLUSTERS clusters: bottom-up merges of Morton neighbours are at their worst where the boxes are
// big (the PLOC-only tree cost 1.10x the host SAH tree's node tests per ray), so the TOP of the tree is built top-down by the host
// builder's binned SAH over the clusters (bvh_build.cpp build_top_tree: a few thousand boxes, well under a millisecond):
// ploc_top_gather_kernel hands the clusters' boxes and primitive counts to the host, ploc_top_apply_kernel wires the result in.
struct PlocState { uint32_t m, next_node; };
__global__ void ploc_init_kernel(int n, uint32_t *cl, PlocState *st) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < (uint32_t)n) cl[k] = (uint32_t)(n - 1) + k;
    if (k == 0) { st[0].m = (uint32_t)n; st[0].next_node = (uint32_t)(n - 1); st[1] = st[0]; }
}
__global__ void ploc_nn_kernel(Work w, const uint32_t *cur, uint32_t *nn, const PlocState *st) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, m = st->m;
    if (i < m && m > 1u) nn[i] = ploc_nearest(w, cur, m, i);
}
__global__ void ploc_role_kernel(const uint32_t *nn, u64t *flags, uint32_t count, const PlocState *st) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, m = st->m;
    if (i >= count) return;
    u64t f = 0;
    if (i < m && m > 1u) {
        const int role = ploc_role(nn, i);
        f = (u64t)(role != 2) | ((u64t)(role == 1) << 32);
    }
    flags[i] = f;
}
__global__ void ploc_merge_kernel(Work w, const uint32_t *cur, uint32_t *nxt, const uint32_t *nn, const u64t *flags, const u64t *pos, const PlocState *st, PlocState *st_next) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, m = st->m;
    if (m <= 1u) { if (i == 0) *st_next = *st; return; }
    if (i >= m) return;
    const u64t f = flags[i], p = pos[i];
    const uint32_t slot = (uint32_t)p, mrank = (uint32_t)(p >> 32);
    if ((uint32_t)f) {
        if (f >> 32) {
            const uint32_t id = st->next_node - 1u - mrank;
            ploc_make_node(w, id, cur[i], cur[nn[i]]);
            nxt[slot] = id;
        } else nxt[slot] = cur[i];
    }
    if (i == m - 1u) { st_next->m = slot + (uint32_t)f; st_next->next_node = st->next_node - (mrank + (uint32_t)(f >> 32)); }
}
__global__ void ploc_top_gather_kernel(Work w, const uint32_t *clusters, uint32_t m, float *boxes, uint32_t *counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t c = clusters[i];
    for (int a = 0; a < 3; a++) { boxes[6 * (size_t)i + a] = w.bmin[c * 3 + a]; boxes[6 * (size_t)i + 3 + a] = w.bmax[c * 3 + a]; }
    counts[i] = w.info[c] & INFO_COUNT;
}
__global__ void ploc_top_apply_kernel(Work w, uint32_t inner, const int32_t *tl, const int32_t *tr, const uint32_t *clusters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < inner) top_apply(w, i, tl, tr, clusters);
}

// one thread per leaf walks up; the second arrival at a node fits it (its two subtrees are then complete)
__global__ void fit_kernel(int n, uint32_t max_leaf, Work w) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || n == 1) return;
    uint32_t cur = w.parent[n - 1 + k];
    while (cur != NO_PARENT) {
        __threadfence();
        if (atomicAdd(&w.flags[cur], 1u) == 0u) return;
        __threadfence();
        rotate_children(n, cur, max_leaf, w);   // both subtrees are complete and nobody else is inside them
        fit_inner(n, cur, max_leaf, w);
        cur = w.parent[cur];
    }
}
__global__ void finish_kernel(Prims p, int n, Work w, uint32_t *prim_pos) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * n - 1) finish_node(p, n, (uint32_t)i, w, prim_pos);
}
// frame[0..2] = qmin, frame[3..5] = qstep, ((uint32_t *)frame)[6] = records per octant
__global__ void frame_kernel(Work w, float *frame) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float rmn[3], rmx[3];
        for (int a = 0; a < 3; a++) { rmn[a] = pad_down(w.bmin[a]); rmx[a] = pad_up(w.bmax[a]); }
        qframe_from_box(rmn, rmx, frame, frame + 3);
        reinterpret_cast<uint32_t *>(frame)[6] = w.size[0];
    }
}
__global__ void emit_kernel(int n, Work w, const float *frame, Node *nodes, QNode *qnodes) {
    int N = 2 * n - 1;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * 8) return;
    int o = idx / N, i = idx - o * N;
    emit_node((uint32_t)i, o, w, frame, frame + 3, nodes, qnodes);
    if (i == 0) {
        const uint32_t total = w.size[0];
        qnodes[(size_t)o * (total + 1u) + total] = qnode_sentinel(o);
    }
}
// primitive arrays -> leaf order per type (left-first depth-first order of the tree)
// the triangle records the kernels read (device_scene.h), for triangles already in leaf order (host-built trees: the derivation runs
// on the device for every builder, so that the same triangle gets the same bits whoever built the tree)
__global__ void tri_derive_kernel(const Tri *in, uint32_t n, TriT *tris, TriS *tri_shade, uint32_t *tri_face) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { tri_derive(in[i], tris[i], tri_shade[i]); tri_face[i] = in[i].face; }
}
__global__ void gather_kernel(Prims p, const mkey_t *keys, const uint32_t *prim_pos, int n, TriT *tris, TriS *tri_shade, uint32_t *tri_face, f4 *spheres, int32_t *sphere_elem, const int32_t *sphere_elem_in, f4 *sphere_lo, const f4 *sphere_lo_in, f4 *cuboids) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t i = key_index(p, keys[k]), d = prim_pos[k];
    if (i < p.num_tris) { const Tri t = p.tris[p.ref_tri ? p.ref_tri[i] : i]; tri_derive(t, tris[d], tri_shade[d]); tri_face[d] = t.face; }
    else if (i < p.num_tris + p.num_spheres) {
        uint32_t l = i - p.num_tris;
        d -= p.num_tris;
        spheres[d] = p.spheres[l]; sphere_elem[d] = sphere_elem_in[l]; sphere_lo[d] = sphere_lo_in[l];
    } else {
        uint32_t l = i - p.num_tris - p.num_spheres;
        d -= p.num_tris + p.num_spheres;
        cuboids[2 * d] = p.cuboids[2 * l]; cuboids[2 * d + 1] = p.cuboids[2 * l + 1];
    }
}

}  // namespace lbvh
}  // namespace hr
