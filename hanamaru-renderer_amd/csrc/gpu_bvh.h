// GPU BVH build: kernels around csrc/lbvh_core.h + the hipCUB radix sort.  Included by hr_api.hip.
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
__global__ void hierarchy_kernel(const mkey_t *keys, int n, Work w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n - 1) hierarchy_node(keys, n, i, w);
}
// one thread per leaf walks up; the second arrival at a node fits it (its two subtrees are then complete)
__global__ void fit_kernel(Prims p, const mkey_t *keys, int n, uint32_t max_leaf, Work w) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    fit_leaf(p, keys, n, k, w);
    if (n == 1) return;
    uint32_t cur = w.parent[n - 1 + k];
    while (cur != NODE_END) {
        __threadfence();
        if (atomicAdd(&w.flags[cur], 1u) == 0u) return;
        __threadfence();
        fit_inner(n, cur, max_leaf, w);
        cur = w.parent[cur];
    }
}
__global__ void emit_kernel(int n, Work w, Node *nodes) {
    int N = 2 * n - 1;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * 8) return;
    int o = idx / N, i = idx - o * N;
    nodes[idx] = emit_node(n, i, o, w);
}
// primitive arrays -> sorted (leaf) order per type
__global__ void gather_kernel(Prims p, const mkey_t *keys, int n, Tri *tris, f4 *spheres, int32_t *sphere_elem, const int32_t *sphere_elem_in, f4 *cuboids) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t i = key_index(keys[k]);
    if (i < p.num_tris) tris[k] = p.tris[i];
    else if (i < p.num_tris + p.num_spheres) {
        uint32_t l = i - p.num_tris, d = (uint32_t)k - p.num_tris;
        spheres[d] = p.spheres[l]; sphere_elem[d] = sphere_elem_in[l];
    } else {
        uint32_t l = i - p.num_tris - p.num_spheres, d = (uint32_t)k - p.num_tris - p.num_spheres;
        cuboids[2 * d] = p.cuboids[2 * l]; cuboids[2 * d + 1] = p.cuboids[2 * l + 1];
    }
}

}  // namespace lbvh
}  // namespace hr
