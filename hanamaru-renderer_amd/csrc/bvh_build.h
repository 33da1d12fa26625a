// Host-side BVH builder for the device traversal (binned SAH, type-homogeneous leaves, threaded links
// for the 8 ray-direction octants).  The reference builds a median-split BVH per mesh plus a top-level
// BVH (bvh.rs:107-211); closest-hit results do not depend on the tree, so the device uses ONE tree over
// all triangles / spheres / cuboids of the scene, built for traversal cost instead.
#pragma once
#include <cstdint>
#include <vector>

#include "device_scene.h"

namespace hr {

struct BuildPrim {
    double bmin[3], bmax[3];
    int type;        // 0 tri, 1 sphere, 2 cuboid
    uint32_t index;  // index into the caller's per-type array
};

struct BuiltBvh {
    std::vector<Node> nodes;          // 8 * num_nodes (+ 1 spare), octant-major, each octant in its own near-first preorder: box + (hit | leaf word, miss)
    uint32_t num_nodes = 0;
    std::vector<QNode> qnodes;        // 8 * (num_nodes + 1): the same trees as 16-byte records (device_scene.h)
    float qmin[3] = {0, 0, 0}, qstep[3] = {1, 1, 1};
    std::vector<uint32_t> order[3];   // per type: leaf-ordered -> caller index
    uint32_t max_depth = 0, num_leaves = 0;
    double sah_cost = 0;              // sum over nodes of area / root area (one box test each) + 1.5 x per leaf primitive
};

void build_bvh(const std::vector<BuildPrim> &prims, int max_leaf, BuiltBvh &out);

// The TOP of a device-built tree (gpu_bvh.h, option bvh_builder = 2): the agglomerative PLOC build stops at <= a few thousand clusters
// — bottom-up merges of Morton neighbours are at their worst where the boxes are big — and this top-down binned-SAH build (the rule of
// build_bvh, weighted by the clusters' primitive counts) joins them.  boxes: 6 floats per cluster (min, max); out_left / out_right
// get m - 1 entries, inner node i's children: a value v >= 0 is inner node v, a value < 0 is cluster ~v.  Inner node 0 is the root.
void build_top_tree(const float *boxes, const uint32_t *counts, uint32_t m, std::vector<int32_t> &out_left, std::vector<int32_t> &out_right);

}  // namespace hr
