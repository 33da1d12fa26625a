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

}  // namespace hr
