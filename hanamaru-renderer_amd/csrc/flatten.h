// hr_scene_desc (f64, reference-native types) -> fp32 device layout on the host: primitive gathering,
// BVH build, leaf-order permutation, material / image / emitter tables.  Pure C++ (no HIP) so the same
// code feeds hipMemcpy in hr_api.hip and the host emulation in tests/emu.
#pragma once
#include <string>
#include <vector>

#include "bvh_build.h"
#include "device_scene.h"
#include "hanamaru_hip.h"

namespace hr {

struct HostScene {
    std::vector<Node> nodes;   // [8][num_nodes]
    std::vector<QNode> qnodes; // [8][num_nodes + 1] (host builder only)
    float qmin[3] = {0, 0, 0}, qstep[3] = {1, 1, 1};
    uint32_t num_nodes = 0;
    uint32_t num_input_tris = 0;   // triangles of the scene (tris.size() counts references: a split triangle appears once per piece)
    std::vector<Tri> tris;       // geometry records (leaf order with host_bvh, input order without): what the builders read
    std::vector<TriT> tri_t;     // derived from tris by derive_triangles(): what the kernels read (device_scene.h)
    std::vector<TriS> tri_s;
    std::vector<uint32_t> tri_face;   // Tri::face per record
    void derive_triangles();
    std::vector<f4> spheres;
    std::vector<int32_t> sphere_elem;
    std::vector<f4> sphere_lo;         // Scene::sphere_lo
    std::vector<f4> cuboids;
    std::vector<f4> cuboid_lo;         // Scene::cuboid_lo
    std::vector<TriX> tri_exact;       // Scene::tri_exact: per INPUT triangle (tri_face[] indexes it)
    std::vector<Material> materials;
    std::vector<ImageRef> images;
    std::vector<Emitter> emitters;
    std::vector<uint32_t> texels;
    int32_t sky_image[6];
    float sky_intensity[3];
    std::vector<uint32_t> sky_quads;   // Scene::sky_quads (empty when the faces differ in size)
    uint32_t sky_w = 0, sky_h = 0;
    CameraF cam;
    CameraD camd;
    uint32_t bvh_max_depth = 0, bvh_leaves = 0;
    double bvh_sah_cost = 0;
    double scene_min[3] = {0, 0, 0}, scene_max[3] = {0, 0, 0};   // filled when host_bvh == false
    // a Scene whose pointers refer to the vectors above (valid on the host only)
    Scene view() const;
};

// returns HR_OK or a negative hr_status; `err` receives the message
// split_ratio > 0 enables early split clipping of triangle references whose box surface area exceeds
// split_ratio x 4 x (triangle area inside the box) and 1e-4 of the scene's; < 0 = automatic (ratio 1.5, kept only when the
// SAH cost drops by more than 30 %); 0 = off
// host_bvh = false: no tree is built, primitives keep their input order (the device builds the tree, csrc/gpu_bvh.h)
int flatten_scene(const hr_scene_desc *sd, HostScene &out, std::string &err, int max_leaf = 4, double split_ratio = -1.0, bool host_bvh = true);

}  // namespace hr
