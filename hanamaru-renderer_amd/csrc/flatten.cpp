#include "flatten.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace hr {

// ---- early split clipping (Ernst & Greiner 2007): a long thin triangle that runs diagonally through space has
// an AABB hundreds of times larger than itself (the wire-frame bunny is made of such triangles).  Before the
// BVH is built its reference is split along the longest axis of its box, clipping the triangle to each half, until
// the box is no longer much larger than the piece of triangle inside it.  Leaves then hold several references to
// the same triangle; closest-hit results are unchanged (the same triangle tested twice gives the same t).
struct P3 { double x[3]; };
static void poly_box(const std::vector<P3> &poly, double *mn, double *mx) {
    for (int a = 0; a < 3; a++) { mn[a] = 1e300; mx[a] = -1e300; }
    for (const P3 &p : poly) for (int a = 0; a < 3; a++) { mn[a] = std::fmin(mn[a], p.x[a]); mx[a] = std::fmax(mx[a], p.x[a]); }
}
static double poly_area(const std::vector<P3> &poly) {
    double ax = 0, ay = 0, az = 0;
    for (size_t i = 1; i + 1 < poly.size(); i++) {
        double e1[3], e2[3];
        for (int a = 0; a < 3; a++) { e1[a] = poly[i].x[a] - poly[0].x[a]; e2[a] = poly[i + 1].x[a] - poly[0].x[a]; }
        ax += e1[1] * e2[2] - e1[2] * e2[1]; ay += e1[2] * e2[0] - e1[0] * e2[2]; az += e1[0] * e2[1] - e1[1] * e2[0];
    }
    return 0.5 * std::sqrt(ax * ax + ay * ay + az * az);
}
static void clip_poly(const std::vector<P3> &in, int axis, double pos, bool keep_low, std::vector<P3> &out) {
    out.clear();
    size_t n = in.size();
    for (size_t i = 0; i < n; i++) {
        const P3 &a = in[i], &b = in[(i + 1) % n];
        bool ia = keep_low ? a.x[axis] <= pos : a.x[axis] >= pos, ib = keep_low ? b.x[axis] <= pos : b.x[axis] >= pos;
        if (ia) out.push_back(a);
        if (ia != ib) {
            double t = (pos - a.x[axis]) / (b.x[axis] - a.x[axis]);
            P3 c;
            for (int k = 0; k < 3; k++) c.x[k] = a.x[k] + t * (b.x[k] - a.x[k]);
            c.x[axis] = pos;
            out.push_back(c);
        }
    }
}
// A reference is split at the middle of its longest box extent while its box is (a) much bigger than the part of the triangle
// inside it (surface area > ratio_max x 4 x area) AND (b) not small next to the scene (surface area > min_sa): finely tessellated
// meshes are left alone, long thin triangles are cut into up to 2^depth pieces.
static void split_refs(const std::vector<P3> &poly, uint32_t tri_index, int depth, double ratio_max, double min_sa, std::vector<BuildPrim> &out) {
    double mn[3], mx[3];
    poly_box(poly, mn, mx);
    double d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    double sa = 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    double area = poly_area(poly);
    int axis = d[0] > d[1] ? (d[0] > d[2] ? 0 : 2) : (d[1] > d[2] ? 1 : 2);
    if (depth <= 0 || poly.size() < 3 || !(sa > ratio_max * 4.0 * area) || !(sa > min_sa) || !(d[axis] > 1e-6)) {
        BuildPrim p{};
        p.type = 0; p.index = tri_index;
        for (int a = 0; a < 3; a++) { p.bmin[a] = mn[a]; p.bmax[a] = mx[a]; }
        out.push_back(p);
        return;
    }
    double pos = 0.5 * (mn[axis] + mx[axis]);
    std::vector<P3> lo, hi;
    clip_poly(poly, axis, pos, true, lo);
    clip_poly(poly, axis, pos, false, hi);
    if (lo.size() >= 3) split_refs(lo, tri_index, depth - 1, ratio_max, min_sa, out);
    if (hi.size() >= 3) split_refs(hi, tri_index, depth - 1, ratio_max, min_sa, out);
}

static int ferr(std::string &err, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
}

void HostScene::derive_triangles() {
    tri_t.resize(tris.size()); tri_s.resize(tris.size()); tri_face.resize(tris.size());
    for (size_t i = 0; i < tris.size(); i++) { tri_derive(tris[i], tri_t[i], tri_s[i]); tri_face[i] = tris[i].face; }
}

Scene HostScene::view() const {
    Scene d;
    memset(&d, 0, sizeof d);
    d.nodes = nodes.data(); d.tris = tri_t.data(); d.tri_shade = tri_s.data(); d.tri_face = tri_face.data();
    d.qnodes = qnodes.empty() ? nullptr : qnodes.data();
    for (int k = 0; k < 3; k++) { d.qmin[k] = qmin[k]; d.qstep[k] = qstep[k]; }
    d.spheres = spheres.data(); d.sphere_elem = sphere_elem.data(); d.sphere_lo = sphere_lo.data(); d.cuboids = cuboids.data();
    d.cuboid_lo = cuboid_lo.data(); d.tri_exact = tri_exact.data();
    d.materials = materials.data(); d.texels = texels.data(); d.images = images.data(); d.emitters = emitters.data();
    d.num_nodes = num_nodes; d.num_tris = (uint32_t)tris.size(); d.num_spheres = (uint32_t)spheres.size();
    d.num_cuboids = (uint32_t)(cuboids.size() / 2); d.num_elements = (uint32_t)materials.size(); d.num_emitters = (uint32_t)emitters.size();
    for (int f = 0; f < 6; f++) d.sky_image[f] = sky_image[f];
    d.sky_quads = sky_quads.empty() ? nullptr : sky_quads.data(); d.sky_w = sky_w; d.sky_h = sky_h;
    for (int k = 0; k < 3; k++) d.sky_intensity[k] = sky_intensity[k];
    d.cam = cam;
    d.camd = &camd;
    return d;
}

int flatten_scene(const hr_scene_desc *sd, HostScene &out, std::string &err, int max_leaf, double split_ratio, bool host_bvh) {
    if (!sd) return ferr(err, HR_ERR_INVALID, "null scene");
    if (!sd->elements || sd->num_elements == 0) return ferr(err, HR_ERR_INVALID, "scene has no elements");
    struct TriD { double v0[3], v1[3], v2[3]; int32_t elem; };
    std::vector<TriD> tris;
    std::vector<f4> spheres, sphere_lo; std::vector<int32_t> sphere_elem;
    std::vector<f4> cuboids, cuboid_lo;
    std::vector<BuildPrim> prims;        // one reference per primitive
    std::vector<BuildPrim> prims_split;  // triangles cut by early split clipping (when enabled), other primitives as they are
    // split_ratio < 0: build both trees and keep the split one only when it cuts the SAH cost by more than 7 % (the 8 octant
    // copies of the bigger tree have to pay for themselves: rtcamp6_v3_1 -10 % SAH, +28 % nodes, measured +2.6 % Mpaths/s;
    // rtcamp5 -2.5 % SAH for 2.5x the nodes: left alone)
    const bool auto_split = split_ratio < 0.0;
    const double ratio = auto_split ? 2.0 : split_ratio;
    out.materials.assign(sd->num_elements, Material{});
    out.emitters.clear();
    auto f3 = [](float *dst, const hr_vec3 &v) { dst[0] = (float)v.x; dst[1] = (float)v.y; dst[2] = (float)v.z; };
    // Non-finite geometry: the reference would panic in its BVH build (`partial_cmp().unwrap()` on a NaN, bvh.rs:107-211); across a C ABI
    // that is an error return, before a NaN can reach a comparator of the builders here.
    auto fin3 = [](const hr_vec3 &v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); };
    for (uint32_t ei = 0; ei < sd->num_elements; ei++) {
        const hr_element &e = sd->elements[ei];
        if (e.kind == HR_SPHERE && !(fin3(e.center) && std::isfinite(e.radius))) return ferr(err, HR_ERR_INVALID, "element %u: sphere centre / radius is not finite", ei);
        if (e.kind == HR_CUBOID && !(fin3(e.aabb_min) && fin3(e.aabb_max))) return ferr(err, HR_ERR_INVALID, "element %u: cuboid bounds are not finite", ei);
        if (e.kind == HR_MESH && e.vertexes)
            for (uint64_t v = 0; v < e.num_vertexes; v++)
                if (!fin3(e.vertexes[v])) return ferr(err, HR_ERR_INVALID, "element %u: vertex %llu is not finite", ei, (unsigned long long)v);
    }
    double split_min_sa = 0.0;   // early split clipping leaves references alone whose box is below 1e-4 of the scene's
    if (ratio > 0.0 && host_bvh) {
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
        auto grow = [&](double x, double y, double z) {
            const double p[3] = {x, y, z};
            for (int a = 0; a < 3; a++) { mn[a] = std::fmin(mn[a], p[a]); mx[a] = std::fmax(mx[a], p[a]); }
        };
        for (uint32_t ei = 0; ei < sd->num_elements; ei++) {
            const hr_element &e = sd->elements[ei];
            if (e.kind == HR_SPHERE) { grow(e.center.x - e.radius, e.center.y - e.radius, e.center.z - e.radius); grow(e.center.x + e.radius, e.center.y + e.radius, e.center.z + e.radius); }
            else if (e.kind == HR_CUBOID) { grow(e.aabb_min.x, e.aabb_min.y, e.aabb_min.z); grow(e.aabb_max.x, e.aabb_max.y, e.aabb_max.z); }
            else if (e.kind == HR_MESH && e.vertexes) for (uint64_t v = 0; v < e.num_vertexes; v++) grow(e.vertexes[v].x, e.vertexes[v].y, e.vertexes[v].z);
        }
        const double d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        if (d[0] >= 0) split_min_sa = 1e-4 * 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
    for (uint32_t ei = 0; ei < sd->num_elements; ei++) {
        const hr_element &e = sd->elements[ei];
        Material &m = out.materials[ei];
        memset(&m, 0, sizeof m);
        if (e.material.surface < HR_DIFFUSE || e.material.surface > HR_GGX_REFRACTION)
            return ferr(err, HR_ERR_INVALID, "element %u: bad surface type %d", ei, e.material.surface);
        for (const hr_texture *t : {&e.material.albedo, &e.material.emission, &e.material.roughness})
            if (t->image >= (int32_t)sd->num_images) return ferr(err, HR_ERR_INVALID, "element %u: image index %d out of range", ei, t->image);
        m.surface = e.material.surface; m.param = (float)e.material.param;
        m.albedo_img = e.material.albedo.image < 0 ? -1 : e.material.albedo.image;
        m.emission_img = e.material.emission.image < 0 ? -1 : e.material.emission.image;
        m.roughness_img = e.material.roughness.image < 0 ? -1 : e.material.roughness.image;
        f3(m.albedo, e.material.albedo.color); f3(m.emission, e.material.emission.color);
        m.roughness = (float)e.material.roughness.color.x;
        if (e.kind == HR_SPHERE) {
            BuildPrim p{};
            p.type = 1; p.index = (uint32_t)spheres.size();
            double c3[3] = {e.center.x, e.center.y, e.center.z};
            for (int a = 0; a < 3; a++) { p.bmin[a] = c3[a] - e.radius; p.bmax[a] = c3[a] + e.radius; }
            prims.push_back(p);
            prims_split.push_back(p);
            spheres.push_back(f4{(float)e.center.x, (float)e.center.y, (float)e.center.z, (float)e.radius});
            sphere_elem.push_back((int32_t)ei);
            {   // a small sphere divides an error of its centre by its radius: the f64 hit point / normal code gets the centre the reference has
                const f4 &q = spheres.back();
                sphere_lo.push_back(f4{(float)(e.center.x - (double)q.x), (float)(e.center.y - (double)q.y), (float)(e.center.z - (double)q.z), (float)(e.radius - (double)q.w)});
            }
            // Scene::emissions (scene.rs:356-358): nee_available() (spheres only, scene.rs:89) && emission tint != 0
            if (e.material.emission.color.x != 0.0 || e.material.emission.color.y != 0.0 || e.material.emission.color.z != 0.0) {
                Emitter em{};
                f3(em.c, e.center); em.r = (float)e.radius; em.element = (int32_t)ei;
                out.emitters.push_back(em);
            }
        } else if (e.kind == HR_CUBOID) {
            BuildPrim p{};
            p.type = 2; p.index = (uint32_t)(cuboids.size() / 2);
            double mn[3] = {e.aabb_min.x, e.aabb_min.y, e.aabb_min.z}, mx[3] = {e.aabb_max.x, e.aabb_max.y, e.aabb_max.z};
            for (int a = 0; a < 3; a++) { p.bmin[a] = mn[a]; p.bmax[a] = mx[a]; }
            prims.push_back(p);
            prims_split.push_back(p);
            union { int32_t i; float f; } cv;
            cv.i = (int32_t)ei;
            cuboids.push_back(f4{(float)mn[0], (float)mn[1], (float)mn[2], cv.f});
            cuboids.push_back(f4{(float)mx[0], (float)mx[1], (float)mx[2], 0.0f});
            {   // what the fp32 bounds lost (precise shading recomputes the hit on the f64 box)
                const f4 &a = cuboids[cuboids.size() - 2], &b = cuboids.back();
                cuboid_lo.resize(2 * (size_t)sd->num_elements, f4{0, 0, 0, 0});   // indexed by ELEMENT: no leaf order to follow
                cuboid_lo[2 * ei] = f4{(float)(mn[0] - (double)a.x), (float)(mn[1] - (double)a.y), (float)(mn[2] - (double)a.z), 0.0f};
                cuboid_lo[2 * ei + 1] = f4{(float)(mx[0] - (double)b.x), (float)(mx[1] - (double)b.y), (float)(mx[2] - (double)b.z), 0.0f};
            }
        } else if (e.kind == HR_MESH) {
            if (!e.vertexes || !e.faces) return ferr(err, HR_ERR_INVALID, "element %u: mesh without data", ei);
            for (uint64_t fi = 0; fi < e.num_faces; fi++) {
                uint64_t i0 = e.faces[fi * 3], i1 = e.faces[fi * 3 + 1], i2 = e.faces[fi * 3 + 2];
                if (i0 >= e.num_vertexes || i1 >= e.num_vertexes || i2 >= e.num_vertexes)
                    return ferr(err, HR_ERR_INVALID, "element %u face %llu: vertex index out of range", ei, (unsigned long long)fi);
                TriD t;
                const hr_vec3 &a = e.vertexes[i0], &b = e.vertexes[i1], &cc = e.vertexes[i2];
                t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z; t.v1[0] = b.x; t.v1[1] = b.y; t.v1[2] = b.z;
                t.v2[0] = cc.x; t.v2[1] = cc.y; t.v2[2] = cc.z; t.elem = (int32_t)ei;
                if (ratio > 0.0 && host_bvh) {
                    std::vector<P3> poly(3);
                    for (int k = 0; k < 3; k++) { poly[0].x[k] = t.v0[k]; poly[1].x[k] = t.v1[k]; poly[2].x[k] = t.v2[k]; }
                    split_refs(poly, (uint32_t)tris.size(), 6, ratio, split_min_sa, prims_split);
                }
                {
                    BuildPrim p{};
                    p.type = 0; p.index = (uint32_t)tris.size();
                    for (int k = 0; k < 3; k++) {
                        p.bmin[k] = std::fmin(std::fmin(t.v0[k], t.v1[k]), t.v2[k]);
                        p.bmax[k] = std::fmax(std::fmax(t.v0[k], t.v1[k]), t.v2[k]);
                    }
                    prims.push_back(p);
                }
                tris.push_back(t);
            }
        } else {
            return ferr(err, HR_ERR_INVALID, "element %u: unknown kind %d", ei, e.kind);
        }
    }
    // the leaf word holds a 24-bit index per type (device_scene.h); the device builders' per-node type counts give spheres and cuboids 20 bits
    if (prims.size() >= MAX_PRIMS_PER_TYPE || tris.size() >= MAX_PRIMS_PER_TYPE || spheres.size() >= (1u << 20) || cuboids.size() / 2 >= (1u << 20))
        return ferr(err, HR_ERR_UNSUPPORTED, "more than 2^24 primitives (or 2^20 spheres / cuboids)");
    out.num_input_tris = (uint32_t)tris.size();

    BuiltBvh bvh;
    if (host_bvh) {
        const bool split_ok = ratio > 0.0 && prims_split.size() < MAX_PRIMS_PER_TYPE;
        if (split_ok && !auto_split) {
            build_bvh(prims_split, max_leaf, bvh);
        } else {
            build_bvh(prims, max_leaf, bvh);
            if (split_ok && auto_split && prims_split.size() > prims.size()) {
                BuiltBvh cut;
                build_bvh(prims_split, max_leaf, cut);
                if (cut.sah_cost < 0.93 * bvh.sah_cost) bvh = std::move(cut);
            }
        }
    } else {
        // the tree is built on the device (csrc/gpu_bvh.h): primitives stay in input order, only the scene bounds are needed
        for (int a = 0; a < 3; a++) { out.scene_min[a] = 1e300; out.scene_max[a] = -1e300; }
        for (const BuildPrim &p : prims) {
            bvh.order[p.type].push_back(p.index);
            for (int a = 0; a < 3; a++) { out.scene_min[a] = std::fmin(out.scene_min[a], p.bmin[a]); out.scene_max[a] = std::fmax(out.scene_max[a], p.bmax[a]); }
        }
    }
    out.nodes = bvh.nodes; out.num_nodes = bvh.num_nodes;
    out.qnodes = bvh.qnodes;
    for (int k = 0; k < 3; k++) { out.qmin[k] = bvh.qmin[k]; out.qstep[k] = bvh.qstep[k]; }
    out.bvh_max_depth = bvh.max_depth; out.bvh_leaves = bvh.num_leaves; out.bvh_sah_cost = bvh.sah_cost;

    out.tris.assign(bvh.order[0].size(), Tri{});   // one record per reference (split triangles appear more than once)
    for (size_t i = 0; i < bvh.order[0].size(); i++) {
        const TriD &s = tris[bvh.order[0][i]];
        Tri &d = out.tris[i];
        memset(&d, 0, sizeof d);
        d.v0[0] = (float)s.v0[0]; d.v0[1] = (float)s.v0[1]; d.v0[2] = (float)s.v0[2];
        // edges formed in f64, rounded once
        d.e1x = (float)(s.v1[0] - s.v0[0]); d.e1y = (float)(s.v1[1] - s.v0[1]); d.e1z = (float)(s.v1[2] - s.v0[2]);
        d.e2x = (float)(s.v2[0] - s.v0[0]); d.e2y = (float)(s.v2[1] - s.v0[1]); d.e2z = (float)(s.v2[2] - s.v0[2]);
        d.element = s.elem;
        d.face = bvh.order[0][i];
    }
    if (host_bvh) out.derive_triangles();   // (the device builders derive the records in their gather, in leaf order)
    out.spheres.resize(spheres.size()); out.sphere_elem.resize(spheres.size()); out.sphere_lo.resize(spheres.size());
    for (size_t i = 0; i < spheres.size(); i++) { out.spheres[i] = spheres[bvh.order[1][i]]; out.sphere_elem[i] = sphere_elem[bvh.order[1][i]]; out.sphere_lo[i] = sphere_lo[bvh.order[1][i]]; }
    out.cuboids.resize(cuboids.size());
    for (size_t i = 0; i < cuboids.size() / 2; i++) { out.cuboids[2 * i] = cuboids[2 * bvh.order[2][i]]; out.cuboids[2 * i + 1] = cuboids[2 * bvh.order[2][i] + 1]; }
    out.cuboid_lo = cuboid_lo;
    // the planes of the input triangles in f64 (bvh.rs:267-268,286: edges of the f64 vertices, their cross product normalised)
    out.tri_exact.resize(tris.size());
    for (size_t i = 0; i < tris.size(); i++) {
        const TriD &s = tris[i];
        const double e1[3] = {s.v1[0] - s.v0[0], s.v1[1] - s.v0[1], s.v1[2] - s.v0[2]}, e2[3] = {s.v2[0] - s.v0[0], s.v2[1] - s.v0[1], s.v2[2] - s.v0[2]};
        const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), il = l > 0.0 ? 1.0 / l : 0.0;
        TriX &x = out.tri_exact[i];
        for (int k = 0; k < 3; k++) { x.n[k] = n[k] * il; x.v0[k] = s.v0[k]; }
    }

    out.images.assign(sd->num_images, ImageRef{});
    size_t total = 0;
    for (uint32_t i = 0; i < sd->num_images; i++) {
        if (!sd->images[i].rgba || !sd->images[i].width || !sd->images[i].height) return ferr(err, HR_ERR_INVALID, "image %u is empty", i);
        out.images[i] = ImageRef{(uint32_t)total, sd->images[i].width, sd->images[i].height, 0};
        total += (size_t)sd->images[i].width * sd->images[i].height;
    }
    if (total >= (1ull << 32)) return ferr(err, HR_ERR_UNSUPPORTED, "texture pool larger than 2^32 texels");
    out.texels.resize(total ? total : 1);
    for (uint32_t i = 0; i < sd->num_images; i++)
        memcpy(&out.texels[out.images[i].offset], sd->images[i].rgba, (size_t)out.images[i].width * out.images[i].height * 4);
    for (int f = 0; f < 6; f++) {
        if (sd->skybox.face_image[f] < 0 || sd->skybox.face_image[f] >= (int32_t)sd->num_images)
            return ferr(err, HR_ERR_INVALID, "skybox face %d has no image", f);
        out.sky_image[f] = sd->skybox.face_image[f];
    }
    f3(out.sky_intensity, sd->skybox.intensity);
    {   // the skybox as bilinear footprints (device_scene.h Scene::sky_quads): column pairs {(x, y), (x, y + 1)} for x = 0 .. w + 1, with
        // exactly the clamps and the flipped row of pt_core.h texel() — texture.rs:29-49 in u32 arithmetic
        const ImageRef im0 = out.images[out.sky_image[0]];
        bool same = true;
        for (int f = 1; f < 6; f++) same = same && out.images[out.sky_image[f]].width == im0.width && out.images[out.sky_image[f]].height == im0.height;
        out.sky_quads.clear(); out.sky_w = out.sky_h = 0;
        if (same && (uint64_t)(im0.width + 2) * (im0.height + 1) * 6 * 2 < (1ull << 31)) {
            const uint32_t w = im0.width, h = im0.height;
            out.sky_w = w; out.sky_h = h;
            out.sky_quads.resize((size_t)6 * (h + 1) * (w + 2) * 2);
            auto row_of = [&](const ImageRef &im, uint32_t y) {   // texel(): the row is flipped (u32 wrap-around and all), then clamped
                uint32_t yy = im.height - y - 1u;
                yy = yy > im.height - 1 ? im.height - 1 : yy;
                return &out.texels[im.offset + (size_t)yy * im.width];
            };
            uint32_t *dst = out.sky_quads.data();
            for (int f = 0; f < 6; f++) {
                const ImageRef im = out.images[out.sky_image[f]];
                for (uint32_t y = 0; y <= h; y++) {
                    const uint32_t *ra = row_of(im, y), *rb = row_of(im, y + 1);
                    for (uint32_t x = 0; x <= w + 1; x++, dst += 2) {   // column pairs: the footprint of corner x is pairs x and x + 1
                        const uint32_t xa = x > w - 1 ? w - 1 : x;
                        dst[0] = ra[xa]; dst[1] = rb[xa];
                    }
                }
            }
        }
    }
    const hr_camera &cam = sd->camera;
    if (!(fin3(cam.eye) && fin3(cam.right) && fin3(cam.up) && fin3(cam.forward) && fin3(cam.plane_half_right) && fin3(cam.plane_half_up) &&
          std::isfinite(cam.lens_radius) && std::isfinite(cam.focus_distance)))
        return ferr(err, HR_ERR_INVALID, "camera is not finite");
    memset(&out.cam, 0, sizeof out.cam);
    f3(out.cam.eye, cam.eye); f3(out.cam.right, cam.right); f3(out.cam.up, cam.up); f3(out.cam.forward, cam.forward);
    f3(out.cam.phr, cam.plane_half_right); f3(out.cam.phu, cam.plane_half_up);
    out.cam.lens_radius = (float)cam.lens_radius; out.cam.focus_distance = (float)cam.focus_distance; out.cam.lens_shape = cam.lens_shape;
    {
        auto d3 = [](double *dst, const hr_vec3 &v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };
        d3(out.camd.eye, cam.eye); d3(out.camd.right, cam.right); d3(out.camd.up, cam.up); d3(out.camd.forward, cam.forward);
        d3(out.camd.phr, cam.plane_half_right); d3(out.camd.phu, cam.plane_half_up);
        out.camd.lens_radius = cam.lens_radius; out.camd.focus_distance = cam.focus_distance;
    }
    return HR_OK;
}

}  // namespace hr
