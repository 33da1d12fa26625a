// Scene authoring on the host: OBJ loader (loader.rs:12-59), Camera::new (camera.rs:45-64),
// hsv_to_rgb (color.rs:50-61) and the scene builders that BASELINE.json's configs need
// (main.rs:1020-1153 is the live one).  Output is an hr_scene_desc for hr_upload_scene().
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <vector>

#include "hanamaru_host.h"
#include "hh_isaac64.h"
#include "hh_math.h"

using namespace hh;

struct hh_scene {
    std::vector<hr_element> elements;
    std::deque<std::vector<hr_vec3>> vertex_store;
    std::deque<std::vector<uint64_t>> face_store;
    std::deque<std::vector<uint8_t>> image_store;
    std::vector<hr_image> images;
    hr_scene_desc desc{};
    std::string root;
};

namespace {

// --- loader.rs:21-56: split on single ' ', "v x y z" -> matrix * v, "f a[/..] b c [d]" 1-based, quad iff 5 tokens
static bool load_obj(const std::string &path, const M44 &m, std::vector<hr_vec3> &verts, std::vector<uint64_t> &faces) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { set_error("cannot open %s", path.c_str()); return false; }
    std::string line;
    std::vector<std::string> tok;
    char buf[1 << 16];
    auto first_index = [](const std::string &t) -> long long {
        return atoll(t.substr(0, t.find('/')).c_str()) - 1;
    };
    while (fgets(buf, sizeof buf, f)) {
        line = buf;
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        tok.clear();
        size_t s = 0;
        for (;;) {
            size_t e = line.find(' ', s);
            tok.push_back(line.substr(s, e == std::string::npos ? std::string::npos : e - s));
            if (e == std::string::npos) break;
            s = e + 1;
        }
        if (tok[0] == "v") {
            if (tok.size() < 4) { set_error("%s: malformed v line", path.c_str()); fclose(f); return false; }
            V3 local(strtod(tok[1].c_str(), nullptr), strtod(tok[2].c_str(), nullptr), strtod(tok[3].c_str(), nullptr));
            verts.push_back((m * local).c());
        } else if (tok[0] == "f") {
            if (tok.size() < 4) { set_error("%s: malformed f line", path.c_str()); fclose(f); return false; }
            long long a = first_index(tok[1]), b = first_index(tok[2]), c = first_index(tok[3]);
            faces.push_back((uint64_t)a); faces.push_back((uint64_t)b); faces.push_back((uint64_t)c);
            if (tok.size() == 5) {  // quad -> (0,1,2) + (0,2,3)
                long long d = first_index(tok[4]);
                faces.push_back((uint64_t)a); faces.push_back((uint64_t)c); faces.push_back((uint64_t)d);
            }
        }
    }
    fclose(f);
    for (uint64_t idx : faces)
        if (idx >= verts.size()) { set_error("%s: face index out of range", path.c_str()); return false; }
    return true;
}

static double saturate(double v) { return std::fmin(std::fmax(v, 0.0), 1.0); }

// color.rs:50-61
static V3 hsv_to_rgb(double h, double s, double v) {
    V3 hue(saturate(std::fabs(h * 6.0 - 3.0) - 1.0), saturate(2.0 - std::fabs(h * 6.0 - 2.0)),
           saturate(2.0 - std::fabs(h * 6.0 - 4.0)));
    return V3(((hue.x - 1.0) * s + 1.0) * v, ((hue.y - 1.0) * s + 1.0) * v, ((hue.z - 1.0) * s + 1.0) * v);
}

static hr_texture tex_color(V3 c) { hr_texture t{}; t.color = c.c(); t.image = -1; return t; }
static hr_texture tex_one(double v) { return tex_color(V3(v, v, v)); }
static hr_texture tex_image(int img, V3 tint = V3(1, 1, 1)) { hr_texture t{}; t.color = tint.c(); t.image = img; return t; }

static hr_material mat(int surface, double param, hr_texture albedo, hr_texture emission, hr_texture roughness) {
    hr_material m{};
    m.surface = surface; m.param = param; m.albedo = albedo; m.emission = emission; m.roughness = roughness;
    return m;
}

struct Builder {
    hh_scene *sc;
    bool ok = true;

    int add_image_file(const std::string &rel) {
        uint8_t *px = nullptr;
        uint32_t w = 0, h = 0;
        std::string p = sc->root + "/" + rel;
        if (hh_decode_image(p.c_str(), &px, &w, &h) != HR_OK) { ok = false; return -1; }
        sc->image_store.emplace_back(px, px + (size_t)w * h * 4);
        free(px);
        sc->images.push_back(hr_image{sc->image_store.back().data(), w, h});
        return (int)sc->images.size() - 1;
    }
    int add_image_rgba(std::vector<uint8_t> &&px, uint32_t w, uint32_t h) {
        sc->image_store.emplace_back(std::move(px));
        sc->images.push_back(hr_image{sc->image_store.back().data(), w, h});
        return (int)sc->images.size() - 1;
    }
    void add_sphere(V3 c, double r, hr_material m) {
        hr_element e{};
        e.kind = HR_SPHERE; e.material = m; e.center = c.c(); e.radius = r;
        sc->elements.push_back(e);
    }
    void add_cuboid(V3 mn, V3 mx, hr_material m) {
        hr_element e{};
        e.kind = HR_CUBOID; e.material = m; e.aabb_min = mn.c(); e.aabb_max = mx.c();
        sc->elements.push_back(e);
    }
    void add_mesh(const std::string &rel, const M44 &mtx, hr_material m) {
        sc->vertex_store.emplace_back();
        sc->face_store.emplace_back();
        if (!load_obj(sc->root + "/" + rel, mtx, sc->vertex_store.back(), sc->face_store.back())) { ok = false; return; }
        hr_element e{};
        e.kind = HR_MESH; e.material = m;
        e.vertexes = sc->vertex_store.back().data(); e.num_vertexes = sc->vertex_store.back().size();
        e.faces = sc->face_store.back().data(); e.num_faces = sc->face_store.back().size() / 3;
        sc->elements.push_back(e);
    }
    // Intersectable::aabb(): sphere scene.rs:82-87, cuboid scene.rs:187, BvhMesh scene.rs:248 = bounds of the triangles
    // (bvh.rs:51-64,91-99, INF-initialised)
    static void element_aabb(const hr_element &e, V3 &mn, V3 &mx) {
        if (e.kind == HR_SPHERE) { mn = V3(e.center) - V3(e.radius, e.radius, e.radius); mx = V3(e.center) + V3(e.radius, e.radius, e.radius); }
        else if (e.kind == HR_CUBOID) { mn = V3(e.aabb_min); mx = V3(e.aabb_max); }
        else {
            const double INF = 1e100;   // config.rs:9
            mn = V3(INF, INF, INF); mx = V3(-INF, -INF, -INF);
            for (uint64_t f = 0; f < e.num_faces * 3; f++) {
                const hr_vec3 &v = e.vertexes[e.faces[f]];
                mn = V3(std::fmin(mn.x, v.x), std::fmin(mn.y, v.y), std::fmin(mn.z, v.z));
                mx = V3(std::fmax(mx.x, v.x), std::fmax(mx.y, v.y), std::fmax(mx.z, v.z));
            }
        }
    }
    // Scene::add_with_check_collisions (scene.rs:366-376) with Aabb::intersect_aabb (bvh.rs:14-18, strict inequalities): the
    // candidate is the LAST element; it is removed again when its box overlaps the box of any earlier element
    bool keep_last_if_no_collision() {
        V3 mn, mx;
        element_aabb(sc->elements.back(), mn, mx);
        for (size_t i = 0; i + 1 < sc->elements.size(); i++) {
            V3 omn, omx;
            element_aabb(sc->elements[i], omn, omx);
            bool hit = omn.x < mx.x && omx.x > mn.x && omn.y < mx.y && omx.y > mn.y && omn.z < mx.z && omx.z > mn.z;
            if (hit) {
                if (sc->elements.back().kind == HR_MESH) { sc->vertex_store.pop_back(); sc->face_store.pop_back(); }
                sc->elements.pop_back();
                return false;
            }
        }
        return true;
    }
    bool sphere_collides(V3 c, double r) const {
        for (const auto &e : sc->elements) {
            V3 omn, omx;
            element_aabb(e, omn, omx);
            if (omn.x < (c.x + r) && omx.x > (c.x - r) && omn.y < (c.y + r) && omx.y > (c.y - r) && omn.z < (c.z + r) && omx.z > (c.z - r)) return true;
        }
        return false;
    }
    void skybox(const char *dir, V3 intensity) {
        static const char *names[6] = {"posx.jpg", "negx.jpg", "posy.jpg", "negy.jpg", "posz.jpg", "negz.jpg"};
        for (int i = 0; i < 6; i++) sc->desc.skybox.face_image[i] = add_image_file(std::string(dir) + "/" + names[i]);
        sc->desc.skybox.intensity = intensity.c();
    }
};

// main.rs:1020-1153
static void build_rtcamp6_v3_1(Builder &b, bool with_dodecahedron) {
    const double scene_scale = 1.0;
    double theta = PI2 * 0.03;
    double r = 6.5 * scene_scale;
    hh_camera_new(V3(r * std::sin(theta), 2.0 * scene_scale, r * std::cos(theta)).c(), V3(0.0, 1.0 * scene_scale, 0.0).c(),
                  normalize(V3(0, 1, 0)).c(), 20.0, 1, 0.03, 5.0 * scene_scale, &b.sc->desc.camera);

    double radius = 0.2, floor_s = 9.0 * scene_scale;
    // 0: light
    b.add_sphere(V3(-0.3, 0.5 + radius, 0.0) * scene_scale, radius * scene_scale,
                 mat(HR_DIFFUSE, 0, tex_one(0), tex_color(V3(30.0, 20.0, 4.0)), tex_one(0)));
    // 1: bunny
    b.add_mesh("models/bunny/bunny_wired_300.obj",
               M44::scale_linear(1.5 * scene_scale) * M44::translate(0, 0, 0) * M44::rotate_y(0.3),
               mat(HR_GGX, 0.8, tex_color(V3(1.0, 0.01, 0.01)), tex_one(0), tex_one(0.05)));
    // 2: mirror
    b.add_mesh("models/box.obj",
               M44::translate(1.0 * scene_scale, 0.0, -3.0 * scene_scale) * M44::rotate_y(-PI / 8.0) *
                   M44::scale(4.0 * 0.9 * scene_scale, 3.0 * 0.9 * scene_scale, 0.1 * 0.9 * scene_scale),
               mat(HR_SPECULAR, 0, tex_one(1), tex_one(0), tex_one(0)));
    // 3: picture frame
    b.add_mesh("models/picture_frame.obj",
               M44::translate(1.0 * scene_scale, 0.0, -3.0 * scene_scale) * M44::rotate_y(-PI / 8.0) *
                   M44::scale(4.0 * scene_scale, 3.0 * scene_scale, scene_scale),
               mat(HR_GGX, 0.9, tex_color(V3(0.33, 0.27, 0.22)), tex_one(0), tex_one(0.3)));
    // 4: floor
    int floor_img = b.add_image_file("textures/2d/magic-circle3.png");
    b.add_cuboid(V3(-floor_s, -1.0, -floor_s), V3(floor_s, 0.0, floor_s),
                 mat(HR_DIFFUSE, 0, tex_image(floor_img), tex_one(0), tex_one(1)));
    b.skybox("textures/cube/Powerlines", V3(1, 1, 1));

    // 5..10: armadillos
    const int count = 6;
    for (int i = 0; i < count; i++) {
        double rr = 2.2 * scene_scale;
        double dr = (double)i / (double)count;
        double th = PI2 * dr;
        double px = rr * std::sin(th), py = 0.0, pz = rr * std::cos(th);
        double offset = 0.45;
        double hue = (offset + dr) - std::trunc(offset + dr);  // f64::fract
        hr_material m = (i % 2 == 0)
                            ? mat(HR_REFRACTION, 1.5, tex_color(hsv_to_rgb(hue, 0.2, 1.0)), tex_one(0), tex_one(0.1))
                            : mat(HR_GGX, 0.8, tex_color(hsv_to_rgb(hue, 1.0, 1.0)), tex_one(0), tex_one(0.05 * (double)i));
        b.add_mesh("models/armadilo_1000.obj", M44::translate(px, py, pz) * M44::rotate_y(th) * M44::scale_linear(scene_scale), m);
    }
    if (with_dodecahedron) {
        // BASELINE config 5 (build-defined placement): material of main.rs:910-915
        b.add_mesh("models/fractal_dodecahedron.obj", M44::translate(0.0, 3.2, -1.0) * M44::scale_linear(0.6),
                   mat(HR_REFRACTION, 1.5, tex_color(V3(0.7, 0.7, 1.0)), tex_one(0), tex_one(0.1)));
    }
}

// main.rs:928-1017 (not live in main()): two NEE emitters, one of them a 1 mm "camera light",
// strong depth of field, white diffuse floor
static void build_rtcamp6_v3(Builder &b) {
    hh_camera_new(V3(0.0, 2.0, 6.0).c(), V3(0.0, 1.0, 0.0).c(), normalize(V3(0, 1, 0)).c(), 20.0, 1, 0.2, 4.9, &b.sc->desc.camera);
    const hr_camera &cam = b.sc->desc.camera;
    double radius = 0.2;
    b.add_sphere(V3(-0.3, 0.5 + radius, 0.0), radius, mat(HR_DIFFUSE, 0, tex_one(0), tex_one(10.0), tex_one(0)));
    b.add_sphere(V3(cam.eye) - V3(cam.forward), 0.001, mat(HR_DIFFUSE, 0, tex_one(0), tex_one(1000.0), tex_one(0)));
    b.add_mesh("models/bunny/bunny_wired_300.obj", M44::scale_linear(1.5) * M44::translate(0, 0, 0) * M44::rotate_y(0.3),
               mat(HR_GGX, 0.8, tex_color(V3(1.0, 0.01, 0.01)), tex_one(0), tex_one(0.05)));
    b.add_cuboid(V3(-5.0, -1.0, -5.0), V3(5.0, 0.0, 5.0), mat(HR_DIFFUSE, 0, tex_one(1), tex_one(0), tex_one(1)));
    b.skybox("textures/cube/Powerlines", V3(1, 1, 1));
}

// main.rs:54-136: GGX floor with image albedo AND image roughness, two coloured NEE emitters,
// aperture 0, LancellottiChapel skybox at intensity 0
static void build_simple(Builder &b) {
    hh_camera_new(V3(0.0, 2.0, 9.0).c(), V3(0.0, 1.0, 0.0).c(), normalize(V3(0, 1, 0)).c(), 10.0, 1, 0.2 * 0.0, 8.8, &b.sc->desc.camera);
    double radius = 0.6;
    b.add_sphere(V3(0.0, radius, 0.0), radius, mat(HR_DIFFUSE, 0, tex_one(1), tex_one(0), tex_one(0.99)));
    b.add_sphere(V3(3.0, 2.0 + radius, -2.0), radius * 0.2, mat(HR_DIFFUSE, 0, tex_one(0), tex_color(V3(200.0, 10.0, 10.0)), tex_one(0.05)));
    b.add_sphere(V3(-3.0, 2.0 + radius, -2.0), radius * 0.2, mat(HR_DIFFUSE, 0, tex_one(0), tex_color(V3(10.0, 200.0, 10.0)), tex_one(0.05)));
    int albedo = b.add_image_file("textures/2d/checkered_diagonal_10_0.5_1.0_512.png");
    int rough = b.add_image_file("textures/2d/checkered_diagonal_10_0.1_0.6_512.png");
    b.add_cuboid(V3(-5.0, -1.0, -5.0), V3(5.0, 0.0, 5.0), mat(HR_GGX, 0.8, tex_image(albedo), tex_one(0), tex_image(rough)));
    b.skybox("textures/cube/LancellottiChapel", V3(0, 0, 0));
}

// main.rs:139-250: one sphere per surface type (Diffuse, GGX, Specular, Refraction, GGXRefraction) under
// a spherical light, checkered diffuse floor, LancellottiChapel skybox at intensity 1
static void build_material_examples(Builder &b) {
    hh_camera_new(V3(0.0, 2.0, 9.0).c(), V3(0.0, 1.0, 0.0).c(), normalize(V3(0, 1, 0)).c(), 10.0, 1, 0.2, 8.8, &b.sc->desc.camera);
    const double radius = 0.4;
    const hr_texture white = tex_one(1), black = tex_one(0), rough = tex_one(0.05);
    b.add_sphere(V3(-2.0, radius, 0.0), radius, mat(HR_DIFFUSE, 0, white, black, rough));
    b.add_sphere(V3(-1.0, radius, 0.0), radius, mat(HR_GGX, 0.8, white, black, rough));
    b.add_sphere(V3(0.0, radius, 0.0), radius, mat(HR_SPECULAR, 0, white, black, rough));
    b.add_sphere(V3(1.0, radius, 0.0), radius, mat(HR_REFRACTION, 1.5, white, black, rough));
    b.add_sphere(V3(2.0, radius, 0.0), radius, mat(HR_GGX_REFRACTION, 1.5, white, black, rough));
    b.add_sphere(V3(0.0, 2.0 + radius, -2.0), radius, mat(HR_DIFFUSE, 0, black, tex_one(20.0), rough));
    int albedo = b.add_image_file("textures/2d/checkered_diagonal_10_0.5_1.0_512.png");
    int roughness = b.add_image_file("textures/2d/checkered_diagonal_10_0.1_0.6_512.png");
    b.add_cuboid(V3(-5.0, -1.0, -5.0), V3(5.0, 0.0, 5.0), mat(HR_DIFFUSE, 0, tex_image(albedo), black, tex_image(roughness)));
    b.skybox("textures/cube/LancellottiChapel", V3(1, 1, 1));
}

// main.rs:725-802: an emissive sphere inside a refractive mesh (houdini_boss.obj), checkered diffuse floor,
// LancellottiChapel skybox at intensity 0.5, pinhole camera
static void build_rtcamp6_v1(Builder &b) {
    hh_camera_new(V3(0.0, 2.0, 10.0).c(), V3(0.0, 1.0, 0.0).c(), normalize(V3(0, 1, 0)).c(), 10.0, 1, 0.2 * 0.0, 8.8, &b.sc->desc.camera);
    const double radius = 0.6;
    b.add_sphere(V3(0.0, 3.1782 * 0.4, 0.0), radius, mat(HR_DIFFUSE, 0, tex_one(1), tex_one(10.0), tex_one(0.05)));
    b.add_mesh("models/houdini_boss.obj", M44::scale_linear(0.4) * M44::translate(0.0, 3.1782, 2.0) * M44::rotate_y(-0.5),
               mat(HR_REFRACTION, 1.5, tex_color(V3(0.7, 0.7, 1.0)), tex_one(0), tex_one(0.1)));
    int albedo = b.add_image_file("textures/2d/checkered_diagonal_10_0.5_1.0_512.png");
    int roughness = b.add_image_file("textures/2d/checkered_diagonal_10_0.1_0.6_512.png");
    b.add_cuboid(V3(-5.0, -1.0, -5.0), V3(5.0, 0.0, 5.0), mat(HR_DIFFUSE, 0, tex_image(albedo), tex_one(0), tex_image(roughness)));
    b.skybox("textures/cube/LancellottiChapel", V3(0.5, 0.5, 0.5));
}

// The sphere generator of main.rs:862-905 (ISAAC-64 seed [870,2000,304,2], gen_range + AABB-collision rejection): 100 floating
// spheres, then 5 emissive ones.  `reference_materials`: GGX f0 0.9 as in init_scene_rtcamp6_v2; otherwise BASELINE config 2's
// alternating Diffuse / Specular.
static void generate_spheres(Builder &b, bool reference_materials) {
    const uint64_t seed[4] = {870, 2000, 304, 2};  // main.rs:805
    Isaac64 rng;
    rng.from_seed(seed, 4);
    int count = 0;
    while (count < 100) {
        // every attempt consumes 5 draws: px, py, pz, hue, roughness (struct-literal evaluation order)
        double px = rng.gen_range(-0.5, 2.0), py = rng.gen_range(-2.0, 2.0), pz = rng.gen_range(-2.0, 2.0);
        double s = 0.1;
        double hue = rng.gen_range(0.0, 1.0);
        double rough = rng.gen_range(0.0, 1.0);
        if (!b.sphere_collides(V3(px, py, pz), s)) {
            int surface = reference_materials ? HR_GGX : ((count % 2 == 0) ? HR_DIFFUSE : HR_SPECULAR);
            b.add_sphere(V3(px, py, pz), s, mat(surface, reference_materials ? 0.9 : 0.0, tex_color(hsv_to_rgb(hue, 1.0, 1.0)), tex_one(0), tex_one(rough)));
            count++;
        }
    }
    count = 0;
    while (count < 5) {
        double px = rng.gen_range(-0.2, 0.5), py = rng.gen_range(-1.0, 1.0), pz = rng.gen_range(-1.0, 1.0);
        double s = 0.1;
        double hue = rng.gen_range(0.0, 1.0);
        double rough = rng.gen_range(0.0, 1.0);
        if (!b.sphere_collides(V3(px, py, pz), s)) {
            b.add_sphere(V3(px, py, pz), s, mat(HR_DIFFUSE, 0, tex_one(0), tex_color(hsv_to_rgb(hue, 1.0, 1.0) * 10.0), tex_one(rough)));
            count++;
        }
    }
}

// BASELINE config 2 (build-defined): the generated spheres only, Diffuse / Specular, no mesh
static void build_spheres(Builder &b) {
    hh_camera_new(V3(-5.0, -1.0, 0.0).c(), V3(0, 0, 0).c(), normalize(V3(0, 1, 0)).c(), 10.0, 1, 0.2 * 0.0, 8.8,
                  &b.sc->desc.camera);
    b.skybox("textures/cube/Ryfjallet", V3(0.5, 0.5, 0.5));
    generate_spheres(b, false);
}

// main.rs:804-926: 100 GGX spheres + 5 emitters (five shadow rays per NEE-capable hit) around the
// refractive fractal dodecahedron, Ryfjallet skybox at intensity 0.5
static void build_rtcamp6_v2(Builder &b) {
    hh_camera_new(V3(-5.0, -1.0, 0.0).c(), V3(0, 0, 0).c(), normalize(V3(0, 1, 0)).c(), 10.0, 1, 0.2 * 0.0, 8.8,
                  &b.sc->desc.camera);
    b.skybox("textures/cube/Ryfjallet", V3(0.5, 0.5, 0.5));
    generate_spheres(b, true);
    b.add_mesh("models/fractal_dodecahedron.obj", M44::scale_linear(1.0) * M44::translate(0.0, 0.0, 0.0) * M44::rotate_y(0.0),
               mat(HR_REFRACTION, 1.5, tex_color(V3(0.7, 0.7, 1.0)), tex_one(0), tex_one(0.1)));
}

// main.rs:252-500 (the reference's `rtcamp5.png`): two bunnies (refraction / GGX), a sphere with a TEXTURED
// EMISSION (earth map, an NEE emitter), a sphere with an image roughness, five coloured GGX spheres, GGX floor with the TIFF
// marble albedo + PNG roughness, and 1 + 12 + 30 diamonds (refractive index 2.42) placed by ISAAC-64 gen_range draws with
// AABB-collision rejection (mesh boxes included)
static void build_rtcamp5(Builder &b) {
    const uint64_t seed[4] = {870, 2000, 304, 2};
    Isaac64 rng;
    rng.from_seed(seed, 4);
    hh_camera_new(V3(0.0, 2.5, 9.0).c(), V3(0.0, 1.0, 0.0).c(), normalize(V3(0, 1, 0)).c(), 17.0, 1, 0.15, 8.5, &b.sc->desc.camera);
    const hr_texture white = tex_one(1), black = tex_one(0);
    const hr_material diamond = mat(HR_REFRACTION, 2.42, white, black, black);
    b.add_mesh("models/bunny/bunny_face1000.obj", M44::scale_linear(1.5) * M44::translate(1.2, 0.0, 0.0) * M44::rotate_y(0.2),
               mat(HR_REFRACTION, 1.5, tex_color(V3(0.7, 0.7, 1.0)), black, tex_one(0.1)));
    b.add_mesh("models/bunny/bunny_face1000_flip.obj", M44::scale(1.5, 1.5, 1.5) * M44::translate(-1.2, 0.0, 0.0) * M44::rotate_y(-0.2),
               mat(HR_GGX, 0.8, tex_color(V3(1.0, 0.04, 0.04)), black, tex_one(0.1)));
    b.add_mesh("models/dia/dia.obj", M44::translate(3.1, 0.0, 0.8) * M44::scale_linear(1.0) * M44::rotate_y(-0.5) * M44::rotate_x(to_radians(40.35)), diamond);
    int earth = b.add_image_file("textures/2d/earth_inverse_2048.jpg");
    b.add_sphere(V3(0.0, 0.5, -0.5), 0.5, mat(HR_GGX, 0.8, white, tex_image(earth, V3(5.0, 5.0, 2.0)), tex_one(0.05)));
    b.add_sphere(V3(-3.5, 0.5, 0.0), 0.5, mat(HR_GGX, 0.8, tex_color(V3(1.0, 1.0, 1.0)), black, tex_image(earth)));
    b.add_sphere(V3(0.5018854352719382, 0.3899602675366644, 1.8484239850862165), 0.3899602675366644,
                 mat(HR_GGX, 0.8, tex_color(hsv_to_rgb(0.2, 1.0, 1.0)), black, tex_one(0.01)));
    b.add_sphere(V3(-0.5748933256792994, 0.2951263257801348, 2.266298272012876), 0.2951263257801348,
                 mat(HR_GGX, 0.8, tex_color(hsv_to_rgb(0.4, 1.0, 1.0)), black, tex_one(0.05)));
    b.add_sphere(V3(-0.9865234498515534, 0.3386858117447873, 2.9809338871934585), 0.3386858117447873,
                 mat(HR_GGX, 0.8, tex_color(hsv_to_rgb(0.6, 1.0, 1.0)), black, tex_one(0.02)));
    b.add_sphere(V3(0.6946459502665004, 0.2764689077971783, 2.7455446851003025), 0.2764689077971783,
                 mat(HR_GGX, 0.8, tex_color(hsv_to_rgb(0.05, 1.0, 1.0)), black, tex_one(0.0)));
    b.add_sphere(V3(3.7027464198816952, 0.3917608374245498, -0.40505849281451556), 0.3917608374245498,
                 mat(HR_GGX, 0.8, tex_color(hsv_to_rgb(0.8, 1.0, 1.0)), black, tex_one(0.1)));
    int floor_albedo = b.add_image_file("textures/2d/MarbleFloorTiles2/TexturesCom_MarbleFloorTiles2_1024_c_diffuse.tiff");
    int floor_rough = b.add_image_file("textures/2d/MarbleFloorTiles2/TexturesCom_MarbleFloorTiles2_1024_roughness.png");
    b.add_cuboid(V3(-5.0, -1.0, -5.0), V3(5.0, 0.0, 5.0), mat(HR_GGX, 0.8, tex_image(floor_albedo), black, tex_image(floor_rough)));
    b.skybox("textures/cube/LancellottiChapel", V3(1, 1, 1));
    if (!b.ok) return;
    // (the first generator loop of the reference, main.rs:428-449, runs `while count < 0`: no draws)
    // diamonds lying on the floor: px, pz, s, ry per attempt (py is the literal 0.0)
    int count = 0;
    while (count < 12 && b.ok) {
        double px = rng.gen_range(-4.5, 4.5), py = 0.0, pz = rng.gen_range(-2.5, 4.5);
        double s = rng.gen_range(0.7, 1.1);
        double ry = rng.gen_range(-to_radians(180.0), to_radians(180.0));
        b.add_mesh("models/dia/dia.obj", M44::translate(px, py, pz) * M44::scale_linear(s) * M44::rotate_y(ry) * M44::rotate_x(to_radians(40.35)), diamond);
        if (b.ok && b.keep_last_if_no_collision()) count++;
    }
    // floating diamonds: px, py, pz, s, ry, rx per attempt
    count = 0;
    while (count < 30 && b.ok) {
        double px = rng.gen_range(-4.5, 4.5), py = rng.gen_range(0.0, 4.0), pz = rng.gen_range(-4.5, 3.5);
        double s = rng.gen_range(0.6, 1.1);
        double ry = rng.gen_range(-to_radians(180.0), to_radians(180.0));
        double rx = rng.gen_range(-to_radians(180.0), to_radians(180.0));
        b.add_mesh("models/dia/dia.obj", M44::translate(px, py, pz) * M44::scale_linear(s) * M44::rotate_y(ry) * M44::rotate_x(rx), diamond);
        if (b.ok && b.keep_last_if_no_collision()) count++;
    }
}

// main.rs:502-724: the KLab logo mesh (GGX), two fixed + 20 generated diamonds, FOUR earth-textured
// emissive spheres (four NEE emitters), 8 generated GGX spheres, marble floor, LancellottiChapel skybox at intensity (2,2,3).
// ISAAC-64 seed [870,2000,304,1].
static void build_tbf3(Builder &b) {
    const uint64_t seed[4] = {870, 2000, 304, 1};
    Isaac64 rng;
    rng.from_seed(seed, 4);
    hh_camera_new(V3(0.0, 2.5, 9.0).c(), V3(0.0, 1.5, 0.0).c(), normalize(V3(0, 1, 0)).c(), 19.0, 1, 0.18, 7.0, &b.sc->desc.camera);
    const hr_texture white = tex_one(1), black = tex_one(0);
    const hr_material diamond = mat(HR_REFRACTION, 2.42, white, black, black);
    b.add_mesh("models/klab_logo/klab_logo_triangle.obj", M44::scale_linear(0.4) * M44::translate(0.0, 3.1782, 2.0) * M44::rotate_y(-0.5),
               mat(HR_GGX, 0.8, tex_color(V3(0.4, 0.4, 1.0)), black, tex_one(0.05)));
    b.add_mesh("models/dia/dia.obj", M44::translate(1.3, 0.0, 2.2) * M44::scale_linear(1.0) * M44::rotate_y(-0.4) * M44::rotate_x(to_radians(40.35)), diamond);
    b.add_mesh("models/dia/dia.obj", M44::translate(-0.1, 0.0, 2.4) * M44::scale_linear(1.0) * M44::rotate_y(-1.4) * M44::rotate_x(to_radians(40.35)), diamond);
    int earth = b.add_image_file("textures/2d/earth_inverse_2048.jpg");
    b.add_sphere(V3(-1.0, 0.4, 4.0), 0.4, mat(HR_GGX, 0.8, tex_color(V3(1, 1, 1)), tex_image(earth, V3(3.0, 3.0, 1.1)), tex_one(0.01)));
    b.add_sphere(V3(-3.0, 0.4, -3.5), 0.4, mat(HR_GGX, 0.8, tex_color(V3(0.5, 1.0, 1.0)), tex_image(earth, V3(1.0, 3.0, 3.5)), tex_one(0.01)));
    b.add_sphere(V3(4.0, 0.2, -4.5), 0.2, mat(HR_GGX, 0.8, tex_color(V3(0.3, 0.7, 1.0)), tex_image(earth, V3(3.0, 3.0, 1.1)), tex_one(0.01)));
    b.add_sphere(V3(3.0, 0.2, -4.2), 0.2, mat(HR_GGX, 0.8, tex_color(V3(1.0, 0.7, 0.9)), tex_image(earth, V3(2.0, 3.0, 1.0)), tex_one(0.01)));
    int floor_albedo = b.add_image_file("textures/2d/MarbleFloorTiles2/TexturesCom_MarbleFloorTiles2_1024_c_diffuse.tiff");
    int floor_rough = b.add_image_file("textures/2d/MarbleFloorTiles2/TexturesCom_MarbleFloorTiles2_1024_roughness.png");
    b.add_cuboid(V3(-5.0, -1.0, -5.0), V3(5.0, 0.0, 5.0), mat(HR_GGX, 0.8, tex_image(floor_albedo), black, tex_image(floor_rough)));
    b.skybox("textures/cube/LancellottiChapel", V3(2.0, 2.0, 3.0));
    if (!b.ok) return;
    // metal spheres: px, pz, r, roughness per attempt (py is the literal 0.0; the hue depends on the running count)
    int count = 0;
    while (count < 8) {
        double px = rng.gen_range(-3.0, 3.0), py = 0.0, pz = rng.gen_range(-5.0, 5.0);
        double r = rng.gen_range(0.2, 0.4);
        double rough = rng.gen_range(0.0, 0.2);
        b.add_sphere(V3(px, r + py, pz), r, mat(HR_GGX, 0.8, tex_color(hsv_to_rgb(0.2 + 0.1 * (double)count, 1.0, 1.0)), black, tex_one(rough)));
        if (b.keep_last_if_no_collision()) count++;
    }
    // diamonds lying on the floor: px, pz, s, ry per attempt
    count = 0;
    while (count < 20 && b.ok) {
        double px = rng.gen_range(-4.0, 4.0), py = 0.0, pz = rng.gen_range(-5.0, 5.0);
        double s = rng.gen_range(0.7, 1.1);
        double ry = rng.gen_range(-to_radians(180.0), to_radians(180.0));
        b.add_mesh("models/dia/dia.obj", M44::translate(px, py, pz) * M44::scale_linear(s) * M44::rotate_y(ry) * M44::rotate_x(to_radians(40.35)), diamond);
        if (b.ok && b.keep_last_if_no_collision()) count++;
    }
    // (the third generator loop of the reference, main.rs:684-705, runs `while count < 0`: no draws)
}

// Small build-defined scene for tests: every surface type, a textured sphere (lat-long UV), textured
// cuboid, a mesh, two NEE emitters, procedural cubemap.  No asset files needed except models/box.obj.
static void build_cornell_mini(Builder &b) {
    hh_camera_new(V3(0.3, 1.6, 5.5).c(), V3(0, 0.8, 0).c(), normalize(V3(0, 1, 0)).c(), 18.0, 1, 0.05, 5.4, &b.sc->desc.camera);
    auto checker = [](uint32_t n, uint32_t cells, uint8_t lo, uint8_t hi, bool colour) {
        std::vector<uint8_t> px((size_t)n * n * 4);
        for (uint32_t y = 0; y < n; y++)
            for (uint32_t x = 0; x < n; x++) {
                bool on = ((x * cells / n) + (y * cells / n)) & 1;
                uint8_t *o = &px[((size_t)y * n + x) * 4];
                uint8_t v = on ? hi : lo;
                o[0] = v; o[1] = colour ? (uint8_t)(x * 255 / n) : v; o[2] = colour ? (uint8_t)(y * 255 / n) : v; o[3] = 255;
            }
        return px;
    };
    int img_floor = b.add_image_rgba(checker(64, 8, 40, 230, false), 64, 64);
    int img_ball = b.add_image_rgba(checker(32, 4, 60, 250, true), 32, 32);
    int img_rough = b.add_image_rgba(checker(16, 4, 20, 120, false), 16, 16);
    for (int f = 0; f < 6; f++) {
        std::vector<uint8_t> px((size_t)16 * 16 * 4);
        for (uint32_t y = 0; y < 16; y++)
            for (uint32_t x = 0; x < 16; x++) {
                uint8_t *o = &px[((size_t)y * 16 + x) * 4];
                o[0] = (uint8_t)(40 + 30 * f + x * 4); o[1] = (uint8_t)(90 + y * 8); o[2] = (uint8_t)(200 - 20 * f); o[3] = 255;
            }
        b.sc->desc.skybox.face_image[f] = b.add_image_rgba(std::move(px), 16, 16);
    }
    b.sc->desc.skybox.intensity = V3(0.8, 0.9, 1.1).c();

    b.add_cuboid(V3(-3, -0.5, -3), V3(3, 0, 3), mat(HR_GGX, 0.7, tex_image(img_floor), tex_one(0), tex_image(img_rough)));
    b.add_sphere(V3(-1.2, 0.5, 0.2), 0.5, mat(HR_DIFFUSE, 0, tex_image(img_ball, V3(1.0, 0.9, 0.8)), tex_one(0), tex_one(1)));
    b.add_sphere(V3(0.0, 0.45, -0.6), 0.45, mat(HR_SPECULAR, 0, tex_color(V3(0.95, 0.95, 0.9)), tex_one(0), tex_one(0)));
    b.add_sphere(V3(1.1, 0.4, 0.5), 0.4, mat(HR_REFRACTION, 1.5, tex_color(V3(0.9, 1.0, 0.9)), tex_one(0), tex_one(0)));
    b.add_sphere(V3(-0.3, 0.3, 1.2), 0.3, mat(HR_GGX_REFRACTION, 1.33, tex_color(V3(0.9, 0.9, 1.0)), tex_one(0), tex_one(0.2)));
    b.add_sphere(V3(0.6, 1.9, 0.4), 0.25, mat(HR_DIFFUSE, 0, tex_one(0), tex_color(V3(18, 14, 9)), tex_one(0)));
    b.add_sphere(V3(-1.6, 1.4, -0.8), 0.15, mat(HR_DIFFUSE, 0, tex_one(0), tex_image(img_ball, V3(25, 25, 40)), tex_one(0)));
    b.add_mesh("models/box.obj", M44::translate(1.6, 0.0, -1.2) * M44::rotate_y(0.5) * M44::scale(0.8, 1.3, 0.8),
               mat(HR_GGX, 0.6, tex_color(V3(0.8, 0.5, 0.2)), tex_one(0), tex_one(0.35)));
    b.add_cuboid(V3(-2.6, 0.0, -2.2), V3(-1.9, 1.1, -1.6), mat(HR_DIFFUSE, 0, tex_color(V3(0.2, 0.4, 0.8)), tex_one(0), tex_one(1)));
}

}  // namespace

extern "C" {

void hh_camera_new(hr_vec3 eye_, hr_vec3 target_, hr_vec3 y_up_, double v_fov_deg, int32_t lens_shape, double aperture,
                   double focus_distance, hr_camera *out) {
    V3 eye(eye_), target(target_), y_up(y_up_);
    double lens_radius = 0.5 * aperture;
    double plane_half_height = std::tan(v_fov_deg * (PI / 180.0));  // camera.rs:48: tan of the FULL angle given
    V3 forward = normalize(target - eye);
    V3 right = normalize(cross(forward, y_up));
    V3 up = normalize(cross(right, forward));
    memset(out, 0, sizeof *out);
    out->eye = eye.c(); out->right = right.c(); out->up = up.c(); out->forward = forward.c();
    out->plane_half_right = (right * plane_half_height * focus_distance).c();
    out->plane_half_up = (up * plane_half_height * focus_distance).c();
    out->lens_radius = lens_radius; out->focus_distance = focus_distance; out->lens_shape = lens_shape;
}

int hh_load_obj(const char *path, const double *matrix, hr_vec3 **vertexes, uint64_t *num_vertexes, uint64_t **faces,
                uint64_t *num_faces) {
    if (!path || !vertexes || !num_vertexes || !faces || !num_faces) { set_error("hh_load_obj: null argument"); return HR_ERR_INVALID; }
    M44 m = M44::identity();
    if (matrix) memcpy(m.e, matrix, sizeof m.e);
    std::vector<hr_vec3> v;
    std::vector<uint64_t> f;
    if (!load_obj(path, m, v, f)) return HR_ERR_INVALID;
    *vertexes = (hr_vec3 *)malloc(v.size() * sizeof(hr_vec3) + 1);
    *faces = (uint64_t *)malloc(f.size() * sizeof(uint64_t) + 1);
    if (!*vertexes || !*faces) { free(*vertexes); free(*faces); *vertexes = nullptr; *faces = nullptr; set_error("hh_load_obj: out of memory"); return HR_ERR_INVALID; }
    if (!v.empty()) memcpy(*vertexes, v.data(), v.size() * sizeof(hr_vec3));   // an OBJ file without vertices or faces is legal: empty arrays
    if (!f.empty()) memcpy(*faces, f.data(), f.size() * sizeof(uint64_t));
    *num_vertexes = v.size();
    *num_faces = f.size() / 3;
    return HR_OK;
}

int hh_scene_create(const char *name, const char *asset_root, hh_scene **out) {
    if (!name || !asset_root || !out) { set_error("hh_scene_create: null argument"); return HR_ERR_INVALID; }
    std::unique_ptr<hh_scene> sc(new hh_scene);
    sc->root = asset_root;
    Builder b{sc.get()};
    std::string n = name;
    for (int i = 0; i < 6; i++) sc->desc.skybox.face_image[i] = -1;
    if (n == "rtcamp6_v3_1") build_rtcamp6_v3_1(b, false);
    else if (n == "rtcamp6_dodeca") build_rtcamp6_v3_1(b, true);
    else if (n == "rtcamp6_v3") build_rtcamp6_v3(b);
    else if (n == "simple") build_simple(b);
    else if (n == "material_examples") build_material_examples(b);
    else if (n == "rtcamp6_v1") build_rtcamp6_v1(b);
    else if (n == "rtcamp6_v2") build_rtcamp6_v2(b);
    else if (n == "rtcamp5") build_rtcamp5(b);
    else if (n == "tbf3") build_tbf3(b);
    else if (n == "spheres") build_spheres(b);
    else if (n == "cornell_mini") build_cornell_mini(b);
    else { set_error("unknown scene '%s'", name); return HR_ERR_INVALID; }
    if (!b.ok) return HR_ERR_INVALID;
    sc->desc.elements = sc->elements.data();
    sc->desc.num_elements = (uint32_t)sc->elements.size();
    sc->desc.images = sc->images.data();
    sc->desc.num_images = (uint32_t)sc->images.size();
    *out = sc.release();
    return HR_OK;
}

const hr_scene_desc *hh_scene_desc(const hh_scene *scene) { return scene ? &scene->desc : nullptr; }
void hh_scene_destroy(hh_scene *scene) { delete scene; }

// test hook: first n next_u64 of StdRng::from_seed(seed) after skipping `skip` outputs
int hh_debug_isaac64(const uint64_t *seed, int nseed, uint64_t skip, uint64_t *out, int n) {
    Isaac64 r;
    r.from_seed(seed, nseed);
    for (uint64_t i = 0; i < skip; i++) r.next_u64();
    for (int i = 0; i < n; i++) out[i] = r.next_u64();
    return HR_OK;
}

}
