/*
 * oracle.cpp — CPU restatement (f64) of hanamaru-renderer's PathTracingRenderer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (hanamaru-renderer_amd/) never links, imports or
 * executes anything in oracle/.
 *
 * The reference is a Rust crate; there is no Rust toolchain in this environment, so the reference
 * cannot be compiled into oracle/_ref.  This is a restatement of its algorithm, function by function,
 * each citing the reference file:line it follows (paths relative to /root/reference/src).
 *
 * PINNING STATUS:
 *   - the reference has no unit tests or fixtures for this path and cannot be built here, BUT its
 *     repository commits one output of the real binary: rtcamp6_1000x4spp.png (default scene, 1920x1080, -s 1000,
 *     README.md:19).  tests/test_oracle.py renders crops of exactly that configuration with this oracle and gets the
 *     same 8-bit pixels (interior of every tested crop: identical).  That pins — against the Rust program itself —
 *     the per-path seeding, the ISAAC-64 u64->f64 conversion, the scene, the texture decoders, the estimator and
 *     the post chain.  PINNED.
 *   - the third-party RNG (rand 0.4.3 StdRng = ISAAC-64, a Cargo.lock dependency not vendored under
 *     /root/reference) is additionally pinned by rand's published known-answer vectors (tests/test_isaac64.py).
 *   - structural known answers (BVH shapes; path statistics, Appendix D.2): tests/test_oracle.py.
 */
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <vector>

#include "hanamaru_hip.h"  // only the POD scene description (hr_scene_desc) — no product code is linked

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace orc {

// ---------------------------------------------------------------------------------------------
// config.rs:1-25
static const double PI = 3.14159265358979323846;
static const double PI2 = 2.0 * PI;
static const double EPS = 1e-4;
static const double OFFSET = 1e-4;
static const double INF = 1e100;
static const double GAMMA_FACTOR = 2.2;
static const uint32_t SUPERSAMPLING = 2;
static const uint32_t PATHTRACING_BOUNCE_LIMIT = 10;
static const double TONE_MAPPING_EXPOSURE = 1.5;
static const double TONE_MAPPING_WHITE_POINT = 20.0;
static const uint32_t BILATERAL_FILTER_ITERATION = 1;
static const uint32_t BILATERAL_FILTER_DIAMETER = 3;
static const double BILATERAL_FILTER_SIGMA_I = 1.0;
static const double BILATERAL_FILTER_SIGMA_S = 16.0;

// ---------------------------------------------------------------------------------------------
// vector.rs
struct V3 {
    double x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    explicit V3(const hr_vec3 &v) : x(v.x), y(v.y), z(v.z) {}
};
static inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator*(V3 a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(double s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
static inline V3 operator/(V3 a, V3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline V3 operator/(V3 a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
static inline bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }  // vector.rs:218-222
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }             // :48-50
static inline V3 cross(V3 a, V3 b) {                                                           // :52-58
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double norm(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }  // :35-37 — SQUARED length
static inline double length(V3 a) { return std::sqrt(norm(a)); }               // :31-33
static inline V3 normalize(V3 a) {                                              // :39-46
    double inv_len = 1.0 / length(a);
    return V3(a.x * inv_len, a.y * inv_len, a.z * inv_len);
}
static inline V3 reflect(V3 v, V3 n) { return v - 2.0 * dot(v, n) * n; }  // :60-62
static inline V3 refract(V3 v, V3 n, double ri) {                          // :64-71
    double k = 1.0 - ri * ri * (1.0 - dot(n, v) * dot(v, n));
    if (k < 0.0) return V3();
    return ri * v - (ri * dot(v, n) + std::sqrt(k)) * n;
}
static inline bool approximately(V3 a, V3 b) { return norm(a - b) < OFFSET * 4.0; }  // :89-91

// math.rs
static inline double clampd(double v, double lo, double hi) { return std::fmin(std::fmax(v, lo), hi); }  // :9-11
static inline double saturate(double v) { return clampd(v, 0.0, 1.0); }                                  // :17-19
static inline uint32_t clamp_u32(uint32_t x, uint32_t lo, uint32_t hi) { return x < lo ? lo : (x > hi ? hi : x); }  // :13-15
static inline bool equals_eps(double a, double b) { return std::fabs(a - b) < EPS; }                     // :21-23
static inline double det(V3 a, V3 b, V3 c) {                                                             // :25-32
    return (a.x * b.y * c.z) + (a.y * b.z * c.x) + (a.z * b.x * c.y) - (a.x * b.z * c.y) - (a.y * b.x * c.z) - (a.z * b.y * c.x);
}
static inline bool sign_negative(double v) { return std::signbit(v); }
static inline bool sign_positive(double v) { return !std::signbit(v); }
// Rust `f64 as u32` (saturating, NaN -> 0)
static inline uint32_t f64_as_u32(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}
// Rust `f64 as usize` on a 64-bit target: truncation toward zero, saturating at both ends, NaN -> 0 (the language's defined behaviour since
// Rust 1.45; a plain C cast of a negative double is undefined and wraps on x86).  It matters for the seed words of renderer.rs:165-166:
// (4 + nc) goes negative when an image is more than four times as wide as high (or as high as wide) — there the word is 0.
static inline uint64_t f64_as_usize(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return ~0ULL;
    return (uint64_t)v;
}

// ---------------------------------------------------------------------------------------------
// rand 0.4.3 StdRng = Isaac64Rng (public-domain ISAAC-64 by Bob Jenkins)
struct Isaac64 {
    uint64_t rsl[256], mem[256];
    uint64_t a, b, c;
    uint32_t cnt;
    uint64_t draws;  // statistics only

    static inline void mix(uint64_t &a, uint64_t &b, uint64_t &c, uint64_t &d, uint64_t &e, uint64_t &f, uint64_t &g, uint64_t &h) {
        a -= e; f ^= h >> 9;  h += a;
        b -= f; g ^= a << 9;  a += b;
        c -= g; h ^= b >> 23; b += c;
        d -= h; a ^= c << 15; c += d;
        e -= a; b ^= d >> 14; d += e;
        f -= b; c ^= e << 20; e += f;
        g -= c; d ^= f >> 17; f += g;
        h -= d; e ^= g << 14; g += h;
    }
    void from_seed(const uint64_t *seed, int n) {  // SeedableRng<&[usize]>::from_seed: seed ++ zeros, a=b=c=0, init(true)
        for (int i = 0; i < 256; i++) rsl[i] = i < n ? seed[i] : 0;
        a = b = c = 0;
        draws = 0;
        uint64_t r0, r1, r2, r3, r4, r5, r6, r7;
        r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = 0x9e3779b97f4a7c13ULL;
        for (int i = 0; i < 4; i++) mix(r0, r1, r2, r3, r4, r5, r6, r7);
        for (int pass = 0; pass < 2; pass++) {
            const uint64_t *src = pass ? mem : rsl;
            for (int i = 0; i < 256; i += 8) {
                r0 += src[i]; r1 += src[i + 1]; r2 += src[i + 2]; r3 += src[i + 3];
                r4 += src[i + 4]; r5 += src[i + 5]; r6 += src[i + 6]; r7 += src[i + 7];
                mix(r0, r1, r2, r3, r4, r5, r6, r7);
                mem[i] = r0; mem[i + 1] = r1; mem[i + 2] = r2; mem[i + 3] = r3;
                mem[i + 4] = r4; mem[i + 5] = r5; mem[i + 6] = r6; mem[i + 7] = r7;
            }
        }
        isaac64();
    }
    void isaac64() {
        c += 1;
        uint64_t aa = a, bb = b + c;
        static const int MP[2][2] = {{0, 128}, {128, 0}};
        for (int hlf = 0; hlf < 2; hlf++) {
            int mr = MP[hlf][0], m2 = MP[hlf][1];
            for (int base = 0; base < 128; base += 4) {
                for (int j = 0; j < 4; j++) {
                    uint64_t mixv;
                    if (j == 0) mixv = ~(aa ^ (aa << 21));
                    else if (j == 1) mixv = aa ^ (aa >> 5);
                    else if (j == 2) mixv = aa ^ (aa << 12);
                    else mixv = aa ^ (aa >> 33);
                    uint64_t x = mem[base + j + mr];
                    aa = mixv + mem[base + j + m2];
                    uint64_t y = mem[(x >> 3) & 255] + aa + bb;
                    mem[base + j + mr] = y;
                    bb = mem[(y >> 11) & 255] + x;
                    rsl[base + j + mr] = bb;
                }
            }
        }
        a = aa; b = bb; cnt = 256;
    }
    uint64_t next_u64() {
        if (cnt == 0) isaac64();
        cnt -= 1;
        draws++;
        return rsl[cnt & 255];
    }
    double next_f64() {
#ifdef ORC_F64_FROM_TOP53
        return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0);
#else
        uint64_t bits = 0x3FF0000000000000ULL | (next_u64() & 0x000FFFFFFFFFFFFFULL);
        double d;
        memcpy(&d, &bits, 8);
        return d - 1.0;
#endif
    }
};

// ---------------------------------------------------------------------------------------------
struct Ray { V3 origin, direction; };  // camera.rs:38-42

struct PointMaterial {                 // material.rs:25-31
    int surface; double param;
    V3 albedo, emission; double roughness;
};
struct Intersection {                  // scene.rs:11-18
    V3 position; double distance; V3 normal; double u, v; PointMaterial material;
    long face = -1;   // bookkeeping for the path log only: the mesh face of the closest triangle hit so far (not a field of scene.rs:11-18)
};
static Intersection intersection_empty() {  // scene.rs:26-39
    Intersection i;
    i.position = V3(); i.distance = INF; i.normal = V3(); i.u = i.v = 0.0;
    i.material.surface = HR_DIFFUSE; i.material.param = 0.0;
    i.material.albedo = V3(1, 1, 1); i.material.emission = V3(); i.material.roughness = 0.2;
    return i;
}

struct Counters {
    uint64_t paths, rays_primary, rays_bounce, rays_shadow, surface_hits, draws, tex_samples, sky_lookups;
    uint64_t top_node_tests, mesh_roots, mesh_node_tests, tri_tests, tri_accepted, sphere_tests, cuboid_tests;
    uint64_t rays_per_path_hist[24], draws_per_path_hist[40];
};

struct Aabb {  // bvh.rs:7-11
    V3 min, max;
    // bvh.rs:20-39
    bool intersect_ray(const Ray &ray, double &distance) const {
        V3 dir_inv(1.0 / ray.direction.x, 1.0 / ray.direction.y, 1.0 / ray.direction.z);
        double t1 = (min.x - ray.origin.x) * dir_inv.x, t2 = (max.x - ray.origin.x) * dir_inv.x;
        double t3 = (min.y - ray.origin.y) * dir_inv.y, t4 = (max.y - ray.origin.y) * dir_inv.y;
        double t5 = (min.z - ray.origin.z) * dir_inv.z, t6 = (max.z - ray.origin.z) * dir_inv.z;
        // Rust f64::min/max == IEEE minNum/maxNum == C fmin/fmax (NaN -> the other operand)
        double tmin = std::fmax(std::fmax(std::fmin(t1, t2), std::fmin(t3, t4)), std::fmin(t5, t6));
        double tmax = std::fmin(std::fmin(std::fmax(t1, t2), std::fmax(t3, t4)), std::fmax(t5, t6));
        bool hit = tmin <= tmax && sign_positive(tmax);
        distance = sign_positive(tmin) ? tmin : tmax;
        return hit;
    }
    void merge(const Aabb &o) {  // :41-49
        min.x = std::fmin(min.x, o.min.x); min.y = std::fmin(min.y, o.min.y); min.z = std::fmin(min.z, o.min.z);
        max.x = std::fmax(max.x, o.max.x); max.y = std::fmax(max.y, o.max.y); max.z = std::fmax(max.z, o.max.z);
    }
};

struct BvhNode {  // bvh.rs:67-77
    Aabb aabb;
    std::unique_ptr<BvhNode> children[2];
    std::vector<size_t> indexes;
    bool leaf() const { return !children[0]; }
};

struct Image { const uint8_t *rgba; uint32_t width, height; };

struct Texture {  // texture.rs:72-75
    V3 color; const Image *image;
};
struct Material { int surface; double param; Texture albedo, emission, roughness; };

struct Element {
    int kind; Material material;
    V3 center; double radius;     // sphere
    Aabb box;                     // cuboid
    std::vector<V3> vertexes;     // mesh
    std::vector<size_t> faces;    // 3 per face
    std::unique_ptr<BvhNode> bvh; // BvhMesh
    size_t face_base = 0;         // path log only: faces of the meshes before this one (element order) — the HIP side's input triangle index
};

struct Scene {
    std::vector<Element> elements;
    std::vector<Image> images;
    std::vector<std::vector<uint8_t>> image_store;
    const Image *sky[6];
    V3 sky_intensity;
    hr_camera camera;
    std::unique_ptr<BvhNode> top;       // BvhScene.bvh (scene.rs:379-383)
    std::vector<size_t> emissions;      // Scene::emissions (scene.rs:356-358)
};

// ---------------------------------------------------------------------------------------------
// textures — texture.rs
static inline V3 rgba_to_color(const uint8_t *p) {  // color.rs:18-24
    return V3((double)p[0] / 255.0, (double)p[1] / 255.0, (double)p[2] / 255.0);
}
static inline V3 gamma_to_linear(V3 c) {            // color.rs:26-36
    return V3(std::pow(c.x, GAMMA_FACTOR), std::pow(c.y, GAMMA_FACTOR), std::pow(c.z, GAMMA_FACTOR));
}
static inline V3 sample_nearest_screen(const Image &im, uint32_t x, uint32_t y) {  // texture.rs:59-63
    x = clamp_u32(x, 0, im.width - 1);
    y = clamp_u32(im.height - y - 1u, 0, im.height - 1);  // wrapping u32 arithmetic (release build)
    return rgba_to_color(&im.rgba[((size_t)y * im.width + x) * 4]);
}
// path-log bookkeeping (PathLog below; no influence on the algorithm): while calc_pixel traces a MAIN ray with a log attached, every bilinear
// lookup adds a 16-bit hash of its integer corner (x1, y1) to *t_quad_sum — the texel quad the value is interpolated in
static thread_local uint32_t *t_quad_sum = nullptr;
static inline uint32_t quad_hash16(uint32_t ix, uint32_t iy) { return ((ix * 0x9E3779B1u) ^ (iy * 0x85EBCA77u)) >> 16; }
static V3 sample_bilinear(const Image &im, double u, double v) {  // texture.rs:29-49
    double x = u * (double)im.width, y = v * (double)im.height;
    double x1 = std::floor(x), y1 = std::floor(y);
    double x2 = x1 + 1.0, y2 = y1 + 1.0;
    if (t_quad_sum) *t_quad_sum = (*t_quad_sum + quad_hash16(f64_as_u32(x1), f64_as_u32(y1))) & 0xffffu;
    V3 p11 = sample_nearest_screen(im, f64_as_u32(x1), f64_as_u32(y1));
    V3 p12 = sample_nearest_screen(im, f64_as_u32(x1), f64_as_u32(y2));
    V3 p21 = sample_nearest_screen(im, f64_as_u32(x2), f64_as_u32(y1));
    V3 p22 = sample_nearest_screen(im, f64_as_u32(x2), f64_as_u32(y2));
    V3 gamma = (p11 * (x2 - x) * (y2 - y) + p21 * (x - x1) * (y2 - y) + p12 * (x2 - x) * (y - y1) + p22 * (x - x1) * (y - y1)) /
               ((x2 - x1) * (y2 - y1));
    return gamma_to_linear(gamma);
}
static inline V3 sample_bilinear_0center(const Image &im, double u, double v) {  // texture.rs:22-26
    return sample_bilinear(im, 0.5 * (u + 1.0), 0.5 * (v + 1.0));
}
static inline V3 texture_sample(const Texture &t, double u, double v, Counters *cn) {  // texture.rs:108-114
    if (t.image) {
        if (cn) cn->tex_samples++;
        return sample_bilinear(*t.image, u, v) * t.color;
    }
    return t.color;
}
// the face skybox_sample() below looks up (path log only)
static int skybox_face(V3 d) {
    double ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
    if (ax > ay && ax > az) return sign_positive(d.x) ? 0 : 1;
    if (ay > ax && ay > az) return sign_positive(d.y) ? 2 : 3;
    return sign_positive(d.z) ? 4 : 5;
}
static V3 skybox_sample(const Scene &s, V3 d) {  // scene.rs:295-319
    double ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
    if (ax > ay && ax > az) {
        if (sign_positive(d.x)) return s.sky_intensity * sample_bilinear_0center(*s.sky[0], -d.z / d.x, d.y / d.x);
        return s.sky_intensity * sample_bilinear_0center(*s.sky[1], -d.z / d.x, -d.y / d.x);
    } else if (ay > ax && ay > az) {
        if (sign_positive(d.y)) return s.sky_intensity * sample_bilinear_0center(*s.sky[2], d.x / d.y, -d.z / d.y);
        return s.sky_intensity * sample_bilinear_0center(*s.sky[3], -d.x / d.y, -d.z / d.y);
    } else {
        if (sign_positive(d.z)) return s.sky_intensity * sample_bilinear_0center(*s.sky[4], d.x / d.z, d.y / d.z);
        return s.sky_intensity * sample_bilinear_0center(*s.sky[5], d.x / d.z, -d.y / d.z);
    }
}

// ---------------------------------------------------------------------------------------------
// primitives
// bvh.rs:266-290
static bool intersect_polygon(V3 v0, V3 v1, V3 v2, const Ray &ray, Intersection &isect, Counters *cn) {
    if (cn) cn->tri_tests++;
    V3 ray_inv = -ray.direction;
    V3 edge1 = v1 - v0, edge2 = v2 - v0;
    double denominator = det(edge1, edge2, ray_inv);
    if (denominator == 0.0) return false;
    double denominator_inv = 1.0 / denominator;
    V3 d = ray.origin - v0;
    double u = det(d, edge2, ray_inv) * denominator_inv;
    if (u < 0.0 || u > 1.0) return false;
    double v = det(edge1, d, ray_inv) * denominator_inv;
    if (v < 0.0 || u + v > 1.0) return false;
    double t = det(edge1, edge2, d) * denominator_inv;
    if (t < 0.0 || t > isect.distance) return false;
    isect.position = ray.origin + ray.direction * t;
    isect.normal = normalize(cross(edge1, edge2));
    isect.distance = t;
    isect.u = u; isect.v = v;
    if (cn) cn->tri_accepted++;
    return true;
}
static inline double signum(double v) { return std::isnan(v) ? v : (std::signbit(v) ? -1.0 : 1.0); }
// scene.rs:58-78
static bool sphere_intersect(const Element &e, const Ray &ray, Intersection &isect, Counters *cn) {
    if (cn) cn->sphere_tests++;
    V3 a = ray.origin - e.center;
    double b = dot(a, ray.direction);
    double c = dot(a, a) - e.radius * e.radius;
    double d = b * b - c;
    double t = -b - std::sqrt(d);
    if (d > 0.0 && t > 0.0 && t < isect.distance) {
        isect.position = ray.origin + ray.direction * t;
        isect.distance = t;
        isect.normal = normalize(isect.position - e.center);
        isect.v = 1.0 - std::acos(isect.normal.y) / PI;
        double xz_len = std::sqrt(isect.normal.x * isect.normal.x + isect.normal.z * isect.normal.z);
        isect.u = 0.5 - signum(isect.normal.z) * std::acos(isect.normal.x / xz_len) / PI2;
        return true;
    }
    return false;
}
// scene.rs:152-183
static bool cuboid_intersect(const Element &e, const Ray &ray, Intersection &isect, Counters *cn) {
    if (cn) cn->cuboid_tests++;
    double distance;
    bool hit = e.box.intersect_ray(ray, distance);
    if (hit && distance < isect.distance) {
        isect.position = ray.origin + ray.direction * distance;
        isect.distance = distance;
        V3 uvw = (isect.position - e.box.min) / (e.box.max - e.box.min);
        if (equals_eps(isect.position.y, e.box.max.y)) { isect.normal = V3(0, 1, 0); isect.u = uvw.x; isect.v = 1.0 - uvw.z; }
        else if (equals_eps(isect.position.y, e.box.min.y)) { isect.normal = V3(0, -1, 0); isect.u = uvw.x; isect.v = 1.0 - uvw.z; }
        else if (equals_eps(isect.position.x, e.box.min.x)) { isect.normal = V3(-1, 0, 0); isect.u = uvw.z; isect.v = uvw.y; }
        else if (equals_eps(isect.position.x, e.box.max.x)) { isect.normal = V3(1, 0, 0); isect.u = uvw.z; isect.v = uvw.y; }
        else if (equals_eps(isect.position.z, e.box.min.z)) { isect.normal = V3(0, 0, -1); isect.u = uvw.x; isect.v = uvw.y; }
        else if (equals_eps(isect.position.z, e.box.max.z)) { isect.normal = V3(0, 0, 1); isect.u = uvw.x; isect.v = uvw.y; }
        return true;
    }
    return false;
}

static Aabb element_aabb(const Element &e) {
    if (e.kind == HR_SPHERE) {  // scene.rs:82-87
        V3 r(e.radius, e.radius, e.radius);
        return Aabb{e.center - r, e.center + r};
    }
    if (e.kind == HR_CUBOID) return e.box;  // :187
    return e.bvh->aabb;                     // :248
}

// ---------------------------------------------------------------------------------------------
// BVH build — bvh.rs:79-211 (median split by count on the longest axis, stable sort, leaf iff len/2 <= 2)
static BvhNode *node_empty() {
    BvhNode *n = new BvhNode;
    n->aabb.min = V3(INF, INF, INF);
    n->aabb.max = V3(-INF, -INF, -INF);
    return n;
}
static Aabb aabb_from_triangle(V3 v0, V3 v1, V3 v2) {  // bvh.rs:51-64
    Aabb a;
    a.min = V3(std::fmin(std::fmin(v0.x, v1.x), v2.x), std::fmin(std::fmin(v0.y, v1.y), v2.y), std::fmin(std::fmin(v0.z, v1.z), v2.z));
    a.max = V3(std::fmax(std::fmax(v0.x, v1.x), v2.x), std::fmax(std::fmax(v0.y, v1.y), v2.y), std::fmax(std::fmax(v0.z, v1.z), v2.z));
    return a;
}
static std::unique_ptr<BvhNode> build_mesh(const Element &m, std::vector<size_t> &face_indexes) {  // bvh.rs:107-153
    std::unique_ptr<BvhNode> node(node_empty());
    for (size_t fi : face_indexes)
        node->aabb.merge(aabb_from_triangle(m.vertexes[m.faces[fi * 3]], m.vertexes[m.faces[fi * 3 + 1]], m.vertexes[m.faces[fi * 3 + 2]]));
    size_t mid = face_indexes.size() / 2;
    if (mid <= 2) {
        node->indexes = face_indexes;
    } else {
        double lx = node->aabb.max.x - node->aabb.min.x, ly = node->aabb.max.y - node->aabb.min.y, lz = node->aabb.max.z - node->aabb.min.z;
        int axis = (lx > ly && lx > lz) ? 0 : ((ly > lx && ly > lz) ? 1 : 2);
        auto key = [&](size_t f) {
            const V3 &a = m.vertexes[m.faces[f * 3]], &b = m.vertexes[m.faces[f * 3 + 1]], &c = m.vertexes[m.faces[f * 3 + 2]];
            return axis == 0 ? a.x + b.x + c.x : (axis == 1 ? a.y + b.y + c.y : a.z + b.z + c.z);
        };
        std::stable_sort(face_indexes.begin(), face_indexes.end(), [&](size_t a, size_t b) { return key(a) < key(b); });
        std::vector<size_t> left(face_indexes.begin() + mid, face_indexes.end());  // split_off(mid)
        face_indexes.resize(mid);
        node->children[0] = build_mesh(m, face_indexes);
        node->children[1] = build_mesh(m, left);
    }
    return node;
}
static std::unique_ptr<BvhNode> build_scene(const Scene &s, std::vector<size_t> &indexes) {  // bvh.rs:155-201
    std::unique_ptr<BvhNode> node(node_empty());
    for (size_t i : indexes) node->aabb.merge(element_aabb(s.elements[i]));
    size_t mid = indexes.size() / 2;
    if (mid <= 2) {
        node->indexes = indexes;
    } else {
        double lx = node->aabb.max.x - node->aabb.min.x, ly = node->aabb.max.y - node->aabb.min.y, lz = node->aabb.max.z - node->aabb.min.z;
        int axis = (lx > ly && lx > lz) ? 0 : ((ly > lx && ly > lz) ? 1 : 2);
        auto key = [&](size_t i) {
            Aabb a = element_aabb(s.elements[i]);
            return axis == 0 ? a.min.x + a.max.x : (axis == 1 ? a.min.y + a.max.y : a.min.z + a.max.z);
        };
        std::stable_sort(indexes.begin(), indexes.end(), [&](size_t a, size_t b) { return key(a) < key(b); });
        std::vector<size_t> left(indexes.begin() + mid, indexes.end());
        indexes.resize(mid);
        node->children[0] = build_scene(s, indexes);
        node->children[1] = build_scene(s, left);
    }
    return node;
}

// ---------------------------------------------------------------------------------------------
// traversal — bvh.rs:213-263 (recursive DFS, child 0 then child 1, no distance culling)
static bool intersect_for_mesh(const BvhNode &n, const Element &m, const Ray &ray, Intersection &isect, Counters *cn) {
    if (cn) cn->mesh_node_tests++;
    double dist;
    if (!n.aabb.intersect_ray(ray, dist)) return false;
    bool any_hit = false;
    if (n.leaf()) {
        for (size_t fi : n.indexes)
            if (intersect_polygon(m.vertexes[m.faces[fi * 3]], m.vertexes[m.faces[fi * 3 + 1]], m.vertexes[m.faces[fi * 3 + 2]], ray, isect, cn)) {
                any_hit = true;
                isect.face = (long)fi;
            }
    } else {
        for (int c = 0; c < 2; c++)
            if (intersect_for_mesh(*n.children[c], m, ray, isect, cn)) any_hit = true;
    }
    return any_hit;
}
static bool element_intersect(const Element &e, const Ray &ray, Intersection &isect, Counters *cn) {
    switch (e.kind) {
        case HR_SPHERE: return sphere_intersect(e, ray, isect, cn);
        case HR_CUBOID: return cuboid_intersect(e, ray, isect, cn);
        default:
            if (cn) cn->mesh_roots++;
            return intersect_for_mesh(*e.bvh, e, ray, isect, cn);  // scene.rs:242-244
    }
}
static long intersect_for_scene(const BvhNode &n, const Scene &s, const Ray &ray, Intersection &isect, Counters *cn) {
    if (cn) cn->top_node_tests++;
    double dist;
    if (!n.aabb.intersect_ray(ray, dist)) return -1;
    long nearest = -1;
    if (n.leaf()) {
        for (size_t idx : n.indexes)
            if (element_intersect(s.elements[idx], ray, isect, cn)) nearest = (long)idx;
    } else {
        for (int c = 0; c < 2; c++) {
            long r = intersect_for_scene(*n.children[c], s, ray, isect, cn);
            if (r >= 0) nearest = r;
        }
    }
    return nearest;
}
// scene.rs:385-401
static bool scene_intersect(const Scene &s, const Ray &ray, Intersection &isect, long *element, Counters *cn) {
    isect = intersection_empty();
    long idx = intersect_for_scene(*s.top, s, ray, isect, cn);
    if (element) *element = idx;
    if (idx >= 0) {
        const Material &m = s.elements[idx].material;
        isect.material.surface = m.surface;
        isect.material.param = m.param;
        isect.material.albedo = texture_sample(m.albedo, isect.u, isect.v, cn);
        isect.material.emission = texture_sample(m.emission, isect.u, isect.v, cn);
        isect.material.roughness = texture_sample(m.roughness, isect.u, isect.v, cn).x;
        return true;
    }
    if (cn) cn->sky_lookups++;
    isect.material.emission = skybox_sample(s, ray.direction);
    return false;
}

// ---------------------------------------------------------------------------------------------
// material.rs
static inline bool nee_available(const PointMaterial &m) { return m.surface == HR_DIFFUSE || m.surface == HR_GGX; }  // :42-51
static inline double roughness_to_alpha2(double r) { return r * r; }                                                  // :250-255
static void tangent_basis(V3 n, V3 &tangent, V3 &binormal) {  // :202-211
    V3 up = std::fabs(n.x) > EPS ? V3(0, 1, 0) : V3(1, 0, 0);
    tangent = normalize(cross(up, n));
    binormal = cross(n, tangent);
}
static V3 importance_sample_diffuse(double r0, double r1, V3 n) {  // :227-248
    V3 t, b;
    tangent_basis(n, t, b);
    double phi = PI2 * r0;
    return (t * std::cos(phi) + b * std::sin(phi)) * std::sqrt(r1) + n * std::sqrt(1.0 - r1);
}
static V3 importance_sample_ggx_half(double r0, double r1, V3 n, double alpha2) {  // :260-269
    V3 t, b;
    tangent_basis(n, t, b);
    double phi = PI2 * r0;
    double cos_theta = std::sqrt((1.0 - r1) / (1.0 + (alpha2 - 1.0) * r1));
    double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
    V3 h(sin_theta * std::cos(phi), sin_theta * std::sin(phi), cos_theta);
    return t * h.x + b * h.y + n * h.z;
}
static inline double g_smith_joint_lambda(double x_dot_n, double alpha2) {  // :271-274
    double a = 1.0 / (x_dot_n * x_dot_n) - 1.0;
    return 0.5 * std::sqrt(1.0 + alpha2 * a) - 0.5;
}
static inline double g_smith_joint(double l_dot_n, double v_dot_n, double alpha2) {  // :276-280
    return 1.0 / (1.0 + g_smith_joint_lambda(l_dot_n, alpha2) + g_smith_joint_lambda(v_dot_n, alpha2));
}
static inline double f_schlick(double v_dot_h, double f0) {  // :282-284; powi(5) == x * (x^2)^2
    double x = 1.0 - v_dot_h, x2 = x * x, x4 = x2 * x2;
    return f0 + (1.0 - f0) * (x * x4);
}
static double material_bsdf(const PointMaterial &m, V3 view, V3 normal, V3 light) {  // :53-89
    if (m.surface == HR_DIFFUSE) return 1.0 / PI;
    if (m.surface == HR_GGX) {
        double alpha2 = roughness_to_alpha2(m.roughness);
        V3 half = normalize(light + view);
        double l_dot_n = dot(light, normal);
        if (sign_negative(l_dot_n)) return 0.0;
        double v_dot_n = dot(view, normal), v_dot_h = dot(view, half), h_dot_n = dot(half, normal);
        double tmp = 1.0 - (1.0 - alpha2) * h_dot_n * h_dot_n;
        double d = alpha2 / (PI * tmp * tmp);
        double g = g_smith_joint(l_dot_n, v_dot_n, alpha2);
        double f = f_schlick(v_dot_h, m.param);
        return d * g * f / (4.0 * l_dot_n * v_dot_n);
    }
    return 0.0;  // unimplemented!() in the reference; unreachable because nee_available() gates the call
}
struct SampleResult { Ray ray; double reflectance; bool transmitted = false; };   // transmitted: bookkeeping for the path log only
static bool sample_refraction(double r0, V3 position, V3 view, V3 normal, double ior, SampleResult &out) {  // :154-199
    bool is_incoming = sign_negative(dot(view, normal));
    V3 oriented_normal = is_incoming ? normal : -normal;
    double nnt = is_incoming ? 1.0 / ior : ior;
    V3 reflect_direction = reflect(view, oriented_normal);
    V3 refract_direction = refract(view, oriented_normal, nnt);
    if (refract_direction == V3()) {
        out.ray.origin = position + OFFSET * oriented_normal;
        out.ray.direction = reflect_direction;
        out.reflectance = 1.0;
        return true;
    }
    double cos_i = dot(view, -oriented_normal);
    double cos_t = dot(refract_direction, -oriented_normal);
    double r_s = (nnt * cos_i - cos_t) * (nnt * cos_i - cos_t) / ((nnt * cos_i + cos_t) * (nnt * cos_i + cos_t));
    double r_p = (nnt * cos_t - cos_i) * (nnt * cos_t - cos_i) / ((nnt * cos_t + cos_i) * (nnt * cos_t + cos_i));
    double fr = 0.5 * (r_s + r_p);
    if (r0 <= fr) {
        out.ray.origin = position + OFFSET * oriented_normal;
        out.ray.direction = reflect_direction;
        out.reflectance = 1.0;
    } else {
        out.ray.origin = position - OFFSET * oriented_normal;
        out.ray.direction = refract_direction;
        out.reflectance = nnt * nnt;
        out.transmitted = true;
    }
    return true;
}
static bool material_sample(const PointMaterial &m, double r0, double r1, V3 position, V3 view, V3 normal, SampleResult &out) {  // :91-151
    V3 ray = -view;
    switch (m.surface) {
        case HR_DIFFUSE:
            out.ray.origin = position + normal * OFFSET;
            out.ray.direction = importance_sample_diffuse(r0, r1, normal);
            out.reflectance = 1.0;
            return true;
        case HR_SPECULAR:
            out.ray.origin = position + normal * OFFSET;
            out.ray.direction = reflect(ray, normal);
            out.reflectance = 1.0;
            return true;
        case HR_REFRACTION: return sample_refraction(r0, position, ray, normal, m.param, out);
        case HR_GGX: {
            double alpha2 = roughness_to_alpha2(m.roughness);
            V3 half = importance_sample_ggx_half(r0, r1, normal, alpha2);
            V3 next_direction = reflect(ray, half);
            double l_dot_n = dot(next_direction, normal);
            if (sign_negative(l_dot_n)) return false;
            double v_dot_n = dot(view, normal), v_dot_h = dot(view, half), h_dot_n = dot(half, normal);
            double g = g_smith_joint(l_dot_n, v_dot_n, alpha2);
            double f = f_schlick(v_dot_h, m.param);
            out.ray.origin = position + normal * OFFSET;
            out.ray.direction = next_direction;
            out.reflectance = f * saturate(g * v_dot_h / (h_dot_n * v_dot_n));
            return true;
        }
        default: {  // GGXRefraction
            double alpha2 = roughness_to_alpha2(m.roughness);
            V3 half = importance_sample_ggx_half(r0, r1, normal, alpha2);
            return sample_refraction(r0, position, ray, half, m.param, out);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// camera.rs:66-96
static Ray ray_with_dof(const hr_camera &c, double ncx, double ncy, Isaac64 &rng) {
    double sqx, sqy;
    for (;;) {
        double u = rng.next_f64();
        double v = rng.next_f64();
        sqx = 2.0 * u - 1.0; sqy = 2.0 * v - 1.0;
        if (c.lens_shape == 0) break;
        if (sqx * sqx + sqy * sqy < 1.0) break;
    }
    double lux = sqx * c.lens_radius, luy = sqy * c.lens_radius;
    V3 lens_pos = V3(c.right) * lux + V3(c.up) * luy;
    Ray r;
    r.origin = V3(c.eye) + lens_pos;
    r.direction = normalize(ncx * V3(c.plane_half_right) + ncy * V3(c.plane_half_up) + c.focus_distance * V3(c.forward) - lens_pos);
    return r;
}

// scene.rs:92-101
struct Surface { V3 position, normal; double pdf; };
static Surface sample_on_surface(const Element &e, double r0, double r1) {
    double theta = PI2 * r0;
    double unit_z = 1.0 - 2.0 * r1;
    double a = std::sqrt(1.0 - unit_z * unit_z);
    Surface s;
    s.normal = V3(a * std::cos(theta), a * std::sin(theta), unit_z);
    s.position = e.center + (e.radius + OFFSET) * s.normal;
    s.pdf = 1.0 / (4.0 * PI * e.radius * e.radius);
    return s;
}

// renderer.rs:269-296
static V3 next_event_estimation(const Scene &s, double r0, double r1, V3 position, V3 view, V3 normal, const PointMaterial &material, Counters *cn,
                                uint32_t *visible_mask = nullptr) {
    V3 accumulation;
    size_t k_em = 0;
    for (size_t ei : s.emissions) {
        const size_t k = k_em++;
        Surface surface = sample_on_surface(s.elements[ei], r0, r1);
        V3 shadow_vec = surface.position - position;
        V3 shadow_dir = normalize(shadow_vec);
        Ray shadow_ray{position, shadow_dir};
        Intersection si;
        if (cn) cn->rays_shadow++;
        bool shadow_hit = scene_intersect(s, shadow_ray, si, nullptr, cn);
        if (shadow_hit && approximately(si.position, surface.position)) {
            double dot_0 = std::fabs(dot(normal, shadow_dir));
            double dot_l = std::fabs(dot(surface.normal, shadow_dir));
            double distance_pow2 = dot(shadow_vec, shadow_vec);
            double g = (dot_0 * dot_l) / distance_pow2;
            double pdf = surface.pdf;
            accumulation = accumulation + si.material.emission * material_bsdf(material, view, normal, shadow_dir) * g / pdf;
            if (visible_mask) *visible_mask |= 16u << (k & 3u);
        }
    }
    return accumulation * material.albedo;
}

// Per-path event log — bookkeeping beside the algorithm, no influence on it.  Same encoding as the HIP path's log
// (hanamaru-renderer_amd/csrc/pt_core.h PathLog): one byte per iteration of renderer.rs:174 — bits 0-2: 0 not reached, 1 miss, 2 + surface type =
// hit and sampled, 7 = hit and PointMaterial::sample returned None; bit 3: transmitted (Refraction / GGXRefraction); bits 4-7: NEE visibility of
// emitter k in bit 4 + (k mod 4) — plus an FNV-style hash of the element indices hit and the number of scene.intersect calls.
// Every other discrete decision of a main ray is there too: the hash takes the face of a cuboid hit (scene.rs:160-182) and the cube-map
// face of the sky lookup that ends a path (scene.rs:295-319); quad_sum is the 16-bit sum over the texel quads (texture.rs:30-33's x1, y1) of
// every image a main ray sampled — surface textures and sky (word 3, bits 16-31).
struct PathLog {
    uint8_t ev[9]; uint32_t hash, rays, sphere_hits, quad_sum;   // sphere_hits: main rays that hit a sphere (word 3, bits 8-15)
    PathLog() { memset(ev, 0, sizeof ev); hash = 0x811c9dc5u; rays = 0; sphere_hits = 0; quad_sum = 0; }
};
static inline int cuboid_face_of(V3 n) { return n.y > 0.0 ? 0 : n.y < 0.0 ? 1 : n.x < 0.0 ? 2 : n.x > 0.0 ? 3 : n.z < 0.0 ? 4 : n.z > 0.0 ? 5 : 6; }
// renderer.rs:163-203
static V3 calc_pixel(const Scene &s, double ncx, double ncy, uint32_t sampling, Counters *cn, PathLog *lg = nullptr) {
    uint64_t seed[4] = {8700304ULL, (uint64_t)sampling, f64_as_usize((4.0 + ncx) * 100870.0), f64_as_usize((4.0 + ncy) * 100304.0)};
    Isaac64 rng;
    rng.from_seed(seed, 4);
    Ray ray = ray_with_dof(s.camera, ncx, ncy, rng);
    V3 accumulation, reflectance(1, 1, 1);
    uint64_t rays0 = cn ? cn->rays_primary + cn->rays_bounce + cn->rays_shadow : 0;
    for (uint32_t it = 1; it < PATHTRACING_BOUNCE_LIMIT; it++) {
        double r0 = rng.next_f64();
        double r1 = rng.next_f64();
        Intersection isect;
        if (cn) { if (it == 1) cn->rays_primary++; else cn->rays_bounce++; }
        long element = -1;
        if (lg) t_quad_sum = &lg->quad_sum;     // the main ray's texture lookups (surface textures, or the sky at a miss) are logged
        bool hit = scene_intersect(s, ray, isect, &element, cn);
        t_quad_sum = nullptr;
        if (lg) {
            lg->rays++;
            if (!hit) { lg->ev[it - 1] = 1; lg->hash = (lg->hash ^ (uint32_t)(0x2000 + skybox_face(ray.direction) + 1)) * 0x01000193u; }
        }
        static const bool verbose = getenv("HR_ORACLE_VERBOSE") != nullptr;   // debugging aid (read once): one line per ray of every path traced
        if (verbose) fprintf(stderr, "ORC it %u hit %d t %.9g o %.9g %.9g %.9g d %.9g %.9g %.9g accum %.9g %.9g %.9g refl %.9g %.9g %.9g n %.9g %.9g %.9g\n", it, (int)hit, isect.distance, ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z, accumulation.x, accumulation.y, accumulation.z, reflectance.x, reflectance.y, reflectance.z, isect.normal.x, isect.normal.y, isect.normal.z);
        double current_reflectance = 1.0;
        if (hit) {
            if (cn) cn->surface_hits++;
            V3 view = -ray.direction;
            SampleResult result;
            if (lg) {
                if (s.elements[element].kind == HR_CUBOID) lg->hash = (lg->hash ^ (uint32_t)(0x1000 + cuboid_face_of(isect.normal) + 1)) * 0x01000193u;
                lg->hash = (lg->hash ^ (uint32_t)(element + 1)) * 0x01000193u;
                if (s.elements[element].kind == HR_SPHERE) lg->sphere_hits++;
                if (s.elements[element].kind == HR_MESH)
                    lg->hash = (lg->hash ^ ((uint32_t)(s.elements[element].face_base + (size_t)isect.face) + 0x9e3779b9u + 1u)) * 0x01000193u;
            }
            if (material_sample(isect.material, r0, r1, isect.position, view, isect.normal, result)) {
                uint32_t visible = 0;
                if (nee_available(isect.material)) {
                    accumulation = accumulation + reflectance * next_event_estimation(s, r0, r1, result.ray.origin, view, isect.normal, isect.material, cn, &visible);
                    if (lg) lg->rays += (uint32_t)s.emissions.size();
                }
                if (lg) lg->ev[it - 1] = (uint8_t)((2u + (uint32_t)isect.material.surface) | (result.transmitted ? 8u : 0u) | visible);
                ray = result.ray;
                current_reflectance = result.reflectance;
            } else {
                if (lg) lg->ev[it - 1] = 7;
                break;
            }
        }
        accumulation = accumulation + reflectance * isect.material.emission;
        reflectance = reflectance * (isect.material.albedo * current_reflectance);
        if (!hit || reflectance == V3()) break;
    }
    if (cn) {
        cn->paths++;
        cn->draws += rng.draws;
        uint64_t nr = cn->rays_primary + cn->rays_bounce + cn->rays_shadow - rays0;
        cn->rays_per_path_hist[nr < 24 ? nr : 23]++;
        cn->draws_per_path_hist[rng.draws < 40 ? rng.draws : 39]++;
    }
    return accumulation;
}

static inline void normalized_coord(uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t sx, uint32_t sy, double &ncx, double &ncy) {
    // renderer.rs:34-36, 53-54
    double fx = (double)x, fy = (double)(H - y);
    double ox = (double)sx / (double)SUPERSAMPLING - 0.5, oy = (double)sy / (double)SUPERSAMPLING - 0.5;
    double rx = (double)W, ry = (double)H;
    double m = std::fmin(rx, ry);
    ncx = ((fx + ox) * 2.0 - rx) / m;
    ncy = ((fy + oy) * 2.0 - ry) / m;
}
// renderer.rs:48-60
static V3 supersampling(const Scene &s, uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t sampling, Counters *cn) {
    V3 acc;
    for (uint32_t sy = 0; sy < SUPERSAMPLING; sy++)
        for (uint32_t sx = 0; sx < SUPERSAMPLING; sx++) {
            double ncx, ncy;
            normalized_coord(W, H, x, y, sx, sy, ncx, ncy);
            acc = acc + calc_pixel(s, ncx, ncy, sampling, cn);
        }
    return acc;
}

static void add_counters(Counters &dst, const Counters &src) {
    const uint64_t *a = (const uint64_t *)&src;
    uint64_t *d = (uint64_t *)&dst;
    for (size_t i = 0; i < sizeof(Counters) / 8; i++) d[i] += a[i];
}

// ---------------------------------------------------------------------------------------------
// post chain — tonemap.rs:22-27, color.rs:38-48, filter.rs:7-58, color.rs:10-16, renderer.rs:64-90
static V3 reinhard(V3 color, double exposure, double white_point) {
    color = color * exposure;
    double luminance = dot(V3(0.22, 0.707, 0.071), color);  // color.rs:63-65
    white_point = white_point * exposure;
    V3 r = color * (luminance / (white_point * white_point) + 1.0) / (luminance + 1.0);
    return V3(saturate(r.x), saturate(r.y), saturate(r.z));
}
static V3 linear_to_gamma(V3 c) {
    double e = 1.0 / GAMMA_FACTOR;
    return V3(std::pow(c.x, e), std::pow(c.y, e), std::pow(c.z, e));
}
static double gaussian(double x, double sigma) { return std::exp(-(x * x) / (2.0 * sigma * sigma)) / (2.0 * PI * sigma * sigma); }
static double filter_distance(uint32_t x, uint32_t y, uint32_t i, uint32_t j) {
    uint32_t dx = x - i, dy = y - j;  // wrapping (release build)
    return std::sqrt((double)(uint32_t)(dx * dx + dy * dy));
}
static V3 bilateral(const std::vector<V3> &img, size_t current, uint32_t width, uint32_t height) {
    uint32_t x = (uint32_t)current % width, y = (uint32_t)current / width;
    const V3 &pixel = img[current];
    double current_sum = pixel.x + pixel.y + pixel.z;
    double sum_scale = 1.0 / 3.0;
    V3 filtered;
    double w_p = 0.0;
    uint32_t half = BILATERAL_FILTER_DIAMETER / 2;
    for (uint32_t i = 0; i < BILATERAL_FILTER_DIAMETER; i++)
        for (uint32_t j = 0; j < BILATERAL_FILTER_DIAMETER; j++) {
            uint32_t nx = clamp_u32(x - (half - i), 0, width - 1);
            uint32_t ny = clamp_u32(y - (half - j), 0, height - 1);
            const V3 &nb = img[(size_t)ny * width + nx];
            double nsum = nb.x + nb.y + nb.z;
            double g_i = gaussian(sum_scale * (nsum - current_sum), BILATERAL_FILTER_SIGMA_I);
            double g_s = gaussian(filter_distance(x, y, nx, ny), BILATERAL_FILTER_SIGMA_S);
            double w = g_i * g_s;
            filtered = filtered + nb * w;
            w_p += w;
        }
    return filtered / w_p;
}

}  // namespace orc

using namespace orc;

struct orc_scene { Scene s; };

ORC_API int orc_scene_create(const hr_scene_desc *d, orc_scene **out) {
    if (!d || !out) return -1;
    std::unique_ptr<orc_scene> os(new orc_scene);
    Scene &s = os->s;
    s.image_store.resize(d->num_images);
    s.images.resize(d->num_images);
    for (uint32_t i = 0; i < d->num_images; i++) {
        const hr_image &im = d->images[i];
        s.image_store[i].assign(im.rgba, im.rgba + (size_t)im.width * im.height * 4);
        s.images[i] = Image{s.image_store[i].data(), im.width, im.height};
    }
    auto tex = [&](const hr_texture &t) { return Texture{V3(t.color), t.image >= 0 ? &s.images[t.image] : nullptr}; };
    s.elements.resize(d->num_elements);
    size_t face_base = 0;
    for (uint32_t i = 0; i < d->num_elements; i++) {
        const hr_element &e = d->elements[i];
        Element &o = s.elements[i];
        o.kind = e.kind;
        o.face_base = face_base;
        if (e.kind == HR_MESH) face_base += e.num_faces;
        o.material = Material{e.material.surface, e.material.param, tex(e.material.albedo), tex(e.material.emission), tex(e.material.roughness)};
        o.center = V3(e.center); o.radius = e.radius;
        o.box = Aabb{V3(e.aabb_min), V3(e.aabb_max)};
        if (e.kind == HR_MESH) {
            o.vertexes.resize(e.num_vertexes);
            for (uint64_t k = 0; k < e.num_vertexes; k++) o.vertexes[k] = V3(e.vertexes[k]);
            o.faces.assign(e.faces, e.faces + e.num_faces * 3);
            std::vector<size_t> idx(e.num_faces);
            for (size_t k = 0; k < idx.size(); k++) idx[k] = k;
            o.bvh = build_mesh(o, idx);  // bvh.rs:203-206
        }
    }
    for (int f = 0; f < 6; f++) {
        if (d->skybox.face_image[f] < 0 || (uint32_t)d->skybox.face_image[f] >= d->num_images) return -2;
        s.sky[f] = &s.images[d->skybox.face_image[f]];
    }
    s.sky_intensity = V3(d->skybox.intensity);
    s.camera = d->camera;
    std::vector<size_t> idx(s.elements.size());
    for (size_t k = 0; k < idx.size(); k++) idx[k] = k;
    s.top = build_scene(s, idx);  // bvh.rs:208-211
    for (size_t k = 0; k < s.elements.size(); k++)  // scene.rs:356-358: nee_available (Sphere only) && emission tint != 0
        if (s.elements[k].kind == HR_SPHERE && !(s.elements[k].material.emission.color == V3())) s.emissions.push_back(k);
    *out = os.release();
    return 0;
}
ORC_API void orc_scene_destroy(orc_scene *s) { delete s; }

ORC_API size_t orc_counters_size(void) { return sizeof(Counters); }

// renderer.rs:25-46 — accumulate samplings begin, begin+stride, ... < end into acc (W*H*3 doubles, += like the reference)
ORC_API int orc_render(const orc_scene *os, uint32_t W, uint32_t H, uint32_t s_begin, uint32_t s_end, uint32_t stride, int nthreads,
                       double *acc, void *counters_out) {
    if (!os || !acc || !W || !H || !stride) return -1;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    const Scene &s = os->s;
    std::vector<Counters> cns(nthreads);
    for (auto &c : cns) memset(&c, 0, sizeof c);
    // renderer.rs:32-43: one parallel pass over the pixels per sampling (rayon work stealing), one barrier between samplings.
    // Here: a persistent pool, chunks of 16 pixels handed out by an atomic counter (row-sized units would leave most of a
    // 256-thread host idle on small images), a barrier per sampling.
    const uint32_t CHUNK = 16;
    const uint64_t pixels = (uint64_t)W * H;
    const uint32_t chunks = (uint32_t)((pixels + CHUNK - 1) / CHUNK);
    std::vector<uint32_t> samplings;
    for (uint32_t sampling = s_begin; sampling < s_end; sampling += stride) samplings.push_back(sampling);
    std::vector<std::atomic<uint32_t>> next(samplings.size());
    for (auto &n : next) n.store(0);
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0, generation = 0;
    auto barrier = [&]() {
        std::unique_lock<std::mutex> lk(mu);
        uint32_t gen = generation;
        if (++arrived == (uint32_t)nthreads) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    };
    auto worker = [&](int tid) {
        Counters *cn = counters_out ? &cns[tid] : nullptr;
        for (size_t k = 0; k < samplings.size(); k++) {
            for (;;) {
                uint32_t c = next[k].fetch_add(1);
                if (c >= chunks) break;
                uint64_t p0 = (uint64_t)c * CHUNK, p1 = std::min<uint64_t>(p0 + CHUNK, pixels);
                for (uint64_t i = p0; i < p1; i++) {
                    uint32_t x = (uint32_t)(i % W), y = (uint32_t)(i / W);
                    V3 col = supersampling(s, W, H, x, y, samplings[k], cn);
                    double *p = &acc[i * 3];
                    p[0] += col.x; p[1] += col.y; p[2] += col.z;
                }
            }
            if (nthreads > 1) barrier();
        }
    };
    if (nthreads == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
        for (auto &t : th) t.join();
    }
    if (counters_out) {
        Counters total;
        memset(&total, 0, sizeof total);
        for (auto &c : cns) add_counters(total, c);
        memcpy(counters_out, &total, sizeof total);
    }
    return 0;
}

// Same as orc_render but only for the pixel rectangle [x0, x0+rw) x [y0, y0+rh) of a W x H image (seeds depend on the
// full-image coordinates).  acc: rw*rh*3 doubles.  Lets the CPU tier compare a crop against the reference binary's
// committed 1920x1080 x 1000-sampling render.
ORC_API int orc_render_region(const orc_scene *os, uint32_t W, uint32_t H, uint32_t x0, uint32_t y0, uint32_t rw, uint32_t rh, uint32_t s_begin,
                              uint32_t s_end, uint32_t stride, int nthreads, double *acc) {
    if (!os || !acc || !W || !H || !stride || x0 + rw > W || y0 + rh > H) return -1;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    const Scene &s = os->s;
    std::atomic<uint32_t> next{0};
    const uint32_t total = rw * rh;
    auto worker = [&]() {
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= total) break;
            uint32_t x = x0 + i % rw, y = y0 + i / rw;
            V3 sum;
            for (uint32_t sampling = s_begin; sampling < s_end; sampling += stride) sum = sum + supersampling(s, W, H, x, y, sampling, nullptr);
            double *p = &acc[(size_t)i * 3];
            p[0] += sum.x; p[1] += sum.y; p[2] += sum.z;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(worker);
    for (auto &t : th) t.join();
    return 0;
}

// DebugRenderer::calc_pixel (renderer.rs:116-139) + Renderer::render's 2x2 supersampling, one sampling.
// mode: 0 Shading, 1 Normal, 2 Depth, 3 FocalPlane (renderer.rs:102-107)
ORC_API int orc_render_debug(const orc_scene *os, uint32_t W, uint32_t H, int mode, double *acc) {
    if (!os || !acc || mode < 0 || mode > 3) return -1;
    const Scene &s = os->s;
    const hr_camera &c = s.camera;
    V3 light_direction = normalize(V3(1.0, 2.0, -1.0));
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            V3 sum;
            for (uint32_t sy = 0; sy < SUPERSAMPLING; sy++)
                for (uint32_t sx = 0; sx < SUPERSAMPLING; sx++) {
                    double ncx, ncy;
                    normalized_coord(W, H, x, y, sx, sy, ncx, ncy);
                    Ray ray;  // camera.rs:98-107 (pinhole)
                    ray.origin = V3(c.eye);
                    ray.direction = normalize(ncx * V3(c.plane_half_right) + ncy * V3(c.plane_half_up) + c.focus_distance * V3(c.forward));
                    Intersection isect;
                    bool hit = scene_intersect(s, ray, isect, nullptr, nullptr);
                    V3 col;
                    if (hit) {
                        if (mode == 0) {
                            Ray shadow_ray{isect.position + isect.normal * OFFSET, light_direction};
                            Intersection si;
                            bool shadow_hit = scene_intersect(s, shadow_ray, si, nullptr, nullptr);
                            double shadow = shadow_hit ? 0.5 : 1.0;
                            double diffuse = std::fmax(dot(isect.normal, light_direction), 0.0);
                            col = isect.material.emission + isect.material.albedo * diffuse * shadow;
                        } else if (mode == 1) col = isect.normal;
                        else if (mode == 2) { double v = 0.5 * isect.distance / c.focus_distance; col = V3(v, v, v); }
                        else { double v = std::fabs(isect.distance - c.focus_distance); col = V3(v, v, v); }
                    } else {
                        col = isect.material.emission;
                    }
                    sum = sum + col;
                }
            double *p = &acc[((size_t)y * W + x) * 3];
            p[0] += sum.x; p[1] += sum.y; p[2] += sum.z;
        }
    return 0;
}

// one calc_pixel (renderer.rs:163) for pixel (x,y) sub-sample (sx,sy)
ORC_API int orc_calc_pixel(const orc_scene *os, uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t sx, uint32_t sy, uint32_t sampling,
                           double *rgb) {
    double ncx, ncy;
    normalized_coord(W, H, x, y, sx, sy, ncx, ncy);
    V3 c = calc_pixel(os->s, ncx, ncy, sampling, nullptr);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    return 0;
}

// every path of ONE sampling with its event log (PathLog above): radiance[(y*W + x)*4 + sy*2 + sx][3] (f64), and per path rays, 9 event
// bytes (padded to 12) and the element hash as words[.][5] = {rays, ev 0-3, ev 4-7, ev 8, hash} — the layout of hr_debug_path_log minus the radiance
ORC_API int orc_path_log(const orc_scene *os, uint32_t W, uint32_t H, uint32_t sampling, int nthreads, double *radiance, uint32_t *words) {
    if (!os || !radiance || !words || !W || !H) return -1;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    const Scene &s = os->s;
    std::atomic<uint32_t> next{0};
    const uint32_t total = W * H;
    auto worker = [&]() {
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= total) break;
            const uint32_t x = i % W, y = i / W;
            for (uint32_t sy = 0; sy < SUPERSAMPLING; sy++)
                for (uint32_t sx = 0; sx < SUPERSAMPLING; sx++) {
                    double ncx, ncy;
                    normalized_coord(W, H, x, y, sx, sy, ncx, ncy);
                    PathLog lg;
                    const V3 c = calc_pixel(s, ncx, ncy, sampling, nullptr, &lg);
                    const size_t p = (size_t)i * 4 + sy * 2 + sx;
                    radiance[p * 3] = c.x; radiance[p * 3 + 1] = c.y; radiance[p * 3 + 2] = c.z;
                    uint32_t *w = words + p * 5;
                    w[0] = lg.rays;
                    w[1] = lg.ev[0] | (uint32_t)lg.ev[1] << 8 | (uint32_t)lg.ev[2] << 16 | (uint32_t)lg.ev[3] << 24;
                    w[2] = lg.ev[4] | (uint32_t)lg.ev[5] << 8 | (uint32_t)lg.ev[6] << 16 | (uint32_t)lg.ev[7] << 24;
                    w[3] = lg.ev[8] | (lg.sphere_hits & 0xffu) << 8 | (lg.quad_sum & 0xffffu) << 16;
                    w[4] = lg.hash;
                }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(worker);
    for (auto &t : th) t.join();
    return 0;
}

// first `count` next_u64 of the per-path generator (renderer.rs:165-168)
ORC_API int orc_path_draws(uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t sx, uint32_t sy, uint32_t sampling, uint64_t *out,
                           int count) {
    double ncx, ncy;
    normalized_coord(W, H, x, y, sx, sy, ncx, ncy);
    uint64_t seed[4] = {8700304ULL, (uint64_t)sampling, f64_as_usize((4.0 + ncx) * 100870.0), f64_as_usize((4.0 + ncy) * 100304.0)};
    Isaac64 rng;
    rng.from_seed(seed, 4);
    for (int i = 0; i < count; i++) out[i] = rng.next_u64();
    return 0;
}
ORC_API int orc_isaac64(const uint64_t *seed, int nseed, uint64_t skip, uint64_t *out, int count) {
    Isaac64 rng;
    rng.from_seed(seed, nseed);
    for (uint64_t i = 0; i < skip; i++) rng.next_u64();
    for (int i = 0; i < count; i++) out[i] = rng.next_u64();
    return 0;
}
ORC_API double orc_u64_to_f64(uint64_t v) {
#ifdef ORC_F64_FROM_TOP53
    return (double)(v >> 11) * (1.0 / 9007199254740992.0);
#else
    uint64_t bits = 0x3FF0000000000000ULL | (v & 0x000FFFFFFFFFFFFFULL);
    double d;
    memcpy(&d, &bits, 8);
    return d - 1.0;
#endif
}

// closest hit for n rays: rays = n*6 doubles; out = n*8 doubles {hit, distance, pos xyz, normal xyz}; element index or -1
ORC_API int orc_intersect(const orc_scene *os, uint32_t n, const double *rays, double *out, int32_t *out_element) {
    for (uint32_t i = 0; i < n; i++) {
        Ray r{V3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), V3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5])};
        Intersection isect;
        long el;
        bool hit = scene_intersect(os->s, r, isect, &el, nullptr);
        double *o = &out[i * 8];
        o[0] = hit ? 1.0 : 0.0; o[1] = isect.distance;
        o[2] = isect.position.x; o[3] = isect.position.y; o[4] = isect.position.z;
        o[5] = isect.normal.x; o[6] = isect.normal.y; o[7] = isect.normal.z;
        if (out_element) out_element[i] = (int32_t)el;
    }
    return 0;
}

// closest hit WITH the material fetch of scene.rs:385-401, for the analytic unit tests: out = 19 doubles
// {hit, distance, pos xyz, normal xyz, u, v, surface, param, albedo rgb, emission rgb, roughness}
ORC_API int orc_intersect_material(const orc_scene *os, const double *ray6, double *out, int32_t *out_element) {
    Ray r{V3(ray6[0], ray6[1], ray6[2]), V3(ray6[3], ray6[4], ray6[5])};
    Intersection isect;
    long el;
    bool hit = scene_intersect(os->s, r, isect, &el, nullptr);
    out[0] = hit ? 1.0 : 0.0; out[1] = isect.distance;
    out[2] = isect.position.x; out[3] = isect.position.y; out[4] = isect.position.z;
    out[5] = isect.normal.x; out[6] = isect.normal.y; out[7] = isect.normal.z;
    out[8] = isect.u; out[9] = isect.v;
    out[10] = (double)isect.material.surface; out[11] = isect.material.param;
    out[12] = isect.material.albedo.x; out[13] = isect.material.albedo.y; out[14] = isect.material.albedo.z;
    out[15] = isect.material.emission.x; out[16] = isect.material.emission.y; out[17] = isect.material.emission.z;
    out[18] = isect.material.roughness;
    if (out_element) *out_element = (int32_t)el;
    return 0;
}
// PointMaterial::sample (material.rs:91-151) and ::bsdf (:53-89) on explicit arguments.
// sample: out = 8 doubles {some(1)/none(0), origin xyz, direction xyz, reflectance}
ORC_API int orc_material_sample(int surface, double param, double roughness, double r0, double r1, const double *position, const double *view,
                                const double *normal, double *out) {
    PointMaterial m;
    m.surface = surface; m.param = param; m.albedo = V3(1, 1, 1); m.emission = V3(); m.roughness = roughness;
    SampleResult res;
    res.ray.origin = V3(); res.ray.direction = V3(); res.reflectance = 0.0;
    bool ok = material_sample(m, r0, r1, V3(position[0], position[1], position[2]), V3(view[0], view[1], view[2]), V3(normal[0], normal[1], normal[2]), res);
    out[0] = ok ? 1.0 : 0.0;
    out[1] = res.ray.origin.x; out[2] = res.ray.origin.y; out[3] = res.ray.origin.z;
    out[4] = res.ray.direction.x; out[5] = res.ray.direction.y; out[6] = res.ray.direction.z;
    out[7] = res.reflectance;
    return 0;
}
ORC_API double orc_material_bsdf(int surface, double param, double roughness, const double *view, const double *normal, const double *light) {
    PointMaterial m;
    m.surface = surface; m.param = param; m.albedo = V3(1, 1, 1); m.emission = V3(); m.roughness = roughness;
    return material_bsdf(m, V3(view[0], view[1], view[2]), V3(normal[0], normal[1], normal[2]), V3(light[0], light[1], light[2]));
}

// material / texture lookups for unit tests: sample a skybox direction
ORC_API int orc_skybox_sample(const orc_scene *os, const double *dir, double *rgb) {
    V3 c = skybox_sample(os->s, V3(dir[0], dir[1], dir[2]));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    return 0;
}
ORC_API int orc_image_sample_bilinear(const orc_scene *os, uint32_t image, double u, double v, double *rgb) {
    if (image >= os->s.images.size()) return -1;
    V3 c = sample_bilinear(os->s.images[image], u, v);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    return 0;
}

// BVH shape statistics.  element < 0 -> top-level BVH.
// out: [0]=nodes [1]=leaves [2]=max depth [3..10]=leaf-size histogram 0..7  ; aabb: 6 doubles
static void bvh_walk(const BvhNode &n, int depth, uint64_t *out) {
    out[0]++;
    if (depth > (int)out[2]) out[2] = depth;
    if (n.leaf()) {
        out[1]++;
        size_t k = n.indexes.size();
        out[3 + (k < 7 ? k : 7)]++;
    } else {
        bvh_walk(*n.children[0], depth + 1, out);
        bvh_walk(*n.children[1], depth + 1, out);
    }
}
ORC_API int orc_bvh_stats(const orc_scene *os, int element, uint64_t *out, double *aabb) {
    const BvhNode *root = nullptr;
    if (element < 0) root = os->s.top.get();
    else if ((size_t)element < os->s.elements.size()) root = os->s.elements[element].bvh.get();
    if (!root) return -1;
    memset(out, 0, 11 * sizeof(uint64_t));
    bvh_walk(*root, 0, out);
    aabb[0] = root->aabb.min.x; aabb[1] = root->aabb.min.y; aabb[2] = root->aabb.min.z;
    aabb[3] = root->aabb.max.x; aabb[4] = root->aabb.max.y; aabb[5] = root->aabb.max.z;
    return 0;
}
// leaves of the top-level BVH in DFS order, -1 separated
ORC_API int orc_top_leaves(const orc_scene *os, int32_t *out, int cap) {
    int n = 0;
    struct W { static void go(const BvhNode &nd, int32_t *out, int cap, int &n) {
        if (nd.leaf()) { for (size_t i : nd.indexes) if (n < cap) out[n++] = (int32_t)i; if (n < cap) out[n++] = -1; }
        else { go(*nd.children[0], out, cap, n); go(*nd.children[1], out, cap, n); }
    } };
    W::go(*os->s.top, out, cap, n);
    return n;
}
ORC_API int orc_num_emissions(const orc_scene *os) { return (int)os->s.emissions.size(); }

// renderer.rs:64-90.  stage_out (optional, W*H*3 doubles) receives the tone-mapped + gamma image before the bilateral pass.
ORC_API int orc_resolve(const double *acc, uint32_t W, uint32_t H, uint32_t sampling, uint8_t *rgb8, double *stage_out) {
    if (!acc || !rgb8 || !W || !H || !sampling) return -1;
    double scale = 1.0 / (double)(sampling * SUPERSAMPLING * SUPERSAMPLING);
    size_t n = (size_t)W * H;
    std::vector<V3> tmp(n);
    for (size_t i = 0; i < n; i++) {
        V3 hdr = V3(acc[i * 3], acc[i * 3 + 1], acc[i * 3 + 2]) * scale;
        V3 ldr = reinhard(hdr, TONE_MAPPING_EXPOSURE, TONE_MAPPING_WHITE_POINT);  // tonemap.rs:11-16
        tmp[i] = linear_to_gamma(ldr);
    }
    if (stage_out)
        for (size_t i = 0; i < n; i++) { stage_out[i * 3] = tmp[i].x; stage_out[i * 3 + 1] = tmp[i].y; stage_out[i * 3 + 2] = tmp[i].z; }
    for (uint32_t it = 0; it < BILATERAL_FILTER_ITERATION; it++) {
        std::vector<V3> next(n);
        for (size_t i = 0; i < n; i++) next[i] = bilateral(tmp, i, W, H);
        tmp.swap(next);
    }
    for (size_t i = 0; i < n; i++) {  // color.rs:10-16: (255 * saturate(c)) as u8 — truncation
        rgb8[i * 3] = (uint8_t)(255.0 * saturate(tmp[i].x));
        rgb8[i * 3 + 1] = (uint8_t)(255.0 * saturate(tmp[i].y));
        rgb8[i * 3 + 2] = (uint8_t)(255.0 * saturate(tmp[i].z));
    }
    return 0;
}
