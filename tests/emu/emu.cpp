// Host emulation of the HIP kernels' per-lane code — TEST INFRASTRUCTURE ONLY.
//
// csrc/pt_core.h, isaac_core.h and post_core.h are __host__ __device__ headers; this file compiles the
// very same functions with g++ and drives them one lane at a time, so the CPU-only test tier can check
// the fp32 path logic, the threaded-BVH traversal and the seed kernel's draw selection against the f64
// oracle without a GPU.  It is NOT a fallback: nothing in hanamaru-renderer_amd/ loads this library,
// and the wave-level machinery (ballot refill, LDS layout, streams) only exists in hr_api.hip.
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "flatten.h"
#include "isaac_core.h"
#include "post_core.h"
#include "pt_core.h"

using namespace hr;

struct emu_scene { HostScene hs; Scene view; };

struct ArrMem { u64 m[256]; u64 ld(int i) const { return m[i]; } void st(int i, u64 v) { m[i] = v; } };
struct ArrStore { u64 t[ISAAC_TAIL]; u64 ld(int k) const { return t[k]; } void st(int k, u64 v) { t[k] = v; } };

// what the seed kernel produces for one path: raw tail + accepted lens attempt
static bool path_tail(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int lens_shape, ArrStore &st, uint32_t &lens_a) {
    static const IsaacWarm warm = isaac_warm();
    ArrMem mem;
    RawLensTail<ArrStore> lt(st, lens_shape);
    u64 s, t;
    path_seed_words(W, H, px, py, sub, s, t);
    isaac_seed_round(mem, warm, 8700304ULL, (u64)sampling, s, t, lt);
    lt.lens_slow();
    bool ok = lt.in_window();
    lens_a = ok ? (uint32_t)lt.accepted : 0u;
    return ok;
}
static bool path_draws(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int lens_shape, float *out20) {
    ArrStore st;
    uint32_t a;
    bool ok = path_tail(W, H, px, py, sub, sampling, lens_shape, st, a);
    out20[0] = draw_lens_f32(st.t[2 * a]);
    out20[1] = draw_lens_f32(st.t[2 * a + 1]);
    for (int d = 2; d < DRAWS_PER_PATH; d++) out20[d] = draw_f32(st.t[2 * a + d]);
    return ok;
}

extern "C" {

static int g_max_leaf = 4;
static double g_split_ratio = 0.0;
void emu_set_build_options(int max_leaf, double split_ratio) { g_max_leaf = max_leaf; g_split_ratio = split_ratio; }

int emu_scene_create(const hr_scene_desc *sd, emu_scene **out) {
    emu_scene *e = new emu_scene;
    std::string err;
    int rc = flatten_scene(sd, e->hs, err, g_max_leaf, g_split_ratio);
    if (rc) { fprintf(stderr, "emu: %s\n", err.c_str()); delete e; return rc; }
    e->view = e->hs.view();
    *out = e;
    return 0;
}
void emu_scene_destroy(emu_scene *e) { delete e; }

// stats: [0]=nodes [1]=leaves [2]=max depth [3]=tris [4]=spheres [5]=cuboids [6]=emitters
void emu_scene_stats(const emu_scene *e, uint64_t *out) {
    out[0] = e->hs.num_nodes; out[1] = e->hs.bvh_leaves; out[2] = e->hs.bvh_max_depth;
    out[3] = e->hs.tris.size(); out[4] = e->hs.spheres.size(); out[5] = e->hs.cuboids.size() / 2; out[6] = e->hs.emitters.size();
}

int emu_path_draws(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int lens_shape, float *out20) {
    return path_draws(W, H, px, py, sub, sampling, lens_shape, out20) ? 0 : 1;
}

// raw tail: out[k] = k-th next_u64, k < window <= ISAAC_TAIL
struct RawTail { uint64_t *out; int window; void put(int step, u64 v) { int k = 255 - step; if (k < window) out[k] = v; } };
int emu_raw_draws(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int window, uint64_t *out) {
    static const IsaacWarm warm = isaac_warm();
    ArrMem mem;
    RawTail rt{out, window};
    u64 s, t;
    path_seed_words(W, H, px, py, sub, s, t);
    isaac_seed_round(mem, warm, 8700304ULL, (u64)sampling, s, t, rt);
    return 0;
}

// same as emu_raw_draws, through the split path (isaac_init_final into a staging array, then isaac_round)
int emu_raw_draws_split(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int window, uint64_t *out) {
    static const IsaacWarm warm = isaac_warm();
    ArrMem stage, mem;
    u64 s, t;
    path_seed_words(W, H, px, py, sub, s, t);
    isaac_init_final(stage, warm, 8700304ULL, (u64)sampling, s, t);
    for (int i = 0; i < 256; i++) mem.st(i, stage.ld(i));
    RawTail rt{out, window};
    isaac_round(mem, rt);
    return 0;
}

// counters: paths, rays, node_tests, tri_tests, sphere_tests, cuboid_tests
int emu_render(const emu_scene *e, uint32_t W, uint32_t H, uint32_t s_begin, uint32_t s_end, uint32_t stride, int nthreads, float *acc,
               uint64_t *counters) {
    const Scene &sc = e->view;
    RenderParams rp{};
    rp.width = W; rp.height = H;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    std::vector<std::vector<uint64_t>> cn(nthreads, std::vector<uint64_t>(6, 0));
    for (uint32_t sampling = s_begin; sampling < s_end; sampling += stride) {
        std::atomic<uint32_t> next{0};
        auto work = [&](int tid) {
            std::vector<u64> draws((size_t)ISAAC_TAIL * 64);
            for (;;) {
                uint32_t y = next.fetch_add(1);
                if (y >= H) break;
                for (uint32_t x = 0; x < W; x++) {
                    float sum[3] = {0, 0, 0};
                    for (uint32_t sub = 0; sub < 4; sub++) {
                        ArrStore st;
                        Path p;
                        p.q = 0; p.draw_base = 0;
                        path_tail(W, H, x, y, sub, sampling, sc.cam.lens_shape, st, p.lens_a);
                        for (int d = 0; d < ISAAC_TAIL; d++) draws[(size_t)d * 64] = st.t[d];
                        path_start(sc, rp, p, x, y, sub, draws.data());
                        LaneCounters lc = {0, 0, 0, 0, 0};
                        for (;;) {
                            while (p.ts.cur != NODE_END) trace_step<true>(sc, p.ray, p.ts, &lc);
                            if (path_advance<true>(sc, p, draws.data(), &lc)) break;
                        }
                        sum[0] += p.accum.x; sum[1] += p.accum.y; sum[2] += p.accum.z;
                        cn[tid][0]++; cn[tid][1] += lc.rays; cn[tid][2] += lc.node_tests; cn[tid][3] += lc.tri_tests;
                        cn[tid][4] += lc.sphere_tests; cn[tid][5] += lc.cuboid_tests;
                    }
                    float *o = &acc[((size_t)y * W + x) * 3];
                    o[0] += sum[0]; o[1] += sum[1]; o[2] += sum[2];
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
        for (auto &t : th) t.join();
    }
    if (counters)
        for (int k = 0; k < 6; k++) { counters[k] = 0; for (auto &c : cn) counters[k] += c[k]; }
    return 0;
}

int emu_render_debug(const emu_scene *e, uint32_t W, uint32_t H, int mode, float *acc) {
    RenderParams rp{};
    rp.width = W; rp.height = H;
    LaneCounters lc;
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            V3f sum = v3(0, 0, 0);
            for (uint32_t sub = 0; sub < 4; sub++) sum = sum + debug_pixel<false>(e->view, rp, x, y, sub, mode, &lc);
            float *o = &acc[((size_t)y * W + x) * 3];
            o[0] += sum.x; o[1] += sum.y; o[2] += sum.z;
        }
    return 0;
}

int emu_intersect(const emu_scene *e, uint32_t n, const float *rays, float *out, int32_t *out_elem) {
    const Scene &sc = e->view;
    for (uint32_t i = 0; i < n; i++) {
        Ray r;
        ray_set(r, v3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), v3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]));
        TraceState ts;
        trace_begin(ts, T_INF);
        LaneCounters lc;
        while (ts.cur != NODE_END) trace_step<false>(sc, r, ts, &lc);
        float *o = out + (size_t)i * 8;
        int32_t elem = -1;
        if (ts.prim >= 0) {
            Surf s;
            hit_surface(sc, r, ts, true, s);
            elem = s.elem;
            o[0] = 1.0f; o[1] = ts.t; o[2] = s.pos.x; o[3] = s.pos.y; o[4] = s.pos.z; o[5] = s.n.x; o[6] = s.n.y; o[7] = s.n.z;
        } else {
            o[0] = 0.0f; o[1] = ts.t;
            for (int k = 2; k < 8; k++) o[k] = 0.0f;
        }
        out_elem[i] = elem;
    }
    return 0;
}

int emu_resolve(const float *acc, uint32_t W, uint32_t H, uint32_t samplings, uint8_t *rgb8) {
    std::vector<float> tmp((size_t)W * H * 3);
    float scale = 1.0f / (float)(samplings * 4u);
    for (size_t i = 0; i < (size_t)W * H; i++) tonemap_gamma(acc[i * 3], acc[i * 3 + 1], acc[i * 3 + 2], scale, &tmp[i * 3]);
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) bilateral_quantise(tmp.data(), W, H, x, y, &rgb8[((size_t)y * W + x) * 3]);
    return 0;
}

}
