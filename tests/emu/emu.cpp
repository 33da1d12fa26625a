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
#include <algorithm>

#include "flatten.h"
#include "isaac_core.h"
#include "lbvh_core.h"
#include "post_core.h"
#include <cstdlib>
#include "pt_core.h"
#include "wf_core.h"

using namespace hr;

struct emu_scene { HostScene hs; Scene view; };

struct ArrMem {
    u64 m[256];
    u64 ld(int i) const { return m[i]; }
    uint32_t off(int i) const { return (uint32_t)i; }
    u64 ldo(uint32_t o) const { return m[o]; }
    void st(int i, u64 v) { m[i] = v; }
    void st8(int i, u64 A, u64 B, u64 C, u64 D, u64 E, u64 F, u64 G, u64 H) { m[i] = A; m[i + 1] = B; m[i + 2] = C; m[i + 3] = D; m[i + 4] = E; m[i + 5] = F; m[i + 6] = G; m[i + 7] = H; }
};
// one hand-off record (device_scene.h) and a window of raw outputs for the fix-up path
static int g_draw_residuals = 1;   // emu_set_draw_residuals: 0 = precise shading on the fp32 draws alone (the ablation)
struct ArrRec {   // one lane (base = 4 * lane) of a one-item block in the device layout [quad][64 lanes][4], and its twin with the draws' residuals
    float f[2 * REC_ITEM_FLOATS];
    uint32_t base = 0;
    void st4(int slot, float a, float b, float c, float d) { float *q = f + rec_slot(base, (uint32_t)slot); q[0] = a; q[1] = b; q[2] = c; q[3] = d; }
    void st4lo(int slot, float a, float b, float c, float d) { float *q = f + REC_ITEM_FLOATS + rec_slot(base, (uint32_t)slot); q[0] = a; q[1] = b; q[2] = c; q[3] = d; }
    float at(int slot) const { return f[rec_slot(base, (uint32_t)slot)]; }
};
// the path of pixel (x, y), sub-sample `sub` as the kernel addresses it: tile, lane of the tile (Path::q bits 0-5), record lane base
static void emu_place_path(RenderParams &rp, uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t sub, Path &p, ArrRec &rec) {
    rp.width = W; rp.height = H; rp.tiles_x = (W + 3) / 4; rp.tiles_y = (H + 3) / 4;
    rp.rec_lo_off = g_draw_residuals ? REC_ITEM_FLOATS : 0u;   // precise shading: as hr_api.hip sets it for a precise launch
    p.tile = (y / 4) * rp.tiles_x + x / 4;
    p.q = ((y % 4) * 4 + x % 4) * 4 + sub;
    rec.base = p.q * 4u;
}
struct ArrWindow { u64 t[ISAAC_TAIL]; void put(int step, u64 v) { t[255 - step] = v; } u64 ld(int k) const { return t[k]; } };

// what the seed kernel (+ the fix-up kernel for the paths it queues) produces for one path: the record
static bool path_record(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int lens_shape, ArrRec &rec, bool *fixed = nullptr) {
    static const IsaacWarm warm = isaac_warm();
    u64 s, t;
    path_seed_words(W, H, px, py, sub, s, t);
    {
        ArrMem mem;
        RecordTail<ArrRec, true> lt(rec, lens_shape);
        isaac_seed_round<REC_DRAWS>(mem, warm, 8700304ULL, (u64)sampling, s, t, lt);
        lt.finish();
        if (fixed) *fixed = lt.overflow();
        if (!lt.overflow()) return true;
    }
    ArrMem mem;
    ArrWindow win;
    isaac_seed_round<ISAAC_TAIL>(mem, warm, 8700304ULL, (u64)sampling, s, t, win);
    return record_from_window<true>(win, ISAAC_TAIL, lens_shape, rec);
}
static bool path_draws(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int lens_shape, float *out20) {
    ArrRec rec;
    bool ok = path_record(W, H, px, py, sub, sampling, lens_shape, rec);
    uint32_t a = float_as_uint(rec.at(REC_HEAD));
    out20[0] = rec.at(REC_HEAD + 1);
    out20[1] = rec.at(REC_HEAD + 2);
    for (int d = 2; d < DRAWS_PER_PATH; d++) out20[d] = rec.at(2 * (int)a + d);
    return ok;
}

struct RawTailPc { uint64_t *out; int window; void put(int step, u64 v) { int k = 255 - step; if (k < window) out[k] = v; } };
// same again through the producer / consumer split (isaac_init_front stores blocks < SPLIT + the 16 registers the sweep
// continues from, isaac_init_back does the rest), then isaac_round
struct TailSink {
    u64 m[256]; u64 end[16];
    void st8(int i, u64 A, u64 B, u64 C, u64 D, u64 E, u64 F, u64 G, u64 H) { m[i] = A; m[i + 1] = B; m[i + 2] = C; m[i + 3] = D; m[i + 4] = E; m[i + 5] = F; m[i + 6] = G; m[i + 7] = H; }
    void end2(int j, u64 v0, u64 v1) { end[j] = v0; end[j + 1] = v1; }
};
template <int HEAD>
static void raw_draws_pc(u64 s, u64 t, uint32_t sampling, int window, uint64_t *out) {
    static const IsaacWarm warm = isaac_warm();
    TailSink sink;
    for (int i = 0; i < 256; i++) sink.m[i] = 0xdeadbeefdeadbeefULL;
    isaac_init_front<HEAD>(sink, warm, 8700304ULL, (u64)sampling, s, t);
    ArrMem mem;
    for (int i = 0; i < 256; i++) mem.st(i, sink.m[i]);
    isaac_init_back<HEAD>(mem, sink.end);
    RawTailPc rt{out, window};
    isaac_round<ISAAC_TAIL>(mem, rt);
}

// the three-run form of the seed_seg_kernel: isaac_init_ahead hands out the sweep's registers at blocks 0, B1, B2; three runs of NBLK
// blocks (the last two overlapping by one block when 3 * NBLK > 32, as in the kernel) fill the state; then isaac_round
struct StateSink {
    u64 st[3][16];
    void state(int k, u64 a, u64 b, u64 c, u64 d, u64 e, u64 f, u64 g, u64 h, u64 A, u64 B, u64 C, u64 D, u64 E, u64 F, u64 G, u64 H) {
        u64 v[16] = {a, b, c, d, e, f, g, h, A, B, C, D, E, F, G, H};
        for (int i = 0; i < 16; i++) st[k][i] = v[i];
    }
};
struct ShiftMem { u64 *m; void st(int i, u64 v) { m[i] = v; } };
template <int B1, int B2, int NBLK>
static void raw_draws_seg(u64 s, u64 t, uint32_t sampling, int window, uint64_t *out) {
    static_assert(B1 <= NBLK && B2 <= B1 + NBLK && 32 <= B2 + NBLK, "runs cover the sweep");
    static const IsaacWarm warm = isaac_warm();
    StateSink sink;
    isaac_init_ahead<B1, B2>(sink, warm, 8700304ULL, (u64)sampling, s, t);
    ArrMem mem;
    for (int i = 0; i < 256; i++) mem.st(i, 0xdeadbeefdeadbeefULL);
    u64 stage[256 + 8 * NBLK];
    const int first[3] = {0, B1, B2};
    for (int k = 2; k >= 0; k--) {   // any order: overlapping blocks get the same values from both runs
        ShiftMem sm{stage};
        isaac_init_run<NBLK>(sm, sink.st[k]);
        for (int i = 0; i < 8 * NBLK && first[k] * 8 + i < 256; i++) mem.st(first[k] * 8 + i, stage[i]);
    }
    RawTailPc rt{out, window};
    isaac_round<ISAAC_TAIL>(mem, rt);
}

extern "C" {

static int g_max_leaf = 4;
static double g_split_ratio = -1.0;   // automatic, as the library
static int g_builder = 0;   // 0 = host SAH (bvh_build.cpp), 1 = LBVH, 2 = PLOC (lbvh_core.h, the device builders' per-thread code run sequentially)
void emu_set_build_options(int max_leaf, double split_ratio) { g_max_leaf = max_leaf; g_split_ratio = split_ratio; }
void emu_set_builder(int builder) { g_builder = builder; }

// what build_bvh_on_device (hr_api.hip) does, with std::sort for the radix sort and loops for the kernels.
// builder 1 = LBVH hierarchy (Karras), 2 = PLOC (the single-workgroup kernel's phases, run by one "thread")
static void device_build_host(HostScene &hs, int max_leaf, int builder, double split_ratio) {
    using namespace lbvh;
    Prims p{};
    p.tris = hs.tris.data(); p.num_tris = (uint32_t)hs.tris.size();
    // early split clipping as build_bvh_on_device does it: count, scan, emit
    std::vector<uint32_t> ref_tri;
    std::vector<float> ref_box;
    if (split_ratio != 0.0 && !hs.tris.empty()) {
        const double e0 = hs.scene_max[0] - hs.scene_min[0], e1 = hs.scene_max[1] - hs.scene_min[1], e2 = hs.scene_max[2] - hs.scene_min[2];
        const double scene_sa = (e0 >= 0 && e1 >= 0 && e2 >= 0) ? 2.0 * (e0 * e1 + e1 * e2 + e2 * e0) : 0.0;
        SplitParams sp{split_ratio < 0 ? 2.0 : split_ratio, 1e-4 * scene_sa, SPLIT_MAX_DEPTH};
        std::vector<uint32_t> offsets(hs.tris.size() + 1, 0);
        for (size_t i = 0; i < hs.tris.size(); i++) offsets[i + 1] = offsets[i] + split_tri(hs.tris[i], sp, nullptr);
        const uint64_t refs = offsets.back();
        if (refs > hs.tris.size() && refs < MAX_PRIMS_PER_TYPE) {
            ref_tri.resize(refs); ref_box.resize(6 * refs);
            for (size_t i = 0; i < hs.tris.size(); i++) {
                uint32_t cnt = split_tri(hs.tris[i], sp, ref_box.data() + 6 * (size_t)offsets[i]);
                for (uint32_t k = 0; k < cnt; k++) ref_tri[offsets[i] + k] = (uint32_t)i;
            }
            p.ref_tri = ref_tri.data(); p.ref_box = ref_box.data(); p.num_tris = (uint32_t)refs;
        }
    }
    p.spheres = hs.spheres.data(); p.num_spheres = (uint32_t)hs.spheres.size();
    p.cuboids = hs.cuboids.data(); p.num_cuboids = (uint32_t)(hs.cuboids.size() / 2);
    for (int a = 0; a < 3; a++) {
        double ext = hs.scene_max[a] - hs.scene_min[a];
        p.smin[a] = (float)hs.scene_min[a];
        p.sinv[a] = ext > 0 ? (float)(1.0 / ext) : 0.0f;
    }
    const int n = (int)(p.num_tris + p.num_spheres + p.num_cuboids), N = 2 * n - 1;
    p.index_bits = key_index_bits_for((uint64_t)n);
    std::vector<mkey_t> keys(n);
    for (int i = 0; i < n; i++) keys[i] = prim_key(p, (uint32_t)i);
    std::sort(keys.begin(), keys.end());
    std::vector<uint32_t> parent(N, NO_PARENT), left(n), right(n), flags(n, 0), info(N, 0), size(N, 0), axis_low(n, 0), word(N, 0), prim_pos(n, 0);
    std::vector<u64t> tc(N, 0);
    std::vector<float> bmin(3 * (size_t)N), bmax(3 * (size_t)N);
    Work w{parent.data(), left.data(), right.data(), flags.data(), bmin.data(), bmax.data(), info.data(), tc.data(), size.data(), axis_low.data(), word.data()};
    for (int k = 0; k < n; k++) fit_leaf(p, keys.data(), n, k, w);
    if (builder == 1) {
        for (int i = 0; i < n - 1; i++) hierarchy_node(keys.data(), n, i, w);
    } else if (n > 1) {
        std::vector<uint32_t> cl(n), nxt(n), nn(n);
        for (int k = 0; k < n; k++) cl[k] = (uint32_t)(n - 1 + k);
        uint32_t m = (uint32_t)n, next_node = (uint32_t)(n - 1);   // internal ids are handed out downwards: the last merge makes node 0, the root
        while (m > PLOC_TOP_CLUSTERS) {   // the merges stop at a few thousand clusters ...
            for (uint32_t i = 0; i < m; i++) nn[i] = ploc_nearest(w, cl.data(), m, i);
            uint32_t pos = 0, made = 0;
            for (uint32_t i = 0; i < m; i++) {
                int role = ploc_role(nn.data(), i);
                if (role == 2) continue;
                if (role == 1) { uint32_t id = next_node - 1u - made++; ploc_make_node(w, id, cl[i], cl[nn[i]]); nxt[pos++] = id; }
                else nxt[pos++] = cl[i];
            }
            next_node -= made;
            m = pos;
            cl.swap(nxt);
        }
        if (m > 1) {   // ... and the host builder's binned SAH joins them top-down (as build_bvh_on_device does)
            std::vector<float> boxes(6 * (size_t)m);
            std::vector<uint32_t> counts(m);
            for (uint32_t i = 0; i < m; i++) {
                for (int a = 0; a < 3; a++) { boxes[6 * (size_t)i + a] = bmin[cl[i] * 3 + a]; boxes[6 * (size_t)i + 3 + a] = bmax[cl[i] * 3 + a]; }
                counts[i] = info[cl[i]] & INFO_COUNT;
            }
            std::vector<int32_t> tl, tr;
            build_top_tree(boxes.data(), counts.data(), m, tl, tr);
            for (uint32_t i = 0; i + 1 < m; i++) top_apply(w, i, tl.data(), tr.data(), cl.data());
        }
        parent[0] = NO_PARENT;
    }
    for (int k = 0; k < n && n > 1; k++) {   // bottom-up fit: the second arrival at a node fits it
        uint32_t cur = parent[n - 1 + k];
        while (cur != NO_PARENT) {
            if (flags[cur]++ == 0) break;
            rotate_children(n, cur, (uint32_t)max_leaf, w);
            fit_inner(n, cur, (uint32_t)max_leaf, w);
            cur = parent[cur];
        }
    }
    for (int i = 0; i < N; i++) finish_node(p, n, (uint32_t)i, w, prim_pos.data());
    {
        double cost = 0;
        for (int i = 0; i < N; i++) cost += sah_share(w, (uint32_t)i);
        hs.bvh_sah_cost = cost / node_area(w, 0);
    }
    const uint32_t total = size[0];
    float rmn[3], rmx[3];
    for (int a = 0; a < 3; a++) { rmn[a] = pad_down(bmin[a]); rmx[a] = pad_up(bmax[a]); }
    qframe_from_box(rmn, rmx, hs.qmin, hs.qstep);
    hs.nodes.assign(8 * (size_t)total + 1, Node{});
    hs.qnodes.assign(8 * ((size_t)total + 1), QNode{});
    for (int o = 0; o < 8; o++) {
        for (int i = 0; i < N; i++) emit_node((uint32_t)i, o, w, hs.qmin, hs.qstep, hs.nodes.data(), hs.qnodes.data());
        hs.qnodes[(size_t)o * (total + 1) + total] = qnode_sentinel(o);
    }
    hs.num_nodes = total;
    std::vector<Tri> tris(p.num_tris);
    std::vector<f4> spheres(hs.spheres.size()), cuboids(hs.cuboids.size());
    std::vector<int32_t> sphere_elem(hs.sphere_elem.size());
    std::vector<f4> sphere_lo(hs.sphere_lo.size());
    for (int k = 0; k < n; k++) {
        uint32_t i = key_index(p, keys[k]), d = prim_pos[k];
        if (i < p.num_tris) tris[d] = hs.tris[p.ref_tri ? p.ref_tri[i] : i];
        else if (i < p.num_tris + p.num_spheres) { uint32_t l = i - p.num_tris; d -= p.num_tris; spheres[d] = hs.spheres[l]; sphere_elem[d] = hs.sphere_elem[l]; sphere_lo[d] = hs.sphere_lo[l]; }
        else { uint32_t l = i - p.num_tris - p.num_spheres; d -= p.num_tris + p.num_spheres; cuboids[2 * d] = hs.cuboids[2 * l]; cuboids[2 * d + 1] = hs.cuboids[2 * l + 1]; }
    }
    hs.tris.swap(tris); hs.derive_triangles(); hs.spheres.swap(spheres); hs.sphere_elem.swap(sphere_elem); hs.sphere_lo.swap(sphere_lo); hs.cuboids.swap(cuboids);
    // leaves / depth of the emitted tree for the stats call
    hs.bvh_leaves = 0; hs.bvh_max_depth = 0;
    std::vector<std::pair<uint32_t, uint32_t>> st{{0u, 0u}};
    while (!st.empty()) {
        auto [id, depth] = st.back();
        st.pop_back();
        hs.bvh_max_depth = std::max(hs.bvh_max_depth, depth);
        if (is_collapsed(w, id)) { hs.bvh_leaves++; continue; }
        st.push_back({left[id], depth + 1}); st.push_back({right[id], depth + 1});
    }
}

int emu_scene_create(const hr_scene_desc *sd, emu_scene **out) {
    emu_scene *e = new emu_scene;
    std::string err;
    int rc = flatten_scene(sd, e->hs, err, g_max_leaf, g_builder ? 0.0 : g_split_ratio, g_builder == 0);
    if (rc) { fprintf(stderr, "emu: %s\n", err.c_str()); delete e; return rc; }
    if (g_builder) device_build_host(e->hs, g_max_leaf, g_builder, g_split_ratio);
    e->view = e->hs.view();
    *out = e;
    return 0;
}
void emu_scene_destroy(emu_scene *e) { delete e; }

double emu_scene_sah_cost(const emu_scene *e) { return e->hs.bvh_sah_cost; }

// stats: [0]=nodes [1]=leaves [2]=max depth [3]=tris [4]=spheres [5]=cuboids [6]=emitters
void emu_scene_stats(const emu_scene *e, uint64_t *out) {
    out[0] = e->hs.num_nodes; out[1] = e->hs.bvh_leaves; out[2] = e->hs.bvh_max_depth;
    out[3] = e->hs.tris.size(); out[4] = e->hs.spheres.size(); out[5] = e->hs.cuboids.size() / 2; out[6] = e->hs.emitters.size();
}

int emu_path_draws(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int lens_shape, float *out20) {
    return path_draws(W, H, px, py, sub, sampling, lens_shape, out20) ? 0 : 1;
}
// the same slots of the record's twin (RecordTail<.., LO> / record_from_window<LO>): what rounding each draw to fp32 took away; slots 0, 1 = the
// residuals of the RAW lens draws u, v (path_draws reports the lens point 2 u - 1 there, from the head)
int emu_path_draw_residuals(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int lens_shape, float *out20) {
    ArrRec rec;
    bool ok = path_record(W, H, px, py, sub, sampling, lens_shape, rec);
    uint32_t a = float_as_uint(rec.at(REC_HEAD));
    for (int d = 0; d < DRAWS_PER_PATH; d++) out20[d] = rec.f[REC_ITEM_FLOATS + rec_slot(rec.base, (uint32_t)(2 * (int)a + d))];
    return ok ? 0 : 1;
}

// raw tail: out[k] = k-th next_u64, k < window <= ISAAC_TAIL
struct RawTail { uint64_t *out; int window; void put(int step, u64 v) { int k = 255 - step; if (k < window) out[k] = v; } };
int emu_raw_draws(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int window, uint64_t *out) {
    static const IsaacWarm warm = isaac_warm();
    ArrMem mem;
    RawTail rt{out, window};
    u64 s, t;
    path_seed_words(W, H, px, py, sub, s, t);
    isaac_seed_round<ISAAC_TAIL>(mem, warm, 8700304ULL, (u64)sampling, s, t, rt);
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
    isaac_round<ISAAC_TAIL>(mem, rt);
    return 0;
}

int emu_raw_draws_pc(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int window, int head, uint64_t *out) {
    u64 s, t;
    path_seed_words(W, H, px, py, sub, s, t);
    switch (head) {
        case 0: raw_draws_pc<0>(s, t, sampling, window, out); return 0;
        case 8: raw_draws_pc<8>(s, t, sampling, window, out); return 0;
        case 16: raw_draws_pc<16>(s, t, sampling, window, out); return 0;
        case 24: raw_draws_pc<24>(s, t, sampling, window, out); return 0;
        case 32: raw_draws_pc<32>(s, t, sampling, window, out); return 0;
        default: return 1;
    }
}

int emu_raw_draws_seg(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, uint32_t sampling, int window, uint64_t *out) {
    u64 s, t;
    path_seed_words(W, H, px, py, sub, s, t);
    raw_draws_seg<SEG_B1, SEG_B2, SEG_NBLK>(s, t, sampling, window, out);
    return 0;
}

// nee_setup's shortcuts (pt_core.h): mask of the ones in force (7 = all: the kernel's default), 0 = every NEE shadow ray is traced
static int g_nee_cull = 7;
void emu_set_nee_cull(int on) { g_nee_cull = on; }
// option precise_shading for the megakernel's per-lane code (path_advance<.., PREC>): emu_render / emu_path_log
static int g_precise = 0;
void emu_set_precise(int on) { g_precise = on; }

// counters: paths, rays, node_tests, tri_tests, sphere_tests, cuboid_tests, shadow_culled
int emu_render(const emu_scene *e, uint32_t W, uint32_t H, uint32_t s_begin, uint32_t s_end, uint32_t stride, int nthreads, float *acc,
               uint64_t *counters) {
    Scene sc = e->view;
    sc.qnodes = nullptr;   // the emulated render walks the 32-byte records (node indices): every ray's walk starts at 0
    RenderParams rp{};
    rp.width = W; rp.height = H;
    rp.nee_cull_off = ~(uint32_t)g_nee_cull & 7u;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    std::vector<std::vector<uint64_t>> cn(nthreads, std::vector<uint64_t>(7, 0));
    for (uint32_t sampling = s_begin; sampling < s_end; sampling += stride) {
        std::atomic<uint32_t> next{0};
        auto work = [&](int tid) {
            for (;;) {
                uint32_t y = next.fetch_add(1);
                if (y >= H) break;
                for (uint32_t x = 0; x < W; x++) {
                    float sum[3] = {0, 0, 0};
                    for (uint32_t sub = 0; sub < 4; sub++) {
                        ArrRec rec;
                        Path p;
                        RenderParams rpp = rp;
                        emu_place_path(rpp, W, H, x, y, sub, p, rec);
                        path_record(W, H, x, y, sub, sampling, sc.cam.lens_shape, rec);
                        path_start(sc, rpp, p, x, y, sub, rec.f);
                        LaneCounters lc = {0, 0, 0, 0, 0, 0};
                        for (;;) {
                            while (p.ts.cur != NODE_END) { trace_step<true>(sc, p.ray, p.ts, &lc); shadow_early_out(p); }
                            if (g_precise ? path_advance<true, false, false, true>(sc, rpp, p, rec.f, &lc) : path_advance<true>(sc, rpp, p, rec.f, &lc)) break;
                        }
                        sum[0] += p.accum.x; sum[1] += p.accum.y; sum[2] += p.accum.z;
                        cn[tid][0]++; cn[tid][1] += lc.rays; cn[tid][2] += lc.node_tests; cn[tid][3] += lc.tri_tests;
                        cn[tid][4] += lc.sphere_tests; cn[tid][5] += lc.cuboid_tests; cn[tid][6] += lc.shadow_culled;
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
        for (int k = 0; k < 7; k++) { counters[k] = 0; for (auto &c : cn) counters[k] += c[k]; }
    return 0;
}

// The split pipeline (csrc/wf_core.h, wf_kernels.h), one path at a time: the same step structure as the kernels — per iteration, walk the
// shadow rays of the iteration before and the main ray, add the contributions in order, shade the main hit, emit — on the same per-lane
// functions.  PREC: precise shading (wf_surface_f64).  tests/test_emu_parity.py compares the fp32 form with emu_render's (path_advance), bit for bit.
}  // extern "C"
struct EmuRay { V3f o; float len; V3f d; float w; V3f o_lo, d_lo; uint32_t emitter; };
template <bool PREC, bool LOG>
static V3f emu_wf_path(const Scene &sc, const RenderParams &rp, uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t sub, uint32_t sampling, uint32_t cull, PathLog *lg) {
    ArrRec rec;
    Path p0;
    RenderParams rpp = rp;
    emu_place_path(rpp, W, H, x, y, sub, p0, rec);
    path_record(W, H, x, y, sub, sampling, sc.cam.lens_shape, rec);
    path_start(sc, rpp, p0, x, y, sub, rec.f);
    const float *prec = rec.f + rec.base;      // the path's record, as wf_rec_base() addresses it on the device
    WfPath p;
    p.pid = 0; p.st = wf_st(1u, true, (p0.q >> 12) & 15u, 0u); p.raybase = 0; p.cur_refl = 1.0f;
    p.accum = v3(0, 0, 0); p.refl = v3(1, 1, 1);
    EmuRay first{p0.ray.o, WF_MAIN_RAY, p0.ray.d, 0.0f, v3(0, 0, 0), v3(0, 0, 0), 0u};
    if (PREC) ray_fix_load(prec, 0u, (p0.q >> 12) & 15u, first.o_lo, first.d_lo);   // path_start parked the f64 camera ray's residuals there
    std::vector<EmuRay> rays{first}, nxt;
    LaneCounters lc = {0, 0, 0, 0, 0, 0};
    for (uint32_t step = 1; step <= 11u; step++) {
        std::vector<WfHitRec> hits;
        for (const EmuRay &r : rays) {
            TravLane l;
            wf_lane_begin(sc, l, r.o, r.d, r.len);
            while (l.ts.cur != NODE_END) { trace_step<false>(sc, l.ray, l.ts, &lc); shadow_early_out(l); }
            hits.push_back(wf_hit_pack(l.ts));
        }
        const uint32_t ns = wf_shadow_rays(p);
        // (the shadow rays belong to the iteration before the main ray's — or, without a main ray, to the iteration the state still names)
        const uint32_t it_shadow = wf_has_main(p) ? wf_iter(p) - 1u : wf_iter(p);
        for (uint32_t k = 0; k < ns; k++) {
            const V3f before = p.accum;
            wf_contribute<false>(sc, p, hits[k], rays[k].o, rays[k].len, rays[k].d, rays[k].w, &lc);
            if (LOG) {
                lg->rays++;
                TraceState ts;
                wf_hit_unpack(hits[k], ts);
                const float dt = ts.t - rays[k].len;
                if (ts.prim >= 0 && dt * dt < OFFSET_F * 4.0f) plog_or(*lg, it_shadow, 16u << (rays[k].emitter & 3u));
                (void)before;
            }
        }
        if (!wf_has_main(p)) break;
        p.refl = p.refl * p.cur_refl;
        WfBounce b;
        WfBounceX bx;
        b.nee = false;
        bx.next_o_lo = bx.next_d_lo = v3(0, 0, 0);
        bool fin;
        if (PREC) fin = wf_surface_f64<false, LOG>(sc, p, prec, rpp.rec_lo_off ? prec + rpp.rec_lo_off : nullptr, rays[ns].o, rays[ns].d, rays[ns].o_lo, rays[ns].d_lo, hits[ns], b, bx, &lc, lg);
        else fin = wf_surface<false, LOG>(sc, p, prec, rays[ns].o, rays[ns].d, hits[ns], b, &lc, lg);
        if (fin) break;
        nxt.clear();
        if (b.nee)
            for (uint32_t k = 0; k < sc.num_emitters; k++) {
                V3f d; float len;
                if (wf_nee_ray(sc, b, k, LOG ? (cull & 5u) : cull, d, len)) nxt.push_back(EmuRay{b.next_o, len, d, wf_nee_weight(sc, b, k, d, len), v3(0, 0, 0), v3(0, 0, 0), k});
                else if (LOG) lg->rays++;
            }
        const uint32_t ns_new = (uint32_t)nxt.size();
        const bool bounce = wf_bounces(p, b);
        if (!ns_new && !bounce) break;
        if (bounce) nxt.push_back(EmuRay{b.next_o, WF_MAIN_RAY, b.next_d, 0.0f, bx.next_o_lo, bx.next_d_lo, 0u});
        p.st = wf_st(wf_iter(p) + (bounce ? 1u : 0u), bounce, wf_a2(p), ns_new);
        p.cur_refl = b.cur_refl;
        rays.swap(nxt);
    }
    return p.accum;
}
extern "C" {
static int g_wf_precise = 0;
extern "C" void emu_set_wf_precise(int on) { g_wf_precise = on; }
extern "C" void emu_set_draw_residuals(int on) { g_draw_residuals = on; }
extern "C" int emu_render_wf(const emu_scene *e, uint32_t W, uint32_t H, uint32_t s_begin, uint32_t s_end, uint32_t stride, int nthreads, float *acc) {
    Scene sc = e->view;
    sc.qnodes = nullptr;
    RenderParams rp{};
    rp.width = W; rp.height = H;
    const uint32_t cull = (uint32_t)g_nee_cull & 7u;
    const bool precise = g_wf_precise != 0;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    for (uint32_t sampling = s_begin; sampling < s_end; sampling += stride) {
        std::atomic<uint32_t> next{0};
        auto work = [&]() {
            for (;;) {
                uint32_t y = next.fetch_add(1);
                if (y >= H) break;
                for (uint32_t x = 0; x < W; x++) {
                    float sum[3] = {0, 0, 0};
                    for (uint32_t sub = 0; sub < 4; sub++) {
                        const V3f a = precise ? emu_wf_path<true, false>(sc, rp, W, H, x, y, sub, sampling, cull, nullptr) : emu_wf_path<false, false>(sc, rp, W, H, x, y, sub, sampling, cull, nullptr);
                        sum[0] += a.x; sum[1] += a.y; sum[2] += a.z;
                    }
                    float *o = &acc[((size_t)y * W + x) * 3];
                    o[0] += sum[0]; o[1] += sum[1]; o[2] += sum[2];
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(work);
        for (auto &t : th) t.join();
    }
    return 0;
}
// emu_path_log's layout from the split pipeline with precise shading
extern "C" int emu_path_log_wf(const emu_scene *e, uint32_t W, uint32_t H, uint32_t sampling, int nthreads, uint32_t *out) {
    Scene sc = e->view;
    sc.qnodes = nullptr;
    RenderParams rp{};
    rp.width = W; rp.height = H;
    const uint32_t cull = (uint32_t)g_nee_cull & 7u;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    std::atomic<uint32_t> next{0};
    auto work = [&]() {
        for (;;) {
            uint32_t y = next.fetch_add(1);
            if (y >= H) break;
            for (uint32_t x = 0; x < W; x++)
                for (uint32_t sub = 0; sub < 4; sub++) {
                    PathLog lg;
                    plog_reset(lg);
                    const V3f a = emu_wf_path<true, true>(sc, rp, W, H, x, y, sub, sampling, cull, &lg);
                    uint32_t *o = out + (((size_t)y * W + x) * 4 + sub) * 8;
                    o[0] = float_as_uint(a.x); o[1] = float_as_uint(a.y); o[2] = float_as_uint(a.z); o[3] = lg.rays;
                    o[4] = (uint32_t)lg.ev; o[5] = (uint32_t)(lg.ev >> 32); o[6] = lg.ev9; o[7] = lg.hash;
                }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work);
    for (auto &t : th) t.join();
    return 0;
}

// ONE path (debugging aid: build with -DHR_PATH_VERBOSE and set HR_V=1 for path_advance's line per finished ray): its radiance, fp32 or
// precise shading as emu_set_precise says
int emu_one_path(const emu_scene *e, uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t sub, uint32_t sampling, float *out3) {
    Scene sc = e->view;
    sc.qnodes = nullptr;
    RenderParams rp{};
    rp.width = W; rp.height = H;
    rp.nee_cull_off = ~(uint32_t)g_nee_cull & 7u;
    ArrRec rec;
    Path p;
    emu_place_path(rp, W, H, x, y, sub, p, rec);
    path_record(W, H, x, y, sub, sampling, sc.cam.lens_shape, rec);
    path_start(sc, rp, p, x, y, sub, rec.f);
    LaneCounters lc = {0, 0, 0, 0, 0, 0};
    for (;;) {
        while (p.ts.cur != NODE_END) { trace_step<true>(sc, p.ray, p.ts, &lc); shadow_early_out(p); }
        if (g_precise ? path_advance<true, false, false, true>(sc, rp, p, rec.f, &lc) : path_advance<true>(sc, rp, p, rec.f, &lc)) break;
    }
    out3[0] = p.accum.x; out3[1] = p.accum.y; out3[2] = p.accum.z;
    return 0;
}

// hr_debug_path_log's layout from the emulated render: eight words per path {r, g, b (float bits), rays, ev 0-3, ev 4-7, ev 8, hash}
// (path_advance<.., LOG> of pt_core.h — the code the kernel's LOG instantiation runs)
int emu_path_log(const emu_scene *e, uint32_t W, uint32_t H, uint32_t sampling, int nthreads, uint32_t *out) {
    Scene sc = e->view;
    sc.qnodes = nullptr;
    RenderParams rp{};
    rp.width = W; rp.height = H;
    rp.nee_cull_off = ~(uint32_t)g_nee_cull & 7u;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    std::atomic<uint32_t> next{0};
    auto work = [&]() {
        for (;;) {
            uint32_t y = next.fetch_add(1);
            if (y >= H) break;
            for (uint32_t x = 0; x < W; x++)
                for (uint32_t sub = 0; sub < 4; sub++) {
                    ArrRec rec;
                    Path p;
                    RenderParams rpp = rp;
                    emu_place_path(rpp, W, H, x, y, sub, p, rec);
                    path_record(W, H, x, y, sub, sampling, sc.cam.lens_shape, rec);
                    path_start(sc, rpp, p, x, y, sub, rec.f);
                    LaneCounters lc = {0, 0, 0, 0, 0, 0};
                    PathLog lg;
                    plog_reset(lg);
#if defined(HR_PATH_VERBOSE)
                    if (getenv("HR_VPIX")) { if ((uint32_t)atoi(getenv("HR_VPIX")) == (y * W + x) * 4 + sub) setenv("HR_V", "1", 1); else unsetenv("HR_V"); }
#endif
                    for (;;) {
                        while (p.ts.cur != NODE_END) { trace_step<true>(sc, p.ray, p.ts, &lc); shadow_early_out(p); }
                        if (g_precise ? path_advance<true, false, true, true>(sc, rpp, p, rec.f, &lc, 0u, 0u, &lg) : path_advance<true, false, true>(sc, rpp, p, rec.f, &lc, 0u, 0u, &lg)) break;
                    }
                    uint32_t *o = out + (((size_t)y * W + x) * 4 + sub) * 8;
                    o[0] = float_as_uint(p.accum.x); o[1] = float_as_uint(p.accum.y); o[2] = float_as_uint(p.accum.z); o[3] = lg.rays;
                    o[4] = (uint32_t)lg.ev; o[5] = (uint32_t)(lg.ev >> 32); o[6] = lg.ev9; o[7] = lg.hash;
                }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work);
    for (auto &t : th) t.join();
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

// 0 = node + leaf per visit (trace_step); 1 = the trace kernel's lane schedule: walk with up to one leaf parked, stop at the
// second (trace_node<.., SPEC>), test the parked leaves in walk order; 2 = the same schedule on the 16-byte quantised nodes
// (trace_qnode, host-built trees only)
static int g_walk_mode = 0;
static uint64_t g_node_tests = 0;
void emu_set_walk_mode(int mode) { g_walk_mode = mode; }
// the governor's wave-budget control law as the device compiles it (device_scene.h)
uint32_t emu_gov_budget_next(uint32_t B, float ratio, uint32_t lo, uint32_t hi, uint32_t step) { return gov_budget_next(B, ratio, lo, hi, step); }
uint64_t emu_last_node_tests(void) { return g_node_tests; }

int emu_intersect(const emu_scene *e, uint32_t n, const float *rays, float *out, int32_t *out_elem) {
    const Scene &sc = e->view;
    g_node_tests = 0;
    if (g_walk_mode == 2 && !sc.qnodes) return 1;
    for (uint32_t i = 0; i < n; i++) {
        Ray r;
        ray_set(r, v3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), v3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]));
        ray_quantise(sc, r);
        TraceState ts;
        trace_begin(ts, T_INF, g_walk_mode == 2 ? r.start : 0u);   // 16-byte records: byte offset of the octant's copy; 32-byte records: node index
        LaneCounters lc = {0, 0, 0, 0, 0, 0};
        if (g_walk_mode == 0) {
            while (ts.cur != NODE_END) trace_step<true>(sc, r, ts, &lc);
        } else {
            while (!trace_done(ts)) {
                if (g_walk_mode == 2) while (ts.cur != NODE_END && ts.leaf2 == 0) trace_qnode<true, true>(sc, r, ts, &lc);
                else while (ts.cur != NODE_END && ts.leaf2 == 0) trace_node<true, true>(sc, r, ts, &lc);
                if (ts.leaf) trace_leaf_next<true>(sc, r, ts, &lc);
            }
        }
        g_node_tests += lc.node_tests;
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
