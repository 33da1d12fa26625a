#include "bvh_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <emmintrin.h>

namespace hr {
namespace {

struct Box {
    double mn[3], mx[3];
    void reset() { for (int a = 0; a < 3; a++) { mn[a] = DBL_MAX; mx[a] = -DBL_MAX; } }
    void grow(const double *bmin, const double *bmax) {
        for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], bmin[a]); mx[a] = std::max(mx[a], bmax[a]); }
    }
    void grow(const Box &b) { grow(b.mn, b.mx); }
    double area() const {
        double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0) return 0.0;
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
};

struct BNode {
    Box box;
    int left = -1, right = -1;
    uint32_t first = 0, count = 0;
    int type = -1, axis = 0;
};

struct Builder {
    const std::vector<BuildPrim> &prims;
    std::vector<uint32_t> idx;
    std::vector<BNode> nodes;
    int max_leaf;
    uint32_t max_depth = 0;

    Builder(const std::vector<BuildPrim> &p, int ml) : prims(p), max_leaf(ml) {
        idx.resize(p.size());
        for (uint32_t i = 0; i < p.size(); i++) idx[i] = i;
    }
    double centroid(uint32_t i, int a) const { return 0.5 * (prims[i].bmin[a] + prims[i].bmax[a]); }

    // decides what becomes of the group [first, first + count): a leaf (returns false) or two groups split at `mid` along `axis`
    bool split_group(int id, uint32_t first, uint32_t count, uint32_t &mid, int &axis) {
        Box box, cbox;
        box.reset(); cbox.reset();
        bool same_type = true;
        for (uint32_t i = first; i < first + count; i++) {
            const BuildPrim &p = prims[idx[i]];
            box.grow(p.bmin, p.bmax);
            double c[3] = {centroid(idx[i], 0), centroid(idx[i], 1), centroid(idx[i], 2)};
            cbox.grow(c, c);
            if (p.type != prims[idx[first]].type) same_type = false;
        }
        nodes[id].box = box;
        mid = 0;
        axis = 0;
        bool split = false;
        if (!same_type) {
            // force type-homogeneous subtrees: peel off the first prim's type
            int t0 = prims[idx[first]].type;
            auto it = std::stable_partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t i) { return prims[i].type == t0; });
            mid = (uint32_t)(it - idx.begin());
            // choose the axis along which the two groups' centroids differ most (for near/far ordering)
            double c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
            for (uint32_t i = first; i < mid; i++) for (int a = 0; a < 3; a++) c0[a] += centroid(idx[i], a) / (mid - first);
            for (uint32_t i = mid; i < first + count; i++) for (int a = 0; a < 3; a++) c1[a] += centroid(idx[i], a) / (first + count - mid);
            double best = -1;
            for (int a = 0; a < 3; a++) if (std::fabs(c1[a] - c0[a]) > best) { best = std::fabs(c1[a] - c0[a]); axis = a; }
            if (c1[axis] < c0[axis]) {  // keep "left = lower coordinate" convention
                std::rotate(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count);
                mid = first + (first + count - mid);
            }
            split = true;
        } else if (count > 1) {
            // binned SAH over the three axes
            const int NB = 32;
            double best_cost = DBL_MAX;
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                double lo = cbox.mn[a], hi = cbox.mx[a];
                if (!(hi > lo)) continue;
                Box bb[NB];
                uint32_t bc[NB];
                for (int b = 0; b < NB; b++) { bb[b].reset(); bc[b] = 0; }
                double scale = NB / (hi - lo);
                for (uint32_t i = first; i < first + count; i++) {
                    int b = (int)((centroid(idx[i], a) - lo) * scale);
                    b = std::min(std::max(b, 0), NB - 1);
                    bb[b].grow(prims[idx[i]].bmin, prims[idx[i]].bmax);
                    bc[b]++;
                }
                double ra[NB];
                uint32_t rc[NB];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int b = NB - 1; b > 0; b--) { acc.grow(bb[b]); cnt += bc[b]; ra[b] = acc.area(); rc[b] = cnt; }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < NB - 1; b++) {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    if (!cnt || !rc[b + 1]) continue;
                    double cost = acc.area() * cnt + ra[b + 1] * rc[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            // A group that already fits a leaf is split further only if that pays with a primitive test priced at HALF a node visit:
            // measured on the headline scene (trace kernel alone, ms per 33 M paths) 1.0 -> 19.19 (37.9 node + 5.9 triangle tests
            // per ray), 0.5 -> 18.75 (36.8 + 6.7), <= 0.35 -> 18.72 (never split), 2.0 -> 20.5.  A leaf's triangles are tested with
            // their loads in flight together; every node visit is a dependent load of its own.
            const double prim_cost = 0.5;
            double leaf_cost = box.area() * count * prim_cost;
            bool want_split = best_axis >= 0 && ((int)count > max_leaf || best_cost * prim_cost + box.area() * 1.0 < leaf_cost);
            if (want_split) {
                axis = best_axis;
                double lo = cbox.mn[axis], hi = cbox.mx[axis];
                const int NBk = 32;
                double scale = NBk / (hi - lo);
                auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t i) {
                    int b = (int)((centroid(i, axis) - lo) * scale);
                    b = std::min(std::max(b, 0), NBk - 1);
                    return b <= best_bin;
                });
                mid = (uint32_t)(it - idx.begin());
                split = mid > first && mid < first + count;
            }
            if (!split && (int)count > max_leaf) {  // degenerate: median split on the widest axis
                axis = 0;
                for (int a = 1; a < 3; a++) if (box.mx[a] - box.mn[a] > box.mx[axis] - box.mn[axis]) axis = a;
                mid = first + count / 2;
                std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count,
                                 [&](uint32_t a, uint32_t b) { return centroid(a, axis) < centroid(b, axis); });
                split = true;
            }
        }
        if (!split) { nodes[id].first = first; nodes[id].count = count; nodes[id].type = prims[idx[first]].type; }
        return split;
    }
    // Nodes are allocated in depth-first preorder, left child first (build_bvh relies on index == memory order).  An explicit stack, not
    // recursion: a split that peels one outlier off per level makes the tree as deep as the scene has primitives, and split_group's
    // frame (the SAH bins) times that depth would not fit a thread's stack.
    void build(uint32_t first0, uint32_t count0) {
        struct Item { uint32_t first, count, depth; int parent; bool is_right; };
        std::vector<Item> todo;
        todo.push_back(Item{first0, count0, 0u, -1, false});
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            const int id = (int)nodes.size();
            nodes.emplace_back();
            if (it.parent >= 0) (it.is_right ? nodes[it.parent].right : nodes[it.parent].left) = id;
            max_depth = std::max(max_depth, it.depth);
            uint32_t mid = 0;
            int axis = 0;
            if (!split_group(id, it.first, it.count, mid, axis)) continue;
            nodes[id].axis = axis;
            todo.push_back(Item{mid, it.first + it.count - mid, it.depth + 1, id, true});     // (popped after the whole left subtree)
            todo.push_back(Item{it.first, mid - it.first, it.depth + 1, id, false});
        }
    }
};


static float round_down(double v, int ulps) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -INFINITY);
    for (int i = 0; i < ulps; i++) f = std::nextafterf(f, -INFINITY);
    return f;
}
static float round_up(double v, int ulps) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, INFINITY);
    for (int i = 0; i < ulps; i++) f = std::nextafterf(f, INFINITY);
    return f;
}

}  // namespace

void build_bvh(const std::vector<BuildPrim> &prims, int max_leaf, BuiltBvh &out) {
    out.nodes.clear();
    out.qnodes.clear();
    for (auto &o : out.order) o.clear();
    out.max_depth = 0; out.num_leaves = 0; out.num_nodes = 0;
    if (prims.empty()) {
        // a single box that nothing hits
        Node n{};
        const float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        node_set_box(n, mn, mx, 0);   // near > far on every axis: nothing hits it in any octant
        n.a = NODE_END; n.b = NODE_END;
        out.nodes.assign(8 + 1, n);
        out.num_nodes = 1;
        return;
    }
    Builder b(prims, max_leaf);
    b.build(0, (uint32_t)prims.size());
    out.max_depth = b.max_depth;
    size_t N = b.nodes.size();
    out.num_nodes = (uint32_t)N;
    // the builder allocates in DFS preorder (left first): index == memory order
    std::vector<uint32_t> leaf_word(N, 0);
    for (size_t i = 0; i < N; i++) {
        const BNode &bn = b.nodes[i];
        if (bn.left < 0) {
            uint32_t first = (uint32_t)out.order[bn.type].size();
            for (uint32_t k = 0; k < bn.count; k++) out.order[bn.type].push_back(prims[b.idx[bn.first + k]].index);
            leaf_word[i] = hr::leaf_word((uint32_t)bn.type, bn.count, first);   // count <= max_leaf <= 15: build() splits every larger group
            out.num_leaves++;
        }
    }
    {
        const double root_area = std::max(b.nodes[0].box.area(), 1e-300);
        double cost = 0;
        for (size_t i = 0; i < N; i++) {
            const BNode &bn = b.nodes[i];
            cost += bn.box.area() / root_area * (1.0 + (bn.left < 0 ? 1.5 * bn.count : 0.0));
        }
        out.sah_cost = cost;
    }
    // Every octant's copy is stored in ITS OWN traversal order (preorder, near child first): the near child of an inner node
    // is then the next record in memory, and the trace kernel fetches a node together with its successor in one go — the
    // dependent load that follows an inner hit is contiguous with its parent (a paired fetch of node + successor was measured: the extra loads cost more than the saved round trips).  One spare record at the very
    // end keeps that paired fetch in bounds.
    out.nodes.assign(8 * N + 1, Node{});
    out.qnodes.assign(8 * (N + 1), QNode{});
    {   // grid of the 16-bit planes (device_scene.h) over the root's padded fp32 box
        float rmn[3], rmx[3];
        for (int a = 0; a < 3; a++) { rmn[a] = round_down(b.nodes[0].box.mn[a], 2); rmx[a] = round_up(b.nodes[0].box.mx[a], 2); }
        qframe_from_box(rmn, rmx, out.qmin, out.qstep);
    }
    std::vector<uint32_t> newid(N);
    for (int o = 0; o < 8; o++) {
        Node *nd = &out.nodes[(size_t)o * N];
        // pass 1: number the nodes in this octant's order
        std::vector<int> st;
        st.push_back(0);
        uint32_t next = 0;
        while (!st.empty()) {
            int id = st.back();
            st.pop_back();
            newid[id] = next++;
            const BNode &bn = b.nodes[id];
            if (bn.left < 0) continue;
            bool neg = (o >> bn.axis) & 1;  // ray travels toward -axis: the higher-coordinate child is nearer
            int nearc = neg ? bn.right : bn.left, farc = neg ? bn.left : bn.right;
            st.push_back(farc);
            st.push_back(nearc);
        }
        // pass 2: boxes and successors, with an explicit stack of (node, next_after_subtree)
        std::vector<std::pair<int, uint32_t>> st2;
        st2.emplace_back(0, NODE_END);
        while (!st2.empty()) {
            auto [id, after] = st2.back();
            st2.pop_back();
            const BNode &bn = b.nodes[id];
            Node &n = nd[newid[id]];
            float mn[3], mx[3];
            for (int a = 0; a < 3; a++) { mn[a] = round_down(bn.box.mn[a], 2); mx[a] = round_up(bn.box.mx[a], 2); }
            node_set_box(n, mn, mx, o);
            n.b = after;
            out.qnodes[(size_t)o * (N + 1) + newid[id]] = qnode_make(mn, mx, o, out.qmin, out.qstep, bn.left < 0 ? leaf_word[id] : qnode_link(o, (uint32_t)(N + 1), after));
            if (bn.left < 0) { n.a = leaf_word[id]; continue; }
            bool neg = (o >> bn.axis) & 1;
            int nearc = neg ? bn.right : bn.left, farc = neg ? bn.left : bn.right;
            n.a = newid[nearc];   // == newid[id] + 1
            st2.emplace_back(nearc, newid[farc]);
            st2.emplace_back(farc, after);
        }
    }
    for (int o = 0; o < 8; o++) out.qnodes[(size_t)o * (N + 1) + N] = qnode_sentinel(o);   // sentinel behind every copy
    {
        Node &pad = out.nodes[8 * N];
        const float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        node_set_box(pad, mn, mx, 0);
        pad.a = NODE_END; pad.b = NODE_END;
    }
}

namespace {
// Binned SAH over cluster boxes, weighted by the primitives each cluster holds.  This runs on the host between two
// device phases of builder 2, so its time is build time: boxes are four-float SSE values (x, y, z, pad), the three axes
// are binned in one pass over the range, and small ranges use few bins (most of a tree's nodes are small).
struct TopBuilder {
    static constexpr int NBMAX = 32;
    struct FBox {
        __m128 mn, mx;
        void reset() { mn = _mm_set1_ps(FLT_MAX); mx = _mm_set1_ps(-FLT_MAX); }
        void grow(const FBox &o) { mn = _mm_min_ps(mn, o.mn); mx = _mm_max_ps(mx, o.mx); }
        float area() const {
            alignas(16) float d[4];
            _mm_store_ps(d, _mm_sub_ps(mx, mn));
            return d[0] < 0 ? 0.0f : d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
        }
    };
    const uint32_t *counts;
    std::vector<uint32_t> idx;
    std::vector<FBox> box;
    struct V4 { __m128 v; };
    std::vector<V4> cen;
    std::vector<int32_t> &left, &right;
    TopBuilder(const float *b, const uint32_t *c, uint32_t m, std::vector<int32_t> &l, std::vector<int32_t> &r) : counts(c), left(l), right(r) {
        idx.resize(m);
        box.resize(m);
        cen.resize(m);
        for (uint32_t i = 0; i < m; i++) {
            const float *p = b + 6 * (size_t)i;
            idx[i] = i;
            box[i].mn = _mm_set_ps(0.0f, p[2], p[1], p[0]);
            box[i].mx = _mm_set_ps(0.0f, p[5], p[4], p[3]);
            cen[i].v = _mm_mul_ps(_mm_set1_ps(0.5f), _mm_add_ps(box[i].mn, box[i].mx));
        }
    }
    // where the range [first, first + count), count >= 2, is split
    uint32_t split_range(uint32_t first, uint32_t count) {
        uint32_t mid = first + 1;
        if (count > 2) {
            __m128 lo = _mm_set1_ps(FLT_MAX), hi = _mm_set1_ps(-FLT_MAX);
            for (uint32_t i = first; i < first + count; i++) { lo = _mm_min_ps(lo, cen[idx[i]].v); hi = _mm_max_ps(hi, cen[idx[i]].v); }
            const int NB = (int)std::min<uint32_t>(NBMAX, std::max<uint32_t>(4u, count));
            alignas(16) float ext[4], lof[4], scalef[4];
            _mm_store_ps(ext, _mm_sub_ps(hi, lo));
            _mm_store_ps(lof, lo);
            for (int a = 0; a < 3; a++) scalef[a] = ext[a] > 0 ? (float)NB / ext[a] : 0.0f;
            scalef[3] = 0.0f;
            const __m128 scale = _mm_load_ps(scalef), top = _mm_set1_ps((float)(NB - 1));
            FBox bb[3][NBMAX];
            float bc[3][NBMAX];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < NB; b++) { bb[a][b].reset(); bc[a][b] = 0; }
            for (uint32_t i = first; i < first + count; i++) {
                const uint32_t c = idx[i];
                alignas(16) int bin[4];
                _mm_store_si128((__m128i *)bin, _mm_cvttps_epi32(_mm_min_ps(_mm_max_ps(_mm_mul_ps(_mm_sub_ps(cen[c].v, lo), scale), _mm_setzero_ps()), top)));
                const float w = (float)counts[c];
                for (int a = 0; a < 3; a++) { bb[a][bin[a]].grow(box[c]); bc[a][bin[a]] += w; }
            }
            float best_cost = FLT_MAX;
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                if (!(scalef[a] > 0)) continue;
                float ra[NBMAX], rc[NBMAX];
                FBox acc;
                acc.reset();
                float cnt = 0;
                for (int b = NB - 1; b > 0; b--) { acc.grow(bb[a][b]); cnt += bc[a][b]; ra[b] = acc.area(); rc[b] = cnt; }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < NB - 1; b++) {
                    acc.grow(bb[a][b]);
                    cnt += bc[a][b];
                    if (!(cnt > 0) || !(rc[b + 1] > 0)) continue;
                    const float cost = acc.area() * cnt + ra[b + 1] * rc[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            mid = first + count / 2;
            if (best_axis >= 0) {
                const float l0 = lof[best_axis], sc = scalef[best_axis], tp = (float)(NB - 1);
                auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t c) {
                    alignas(16) float cc[4];
                    _mm_store_ps(cc, cen[c].v);
                    return (int)std::min(std::max((cc[best_axis] - l0) * sc, 0.0f), tp) <= best_bin;
                });
                const uint32_t m2 = (uint32_t)(it - idx.begin());
                if (m2 > first && m2 < first + count) mid = m2;
            }
        }
        return mid;
    }
    // inner nodes in preorder, left first; a single cluster c is the link ~c.  Explicit stack (see Builder::build: ploc_top may be 65,536,
    // and a split that peels off one cluster per level is as deep as that).
    void build(uint32_t first0, uint32_t count0) {
        struct Item { uint32_t first, count; int32_t parent; bool is_right; };
        std::vector<Item> todo;
        todo.push_back(Item{first0, count0, -1, false});
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            int32_t link;
            if (it.count == 1) link = ~(int32_t)idx[it.first];
            else {
                link = (int32_t)left.size();
                left.push_back(0); right.push_back(0);
                const uint32_t mid = split_range(it.first, it.count);
                todo.push_back(Item{mid, it.first + it.count - mid, link, true});
                todo.push_back(Item{it.first, mid - it.first, link, false});
            }
            if (it.parent >= 0) (it.is_right ? right : left)[(size_t)it.parent] = link;
        }
    }
};
}  // namespace

void build_top_tree(const float *boxes, const uint32_t *counts, uint32_t m, std::vector<int32_t> &out_left, std::vector<int32_t> &out_right) {
    out_left.clear(); out_right.clear();
    if (m < 2) return;
    out_left.reserve(m - 1); out_right.reserve(m - 1);
    TopBuilder b(boxes, counts, m, out_left, out_right);
    b.build(0, m);
}

}  // namespace hr
