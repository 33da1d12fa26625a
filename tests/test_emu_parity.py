"""CPU-only tier: the kernels' per-lane code (csrc/*_core.h compiled for the host, tests/emu) against the f64
oracle.  Same tolerances as tests/test_gpu_parity.py — the GPU tier repeats these through the C ABI."""
import numpy as np
import pytest

ATOL_REL = 1e-2
FRAC_OK = 0.995


@pytest.fixture(scope="module")
def emu_scenes(scenes, emu):
    cache = {}

    def get(name):
        if name not in cache:
            sc, o = scenes(name)
            cache[name] = (sc, o, emu.EmuScene(sc.desc_ptr))
        return cache[name]
    return get


def test_device_bvh_shape(emu_scenes):
    _, _, e = emu_scenes("rtcamp6_v3_1")
    st = e.stats()
    # 12,294 triangles in the scene; early split clipping of long thin ones (kept here: it cuts the SAH cost by 10 %) stores
    # some of them as several references
    assert 12294 <= st["tris"] < 2 * 12294 and st["spheres"] == 1 and st["cuboids"] == 1 and st["emitters"] == 1
    assert st["leaves"] * 2 - 1 == st["nodes"] and st["max_depth"] < 40
    _, _, e2 = emu_scenes("spheres")
    assert e2.stats()["spheres"] == 105 and e2.stats()["emitters"] == 5 and e2.stats()["tris"] == 0


@pytest.mark.parametrize("name", ["rtcamp6_v3_1", "cornell_mini", "spheres"])
def test_closest_hit(emu_scenes, name):
    sc, o, e = emu_scenes(name)
    rng = np.random.default_rng(11)
    n = 3000
    eye = np.array(sc.desc.camera.eye.tuple())
    org = eye + rng.normal(size=(n, 3)) * 0.3
    tgt = rng.uniform(-2.5, 2.5, size=(n, 3)) * np.array([1.0, 0.6, 1.0]) + np.array([0, 0.8, 0])
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays32 = np.concatenate([org, d], axis=1).astype(np.float32)
    got, gel = e.intersect(rays32)
    ref, rel = o.intersect(rays32.astype(np.float64))
    same = got[:, 0] == ref[:, 0]
    assert same.mean() > 0.999
    both = same & (ref[:, 0] == 1)
    assert (gel[both] == rel[both]).mean() > 0.998
    ok = both & (gel == rel)
    terr = np.abs(got[ok, 1] - ref[ok, 1]) / np.maximum(1.0, ref[ok, 1])
    assert np.quantile(terr, 0.99) < 2e-5 and terr.max() < 1e-3


def test_axis_aligned_and_degenerate_rays(emu_scenes):
    """Zero direction components give +-inf reciprocals and NaN slab terms (bvh.rs:20-39); results must match."""
    sc, o, e = emu_scenes("cornell_mini")
    rays = np.array([[0.0, 3.0, 2.5, 0, -1, 0], [0.6, 5, 0.4, 0, -1, 0], [-5, 0.25, 0.2, 1, 0, 0], [0.3, 0.2, 7, 0, 0, -1],
                     [0, 0, 0, 0, 1, 0], [-3, 0.0, 0, 1, 0, 0], [10, 10, 10, 0, 1, 0]], dtype=np.float32)
    got, gel = e.intersect(rays)
    ref, rel = o.intersect(rays.astype(np.float64))
    assert np.array_equal(got[:, 0], ref[:, 0].astype(np.float32))
    hit = ref[:, 0] == 1
    assert np.array_equal(gel[hit], rel[hit])
    assert np.allclose(got[hit, 1], ref[hit, 1], rtol=1e-5)


@pytest.mark.parametrize("name,w,h,s", [("rtcamp6_v3_1", 96, 54, 2), ("cornell_mini", 64, 48, 3), ("spheres", 80, 45, 1), ("rtcamp6_dodeca", 50, 29, 1), ("rtcamp6_v3", 72, 40, 2), ("simple", 80, 45, 2),
                                         ("material_examples", 80, 45, 2), ("rtcamp6_v1", 64, 36, 2), ("rtcamp6_v2", 64, 36, 1), ("rtcamp5", 64, 36, 1), ("tbf3", 64, 36, 1)])
def test_radiance_accumulator(emu_scenes, name, w, h, s):
    _, o, e = emu_scenes(name)
    acc, cn = e.render(w, h, 1, s + 1, threads=0)
    ref, rc = o.render(w, h, 1, s + 1, threads=0, counters=True)
    assert np.isfinite(acc).all()
    rel = np.abs(acc.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    assert (rel <= ATOL_REL).mean() >= FRAC_OK
    assert abs(acc.mean() - ref.mean()) <= 2e-3 * max(1.0, ref.mean())
    # path structure: same number of rays within 0.1 % (an fp32 branch flip changes a path's length)
    ref_rays = rc["rays_primary"] + rc["rays_bounce"] + rc["rays_shadow"]
    assert cn["paths"] == rc["paths"] == w * h * 4 * s
    # (the shadow rays nee_setup knows to add nothing are scene.intersect calls of the reference that the per-lane code does not make)
    assert abs(cn["rays"] + cn["shadow_culled"] - ref_rays) <= 1e-3 * ref_rays
    # on the mesh scenes the device tree + culling must do less work than the reference-order walk it replaces
    if name.startswith("rtcamp6"):
        assert cn["node_tests"] * 1.5 < rc["mesh_node_tests"] + rc["top_node_tests"] and cn["tri_tests"] * 3 < rc["tri_tests"]


def test_ragged_and_tiny_resolutions(emu_scenes):
    _, o, e = emu_scenes("cornell_mini")
    for w, h in [(1, 1), (5, 3), (4, 9), (17, 2)]:
        acc, _ = e.render(w, h, 2, 4, threads=1)
        ref, _ = o.render(w, h, 2, 4, threads=1)
        rel = np.abs(acc - ref) / np.maximum(1.0, np.abs(ref))
        assert (rel <= ATOL_REL).mean() >= 0.98, (w, h)


def test_post_chain(emu_scenes, emu, orc):
    _, o, _ = emu_scenes("cornell_mini")
    for (w, h) in [(64, 40), (8, 5), (1, 1), (3, 1), (1, 4)]:
        ref, _ = o.render(w, h, 1, 3, threads=0)
        a32 = ref.astype(np.float32)
        img = emu.resolve(a32, 2)
        exp = orc.resolve(a32.astype(np.float64), 2)
        diff = np.abs(img.astype(int) - exp.astype(int))
        assert diff.max() <= 1
        assert (diff == 0).mean() > 0.99 or w * h < 50
    # saturation and zeros
    acc = np.zeros((4, 4, 3), dtype=np.float32)
    acc[1, 1] = 1e6
    assert np.array_equal(emu.resolve(acc, 1), orc.resolve(acc.astype(np.float64), 1))


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_debug_renderer(emu_scenes, mode):
    """DebugRenderer modes (renderer.rs:101-146): Shading / Normal / Depth / FocalPlane, pinhole rays, no RNG."""
    _, o, e = emu_scenes("rtcamp6_v3_1")
    w, h = 64, 36
    got = e.render_debug(w, h, mode).astype(np.float64)
    ref = o.render_debug(w, h, mode)
    d = np.abs(got - ref)
    tol = 2e-3 * np.maximum(1.0, np.abs(ref))
    assert (d <= tol).mean() > 0.99        # silhouette pixels may resolve to a different primitive in fp32
    assert abs(got.mean() - ref.mean()) < 2e-3 * max(1.0, abs(ref.mean()))


# ---- LBVH: the device BVH builder's per-thread code (csrc/lbvh_core.h) run sequentially on the host ----

def _random_rays(sc, n, seed):
    rng = np.random.default_rng(seed)
    eye = np.array(sc.desc.camera.eye.tuple())
    org = eye + rng.normal(size=(n, 3)) * 0.3
    tgt = rng.uniform(-2.5, 2.5, size=(n, 3)) * np.array([1.0, 0.6, 1.0]) + np.array([0, 0.8, 0])
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([org, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("builder", [1, 2])
@pytest.mark.parametrize("name,max_leaf", [("rtcamp6_v3_1", 4), ("rtcamp6_v3_1", 1), ("cornell_mini", 4), ("spheres", 2), ("rtcamp6_dodeca", 8)])
def test_lbvh_closest_hit_is_tree_independent(scenes, emu, emu_scenes, name, max_leaf, builder):
    """The closest hit must not depend on the tree: the device builders' trees (1 = LBVH, 2 = PLOC) and the host-SAH tree over
    the same fp32 primitives return the same element and the same t bit for bit (the primitive tests are the same code), on
    the 32-byte records and on the 16-byte quantised ones, and agree with the f64 oracle."""
    sc, o, e_sah = emu_scenes(name)
    emu.set_build_options(max_leaf=max_leaf, builder=builder)
    try:
        e = emu.EmuScene(sc.desc_ptr)
    finally:
        emu.set_build_options()
    st, st0 = e.stats(), e_sah.stats()
    n = st["tris"] + st["spheres"] + st["cuboids"]
    # (the host tree may hold long thin triangles as several split references; the device LBVH holds every triangle once)
    assert st["tris"] <= st0["tris"] and (st["spheres"], st["cuboids"]) == (st0["spheres"], st0["cuboids"])
    assert st["nodes"] == 2 * st["leaves"] - 1 and st["leaves"] >= (n + max_leaf - 1) // max_leaf and st["max_depth"] < 64
    rays = _random_rays(sc, 4000, 23)
    got, gel = e.intersect(rays)
    ref, rel = e_sah.intersect(rays)
    emu.set_walk_mode(2)                     # the trace kernel's walk on the quantised records of the device-built tree
    try:
        gotq, gelq = e.intersect(rays)
    finally:
        emu.set_walk_mode(0)
    assert np.array_equal(gotq, got) and np.array_equal(gelq, gel)
    assert np.array_equal(got[:, 0], ref[:, 0])
    hit = ref[:, 0] == 1
    # equal-t ties between adjacent triangles may resolve to either one; everything else is identical
    assert np.array_equal(got[hit, 1], ref[hit, 1])
    assert (gel[hit] == rel[hit]).mean() > 0.999
    oref, oel = o.intersect(rays.astype(np.float64))
    assert (got[:, 0] == oref[:, 0]).mean() > 0.999


@pytest.mark.parametrize("builder", [1, 2])
def test_lbvh_radiance(scenes, emu, orc, builder):
    sc, o = scenes("cornell_mini")
    emu.set_build_options(builder=builder)
    try:
        e = emu.EmuScene(sc.desc_ptr)
    finally:
        emu.set_build_options()
    acc, _ = e.render(64, 48, 1, 3, threads=0)
    ref, _ = o.render(64, 48, 1, 3, threads=0, counters=True)
    assert np.isfinite(acc).all()
    err = np.abs(acc - ref) / np.maximum(1.0, np.abs(ref))
    assert (err < ATOL_REL).mean() > FRAC_OK


def _triangle_scene(ha, verts, faces):
    """One mesh element with the given triangles, camera / skybox / images borrowed from cornell_mini."""
    import ctypes as C
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    faces = np.ascontiguousarray(faces, dtype=np.uint64)
    base = ha.Scene("cornell_mini")
    el = (ha.Element * 1)()
    el[0].kind = ha.MESH
    el[0].material.albedo.color = ha.Vec3(0.7, 0.7, 0.7)
    el[0].material.albedo.image = el[0].material.emission.image = el[0].material.roughness.image = -1
    el[0].vertexes = verts.ctypes.data_as(C.POINTER(ha.Vec3))
    el[0].num_vertexes = verts.shape[0]
    el[0].faces = faces.ctypes.data_as(C.POINTER(C.c_uint64))
    el[0].num_faces = faces.shape[0]
    d = ha.SceneDesc()
    C.memmove(C.byref(d), base.desc_ptr, C.sizeof(d))
    d.elements = C.cast(el, C.POINTER(ha.Element))
    d.num_elements = 1
    return d, (base, el, verts, faces)


# rays against the unit right triangle (0,0,0) (1,0,0) (0,1,0) (+ a second one behind it and a degenerate one): every number is exact
# in fp32 and f64, so the triangle test's boundary rules (bvh.rs:266-290) can be checked to the bit.  The rays are slightly tilted
# (direction (2^-10, 2^-10, -1), not normalised — the triangle test does not care): an axis-parallel ray that runs exactly IN a face
# of a node's box is a miss in the reference whatever the triangle test says (0 x inf = NaN in its slab test, bvh.rs:20-39), and the
# triangle's legs lie in faces of its leaf box.
_D = 2.0 ** -10


def _tilted(tx, ty, back=False):
    """the ray that reaches (tx, ty, 0) at t = 1"""
    d = np.array([_D, _D, 1.0 if back else -1.0])
    return list(np.array([tx, ty, 0.0]) - d) + list(d)


TRI_EDGE_RAYS = np.array([
    _tilted(0.25, 0.25),           # 0 interior
    _tilted(0.5, 0.0),             # 1 on the edge v = 0: accepted (v >= 0)
    _tilted(0.0, 0.5),             # 2 on the edge u = 0
    _tilted(0.5, 0.5),             # 3 on the hypotenuse: u + v == 1 is accepted
    _tilted(0.0, 0.0),             # 4 the vertex v0
    _tilted(1.0, 0.0),             # 5 the vertex v1 (u == 1)
    _tilted(0.75, 0.75),           # 6 beyond the hypotenuse: the second triangle, one unit further (t = 2)
    _tilted(-0.25, 0.5),           # 7 u < 0, and outside the far triangle too
    _tilted(0.25, 0.25, True),     # 8 from behind: two-sided
    [0.25, 0.25, 0, _D, _D, -1],   # 9 origin on the triangle: t == 0 is a hit (t >= 0)
    [0.25, 0.25, 1, _D, _D, 1],    # 10 pointing away: t < 0
    [-1.0, 0.25, 0.0, 1, _D, 0],   # 11 in the triangle's plane: det == 0 rejects (bvh.rs:271)
    _tilted(3.0, 3.0),             # 12 over the degenerate triangle only: no hit
], dtype=np.float32)


def _triangle_edge_geometry():
    verts = [[0, 0, 0], [1, 0, 0], [0, 1, 0],              # the unit right triangle in z = 0
             [0, 0, -1], [2, 0, -1], [0, 2, -1],            # a larger one in z = -1 behind it
             [3, 3, 0], [3, 3, 0], [3, 3, 0]]               # a degenerate one (three equal vertices)
    return verts, [[0, 1, 2], [3, 4, 5], [6, 7, 8]]


def test_triangle_test_boundary_rules(ha, emu, orc):
    """Edges, vertices, t == 0, det == 0, a degenerate triangle: the derived-record triangle test (pt_core.h tri_test on TriT) decides
    exactly as the reference's Cramer's rule (oracle, f64) on geometry whose numbers are exact in both."""
    import ctypes as C
    verts, faces = _triangle_edge_geometry()
    d, keep = _triangle_scene(ha, verts, faces)
    e = emu.EmuScene(C.addressof(d))
    o = orc.OracleScene(C.addressof(d))
    got, gel = e.intersect(TRI_EDGE_RAYS)
    ref, rel = o.intersect(TRI_EDGE_RAYS.astype(np.float64))
    assert np.array_equal(got[:, 0], ref[:, 0].astype(np.float32)), (got[:, 0], ref[:, 0])
    expect_hit = np.array([1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 0, 0, 0], dtype=np.float32)
    assert np.array_equal(got[:, 0], expect_hit)
    hit = expect_hit == 1
    assert np.array_equal(got[hit, 1], ref[hit, 1].astype(np.float32))      # distances: 1, ..., 2 (the far triangle), 1, 0 — exact
    assert got[6, 1] == 2.0 and got[9, 1] == 0.0


def _scaled_triangle_soup(seed=5):
    """Triangles of edge length 1e-3 .. 1e3 scattered up to 1e3 from the origin, and rays aimed at interior points and at points
    outside them.  Returns verts, faces, rays (fp32, shape (n, 6)), expect_hit."""
    rng = np.random.default_rng(seed)
    verts, faces, rays, expect = [], [], [], []
    for k in range(96):
        size = 10.0 ** rng.uniform(-3, 3)
        centre = rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(0, 3)
        e1 = rng.normal(size=3); e1 *= size / np.linalg.norm(e1)
        e2 = rng.normal(size=3); e2 -= e1 * (e2 @ e1) / (e1 @ e1) * 0.5; e2 *= size * rng.uniform(0.3, 1.0) / np.linalg.norm(e2)
        v0 = centre - (e1 + e2) / 3.0
        base = len(verts)
        verts += [v0, v0 + e1, v0 + e2]
        faces.append([base, base + 1, base + 2])
        n = np.cross(e1, e2); n /= np.linalg.norm(n)
        for inside in (True, False):
            u, v = (rng.uniform(0.15, 0.4), rng.uniform(0.15, 0.4)) if inside else (rng.uniform(0.7, 0.9), rng.uniform(0.5, 0.9))
            target = v0 + u * e1 + v * e2
            d = n * rng.choice([-1.0, 1.0]) + 0.3 * rng.normal(size=3)
            d /= np.linalg.norm(d)
            o = target - d * size * rng.uniform(2.0, 20.0)
            rays.append(np.concatenate([o, d]))
            expect.append(inside)
    return np.array(verts), np.array(faces), np.array(rays, dtype=np.float32), np.array(expect)


def _hit_scale(verts, faces, rays, t_ref):
    """|o| + |o - v0| + |e1| + |e2| + t of the triangle each ray hits first (brute force over the soup, f64)."""
    v0 = verts[faces[:, 0]]; e1 = verts[faces[:, 1]] - v0; e2 = verts[faces[:, 2]] - v0
    n = np.cross(e1, e2)
    nn = (n * n).sum(axis=1)
    out = np.empty(len(rays))
    for i, r in enumerate(rays):
        o, d = r[:3], r[3:]
        den = n @ d
        with np.errstate(divide="ignore", invalid="ignore"):
            t = -((o - v0) * n).sum(axis=1) / den
            p = o + t[:, None] * d - v0
            u = (np.cross(p, e2) * n).sum(axis=1) / nn
            v = (np.cross(e1, p) * n).sum(axis=1) / nn
        ok = (den != 0) & (t >= 0) & (u >= 0) & (v >= 0) & (u + v <= 1)
        k = np.where(ok)[0][np.argmin(t[ok])]
        assert abs(t[k] - t_ref[i]) <= 1e-9 * max(1.0, t_ref[i])
        out[i] = np.linalg.norm(o) + np.linalg.norm(o - v0[k]) + np.linalg.norm(e1[k]) + np.linalg.norm(e2[k]) + t[k]
    return out


def test_triangle_test_across_scales(ha, emu, orc):
    """The derived triangle record (unit normal, barycentric gradients of magnitude 1 / size) over six decades of triangle size and
    three of distance from the origin: the same hits as the f64 oracle for rays well inside / well outside, distances to a few fp32
    roundings of |origin| + t."""
    import ctypes as C
    verts, faces, rays, expect = _scaled_triangle_soup()
    d, keep = _triangle_scene(ha, verts, faces)
    e = emu.EmuScene(C.addressof(d))
    o = orc.OracleScene(C.addressof(d))
    got, gel = e.intersect(rays)
    ref, rel = o.intersect(rays.astype(np.float64))
    # a ray may hit ANOTHER triangle of the soup first or instead: the oracle says what is right, `expect` only that the set-up works
    assert (ref[:, 0] == 1)[expect].mean() > 0.95
    assert np.array_equal(got[:, 0], ref[:, 0].astype(np.float32))
    hit = ref[:, 0] == 1
    assert np.array_equal(gel[hit], rel[hit])
    # fp32 arithmetic on the ray and the triangle: the distance is good to a few roundings of the largest quantity involved — the
    # origin, the origin's offset from the v0 of the triangle that is hit (a large triangle's v0 may be far away), its edges, t
    scale = _hit_scale(verts, faces, rays[hit].astype(np.float64), ref[hit, 1])
    terr = np.abs(got[hit, 1] - ref[hit, 1]) / scale
    assert terr.max() < 2e-6 and np.quantile(terr, 0.9) < 1e-7, (terr.max(), np.quantile(terr, 0.9))   # measured 6.9e-7 / 2.9e-8


@pytest.mark.parametrize("builder", [1, 2])
def test_lbvh_single_primitive(ha, emu, builder):
    """n = 1: no internal node, the lone leaf is the root."""
    import ctypes as C
    el = (ha.Element * 1)()
    el[0].kind = 0
    el[0].center = ha.Vec3(0.0, 0.0, -3.0)
    el[0].radius = 1.0
    el[0].material.albedo.image = el[0].material.emission.image = el[0].material.roughness.image = -1
    base = ha.Scene("cornell_mini")           # borrow images / skybox / camera
    d = ha.SceneDesc()
    C.memmove(C.byref(d), base.desc_ptr, C.sizeof(d))
    d.elements = C.cast(el, C.POINTER(ha.Element))
    d.num_elements = 1
    emu.set_build_options(builder=builder)
    try:
        e = emu.EmuScene(C.addressof(d))
    finally:
        emu.set_build_options()
    assert e.stats()["nodes"] == 1
    got, gel = e.intersect(np.array([[0, 0, 0, 0, 0, -1], [0, 0, 0, 0, 1, 0]], dtype=np.float32))
    assert got[0, 0] == 1 and abs(got[0, 1] - 2.0) < 1e-6 and gel[0] == 0 and got[1, 0] == 0


@pytest.mark.parametrize("name", ["rtcamp6_v3_1", "spheres", "cornell_mini"])
def test_postponed_leaf_walk_gives_the_same_hits(emu, emu_scenes, name):
    """The trace kernel lets a lane keep walking with one leaf parked (trace_node<.., SPEC>): boxes are then culled against a
    closest hit that lags by one leaf, leaves are still tested in walk order — hits, t and elements must be identical."""
    sc, _, e = emu_scenes(name)
    rays = _random_rays(sc, 3000, 31)
    ref, rel = e.intersect(rays)
    emu.set_walk_mode(1)
    try:
        got, gel = e.intersect(rays)
    finally:
        emu.set_walk_mode(0)
    assert np.array_equal(got, ref) and np.array_equal(gel, rel)


@pytest.mark.parametrize("name", ["rtcamp6_v3_1", "rtcamp6_dodeca", "spheres", "cornell_mini", "rtcamp5", "rtcamp6_v2"])
def test_quantised_nodes_give_the_same_hits(emu, emu_scenes, name):
    """The trace kernel's 16-byte nodes (device_scene.h QNode): planes on a 16-bit grid, rounded outward, links implied by the
    per-octant preorder.  Boxes that only grow can add node visits, never lose a hit: hits, t and elements are bit-identical
    to the walk on the 32-byte fp32 records of the same tree; the fatter boxes add ~0.1 % of visits, the end-of-walk sentinel
    record at most one per ray."""
    sc, _, e = emu_scenes(name)
    rays = _random_rays(sc, 6000, 41)
    # axis-parallel and grazing rays as well: zero direction components make the grid-space FMA form produce inf - inf
    extra = np.array([[0.1, 3.0, 0.2, 0, -1, 0], [0.1, 0.5, 6.0, 0, 0, -1], [-6.0, 0.4, 0.1, 1, 0, 0], [0.0, 1e-3, 5.0, 0, 0, -1],
                      [0.3, 2.0, 0.3, 0.0, -0.6, -0.8], [0.3, 2.0, 0.3, 0.8, -0.6, 0.0]], dtype=np.float32)
    rays = np.concatenate([rays, extra])
    emu.set_walk_mode(1)
    try:
        ref, rel = e.intersect(rays)
        plain = emu.last_node_tests()
        emu.set_walk_mode(2)
        got, gel = e.intersect(rays)
        quant = emu.last_node_tests()
    finally:
        emu.set_walk_mode(0)
    assert np.array_equal(got, ref) and np.array_equal(gel, rel)
    print("%s: node tests plain %d, quantised %d (+%.2f %%)" % (name, plain, quant, 100.0 * (quant - plain) / plain))
    assert plain <= quant <= 1.01 * plain + len(rays)


def test_device_builders_tree_quality(scenes, emu, emu_scenes):
    """Node tests per ray of the device-built trees against the host SAH tree (same rays): the Morton-median LBVH costs about 1.4x,
    the agglomerative PLOC build must stay within 15 %."""
    out = {}
    for name in ("rtcamp6_v3_1", "rtcamp6_dodeca"):
        sc, _, e_sah = emu_scenes(name)
        rays = _random_rays(sc, 6000, 57)
        emu.set_walk_mode(1)
        try:
            e_sah.intersect(rays)
            counts = [emu.last_node_tests()]
            for builder in (1, 2):
                emu.set_build_options(builder=builder)
                try:
                    e = emu.EmuScene(sc.desc_ptr)
                finally:
                    emu.set_build_options()
                e.intersect(rays)
                counts.append(emu.last_node_tests())
        finally:
            emu.set_walk_mode(0)
        out[name] = counts
        print("%s: node tests host SAH %d, LBVH %d (x%.2f), PLOC %d (x%.2f)" % (name, counts[0], counts[1], counts[1] / counts[0], counts[2], counts[2] / counts[0]))
    for name, (sah, lb, pl) in out.items():
        assert pl <= 1.15 * sah and pl < lb, (name, sah, lb, pl)


@pytest.mark.parametrize("name,w,h", [("cornell_mini", 96, 64), ("rtcamp6_v3_1", 160, 90), ("spheres", 128, 72)])
def test_per_path_event_log(emu_scenes, name, w, h):
    """The per-path accounting of the GPU tier (test_per_path_parity_accounting) on the host emulation of the same per-lane code
    (path_advance<.., LOG> of pt_core.h): the event log's encoding agrees with the oracle's, path by path — the same events, the same
    elements and mesh triangles, the same number of rays — for all but a few paths per 100,000, and the logged radiances are the
    emulated render's."""
    import path_parity
    _, o, e = emu_scenes(name)
    g = e.path_log(w, h, 1)
    r = o.path_log(w, h, 1)
    acc, cn = e.render(w, h, 1, 2)
    assert cn["rays"] + cn["shadow_culled"] == int(g[1].sum())                                        # the log's ray count is the counters' ray count
    np.testing.assert_allclose(g[0].astype(np.float64).sum(axis=2), acc, rtol=1e-5, atol=1e-6)
    ref, ocn = o.render(w, h, 1, 2, counters=True)
    assert ocn["rays_primary"] + ocn["rays_bounce"] + ocn["rays_shadow"] == int(r[1].sum())
    np.testing.assert_allclose(r[0].sum(axis=2), ref, rtol=1e-12, atol=1e-12)
    a = path_parity.account(g, r)
    sb = a["same_branch"]
    assert sb["rays_equal"] and a["divergent_ppm"] <= 150.0, a
    assert (r[2][..., 9] == g[2][..., 9])[(g[3] == r[3]) & (g[2][..., :9] == r[2][..., :9]).all(axis=-1)].all()   # sphere-hit counts of same-branch paths
    first = r[2][..., 0] & 7
    assert (first != 0).all() and set(np.unique(first)) <= {1, 2, 3, 4, 5, 6, 7}                                  # every path has a first event
    if name != "spheres":
        assert sb["max_rel_floor1"] <= 1e-3, sb                                  # no sphere chains: same branches, same radiance
    else:
        assert sb["no_sphere_bounce"]["over_1e-3_floor1_ppm"] == 0.0 and sb["over_1e-3_floor1_ppm"] <= 1500.0, sb
        assert all(int(k) >= 2 for k in sb["over_1e-3_by_sphere_bounces_ppm"]), sb  # the tail needs at least two sphere bounces


@pytest.mark.parametrize("name", ["rtcamp6_v3_1", "rtcamp6_v2", "tbf3", "rtcamp5", "spheres", "cornell_mini", "material_examples"])
def test_nee_culls_do_not_change_a_bit(emu, emu_scenes, name):
    """The shadow rays nee_setup does not trace (sample on the emitter's far side, GGX below the horizon) are
    rays the reference traces and discards (renderer.rs:279-280): with the shortcuts off the per-lane code renders the same accumulator,
    bit for bit, from more rays."""
    _, _, e = emu_scenes(name)
    try:
        emu.set_nee_cull(True)
        a, ca = e.render(96, 54, 1, 3)
        la = e.path_log(96, 54, 1)
        emu.set_nee_cull(False)
        b, cb = e.render(96, 54, 1, 3)
        lb = e.path_log(96, 54, 1)
    finally:
        emu.set_nee_cull(True)
    assert np.array_equal(a, b)
    assert cb["shadow_culled"] == 0 and ca["shadow_culled"] > 0 and ca["rays"] + ca["shadow_culled"] == cb["rays"]
    for x, y in zip(la, lb):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("name", ["rtcamp6_v3_1", "rtcamp6_v2", "tbf3", "rtcamp5", "spheres", "cornell_mini", "material_examples", "simple"])
def test_split_pipeline_is_the_same_arithmetic(emu, emu_scenes, name):
    """csrc/wf_core.h — the split pipeline's per-lane code (option trace_mode 1: a traversal kernel and a shading kernel per path iteration,
    the NEE weight computed when the shadow ray is emitted, the contributions added a step later) — is path_advance cut into pieces: driven
    path by path on the host it renders the accumulator of the megakernel's per-lane code, bit for bit.  (On the GPU:
    test_kernel_variants_render_the_same_bits.)"""
    _, _, e = emu_scenes(name)
    a, _ = e.render(96, 54, 1, 3)
    b = e.render_wf(96, 54, 1, 3)
    assert a.sum() > 0 and np.array_equal(a, b)
    try:
        emu.set_nee_cull(False)
        c = e.render_wf(96, 54, 1, 3)
    finally:
        emu.set_nee_cull(True)
    assert np.array_equal(a, c)


@pytest.mark.parametrize("name,fp32_at_least,precise_at_most", [("rtcamp6_v2", 400.0, 0.0), ("spheres", 20.0, 0.0), ("material_examples", 40.0, 0.0)])
def test_precise_shading_closes_the_same_branch_tail(emu, emu_scenes, name, fp32_at_least, precise_at_most):
    """Option precise_shading on the host: csrc/wf_core.h wf_surface_f64 — hit distance, hit point, normal, mirror / Snell / Fresnel and the
    sampled lobe directions in f64 from the reference's f64 draws (the record's fp32 value + its residual in the record's twin, as the seed
    kernel's RecordTail<.., LO> writes them), the ray carried as fp32 + residual — driven path by path against the oracle's path log.  The
    paths that take the oracle's branches and still differ by more than 1e-3 (refraction chains through faceted glass, bounces off r = 0.1
    spheres) vanish — none left at this size, the worst same-branch path below 2e-4 —, and fewer paths diverge; ray counts stay equal.
    Without the residuals (emu.set_draw_residuals(False): precise shading on the fp32 draws alone) a tail remains: that is what the
    rounding of the draws costs.  (GPU: test_per_path_parity_accounting_precise_shading.)"""
    import path_parity
    sc, o, e = emu_scenes(name)
    w, h = 128, 72
    ref = o.path_log(w, h, 1)
    a32 = path_parity.account(e.path_log(w, h, 1), ref)
    lw = e.path_log_wf(w, h, 1)
    a64 = path_parity.account(lw, ref)
    try:        # the megakernel's per-lane code with PREC (path_advance<.., PREC>: residuals parked in the path's record): the same log, word for word
        emu.set_precise(True)
        for x, y in zip(lw, e.path_log(w, h, 1)):
            assert np.array_equal(x, y)
    finally:
        emu.set_precise(False)
    s32, s64 = a32["same_branch"], a64["same_branch"]
    print("%s: fp32 divergent %.0f ppm, same-branch beyond 1e-3 %.0f ppm (worst %.3g); precise %.0f / %.0f ppm (worst %.3g)" % (
        name, a32["divergent_ppm"], s32["over_1e-3_floor1_ppm"], s32["max_rel_floor1"], a64["divergent_ppm"], s64["over_1e-3_floor1_ppm"], s64["max_rel_floor1"]))
    assert s64["rays_equal"] and s32["over_1e-3_floor1_ppm"] >= fp32_at_least and s64["over_1e-3_floor1_ppm"] <= precise_at_most
    assert s64["max_rel_floor1"] <= 2e-4, s64
    if name == "spheres":
        try:
            emu.set_draw_residuals(False)
            lone = path_parity.account(e.path_log_wf(w, h, 1), ref)["same_branch"]
        finally:
            emu.set_draw_residuals(True)
        print("spheres, precise shading on the fp32 draws alone: beyond 1e-4 %.0f ppm (with the residuals %.0f), worst %.3g" % (lone["over_1e-4_floor1_ppm"], s64["over_1e-4_floor1_ppm"], lone["max_rel_floor1"]))
        assert s64["over_1e-4_floor1_ppm"] == 0.0 and lone["over_1e-4_floor1_ppm"] >= 100.0 and lone["max_rel_floor1"] > 10.0 * s64["max_rel_floor1"]
    assert a64["divergent_ppm"] <= a32["divergent_ppm"] + 30.0 and s64["over_1e-4_floor1_ppm"] <= s32["over_1e-4_floor1_ppm"]
    assert abs(a64["mean_radiance"]["gpu"] - a64["mean_radiance"]["oracle"]) <= 1e-3 * a64["mean_radiance"]["oracle"]


def test_ggx_nee_sample_exactly_at_the_horizon_adds_nothing(emu, emu_scenes):
    """Round 6's second non-finite pixel, found by rendering every scene at 1920x1080 x 1,024 samplings: cornell_mini, sampling 732, pixel
    (1394, 371), sub-sample 2 — a GGX hit on a horizontal cuboid face whose NEE sample on the second emitter lies, after fp32 rounding, at
    EXACTLY the face's height: l.n = +0, material.rs:64-67 lets +0 through (it tests the sign), the Smith term is 0 and the denominator
    4 (l.n)(v.n) too: 0 / 0.  In the reference's f64 the case does not occur (l.n ~ 1e-9 there: the term is ~0).  bsdf_eval returns the limit, 0;
    the pixel then equals the oracle's."""
    _, o, e = emu_scenes("cornell_mini")
    w, h, x, y, s = 1920, 1080, 1394, 371, 732
    ref = o.render_region(w, h, x, y, 1, 1, s, s + 1, threads=1)[0, 0]
    for precise in (False, True):
        try:
            emu.set_precise(precise)
            subs = np.asarray([e.one_path(w, h, x, y, sub, s) for sub in range(4)], dtype=np.float64)
        finally:
            emu.set_precise(False)
        assert np.isfinite(subs).all() and subs[2].min() > 0.0, subs
        assert np.abs(subs.sum(axis=0) - ref).max() <= (1e-6 if precise else 1e-5), (precise, subs.sum(axis=0), ref)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_scenes_path_by_path(ha, orc, emu, seed):
    """Fuzz tier (tests/random_scenes.py): every element kind x surface type x textured / constant parameters, overlapping and nested —
    combinations the reference's scenes do not contain.  The host emulation of the per-lane code against the oracle, path by path: the
    same rays per path on every same-branch path, few divergent paths, no systematic difference."""
    import path_parity
    import random_scenes
    sc = random_scenes.build(ha, seed)
    o = orc.OracleScene(sc.desc_ptr)
    e = emu.EmuScene(sc.desc_ptr)
    w, h = 96, 54
    a = path_parity.account(e.path_log(w, h, 1), o.path_log(w, h, 1))
    sb = a["same_branch"]
    print("random scene %d: divergent %.0f ppm %s, same-branch over 1e-3 %.0f ppm, max %.3g, mean %.6g / %.6g" % (
        seed, a["divergent_ppm"], a["divergent_by_class_ppm"], sb["over_1e-3_floor1_ppm"], sb["max_rel_floor1"], a["mean_radiance"]["gpu"], a["mean_radiance"]["oracle"]))
    assert sb["rays_equal"] and a["divergent_ppm"] <= 3000.0 and sb["over_1e-3_floor1_ppm"] <= 3000.0, a
    assert abs(a["mean_radiance"]["gpu"] - a["mean_radiance"]["oracle"]) <= 5e-3 * a["mean_radiance"]["oracle"], a["mean_radiance"]
    acc, _ = e.render(w, h, 1, 3)
    ref, _ = o.render(w, h, 1, 3)
    rel = np.abs(acc - ref) / np.maximum(1.0, np.abs(ref))
    assert np.isfinite(acc).all() and (rel <= 1e-2).mean() >= 0.997, (rel <= 1e-2).mean()


def test_non_finite_scene_input_is_rejected(ha, emu):
    """flatten_scene (shared by the HIP library and this emulation): a NaN or an infinity in the geometry or the camera is HR_ERR_INVALID
    at upload — the reference would panic in its BVH build (partial_cmp().unwrap(), bvh.rs) — instead of reaching a builder's comparator."""
    import random_scenes
    import ctypes as C
    ok = random_scenes.build(ha, 5, spheres=3, cuboids=2, meshes=1)
    emu.EmuScene(ok.desc_ptr)                          # the untouched scene builds

    def rejected(mutate):
        sc = random_scenes.build(ha, 5, spheres=3, cuboids=2, meshes=1)
        d = C.cast(sc.desc_ptr, C.POINTER(ha.SceneDesc)).contents
        mutate(d)
        with pytest.raises(RuntimeError):
            emu.EmuScene(sc.desc_ptr)

    def first(d, kind):
        return next(d.elements[i] for i in range(d.num_elements) if d.elements[i].kind == kind)

    def nan_vertex(d):
        first(d, ha.MESH).vertexes[2].y = float("nan")
    rejected(lambda d: setattr(first(d, ha.SPHERE), "radius", float("inf")))
    rejected(lambda d: setattr(first(d, ha.SPHERE).center, "x", float("nan")))
    rejected(lambda d: setattr(first(d, ha.CUBOID).aabb_max, "z", float("-inf")))
    rejected(nan_vertex)
    rejected(lambda d: setattr(d.camera.eye, "y", float("nan")))
    rejected(lambda d: setattr(d.camera, "focus_distance", float("inf")))
