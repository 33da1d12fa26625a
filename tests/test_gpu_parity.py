"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of
include/hanamaru_hip.h; the f64 oracle is the checker.

Tolerances (stated once, used everywhere):
  * integer work (ISAAC-64 outputs, seed words): bit-exact.
  * fp32 draws handed to the trace kernel: equal to the oracle's f64 draws rounded once to fp32.
  * closest-hit queries: same element; |t_gpu - t_ref| <= 2e-5 * max(1, t_ref) for 99 % (max 1e-3: grazing spheres); normals: median error < 1e-5 (meshes) /
    1e-4 (r = 0.1 spheres five units away: fp32 position error over the radius), 99.9 % < 2e-3.
  * radiance accumulator, per channel (GATES; round 3 raised them after the f64 sphere test and the GGX half-vector fix): |gpu - oracle| <= 1e-2 * max(1, |oracle|) and <= 1e-3 * max(1, |oracle|) for the
    fractions of GATES below — per scene, set just under what is measured and, since round 4, CHECKED against what the per-path
    accounting predicts (test_per_path_parity_accounting: 4 - 320 paths per million take another branch than the f64 oracle —
    Fresnel coin, hit / miss at a silhouette, the neighbouring triangle — and decorrelate; in the scenes full of small spheres or
    refracting diamonds a larger number stays on the oracle's branches and still drifts by more than 1e-3, amplified bounce by
    bounce), and the image mean agrees to 2e-3 relative (5e-3 at full size against full-size oracle crops).
  * per path (hr_debug_path_log vs orc_path_log): same-branch paths trace the same number of rays; divergent paths and same-branch
    outliers per million below PATH_LIMITS; in scenes without small spheres no same-branch path off by more than 1e-3.
  * 8-bit image after the post chain, fed the SAME accumulator: <= 1 LSB on every channel, > 99 % exact.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL_REL = 1e-2
# scene -> (least fraction of channels within 1e-2, within 1e-3); measured values in profiles/r03_parity_report.json and
# profiles/r03_parity_lines.txt.  Round 3 (f64 sphere test and normal, cancellation-free GGX half vector) moved the three hard
# scenes: spheres 0.9898 -> 0.9955 at this size, rtcamp5 0.9881 -> 0.9980, rtcamp6_v2 0.9947 -> 0.9974.  What is left in the
# sphere scenes are mirror spheres of radius 0.1: a reflection multiplies the fp32 rounding of the incoming ray (6e-8) by
# ~2 x distance / radius = 100, so a path that meets two of them looks up the sky a fraction of a texel away from where the f64
# reference does — a property of fp32 rays, not of a test; the more samplings a pixel holds, the likelier it holds such a path
# (64 samplings at full size: 0.991, hence the crop gate below).
# Round 4 replaced the explanation by a measurement (test_per_path_parity_accounting, profiles/r04_parity_report.json "per_path"): paths that
# take another branch than the oracle's are 7 - 300 per million; the larger part of the sphere scenes' tail took the SAME branches and
# bounced off two or more small spheres.  The sphere hit point / normal from the f64 root and, for primary rays, from the f64 camera ray
# (pt_core.h sphere_surface, path_start) cut that tail by 3.3 (spheres: 1,804 -> 556 ppm of the paths); the gates moved up with it:
# spheres 0.9950 -> 0.9975 (-> 0.9993 once the sphere centres were f64 too: an fp32 centre is off by 3e-8 |c|, the normal of an r = 0.1 sphere by ten
# times that — the largest term left; spheres' same-branch tail 556 -> 102 ppm, no divergent path left), rtcamp6_v2 0.9955 -> 0.9965, rtcamp5 0.9965 -> 0.9972, tbf3 0.9975 -> 0.9980, material_examples 0.9985 -> 0.9990.
GATES = {
    "rtcamp6_v3_1": (0.9998, 0.9995), "rtcamp6_dodeca": (0.9996, 0.9992), "rtcamp6_v3": (0.9998, 0.9995), "rtcamp6_v1": (0.9996, 0.9994),
    "material_examples": (0.9997, 0.9990), "simple": (0.9998, 0.9995), "cornell_mini": (0.9997, 0.9995), "tbf3": (0.9995, 0.9980),
    "rtcamp6_v2": (0.9990, 0.9965), "spheres": (0.9997, 0.9993), "rtcamp5": (0.9990, 0.9972),
}
# the same with option precise_shading (f64 bounce geometry, the reference's f64 draws, roughness maps at f64 coordinates).  Measured at these sizes
# (profiles/r06_precise_tests_gpu.txt): every scene 0.99995 - 1.00000 within 1e-3 except rtcamp5 (0.99982: divergent paths — other element at a
# silhouette, GGX sample below the horizon); the gates sit at about three times the measured deficit.
GATES_PRECISE = {k: (0.9998, 0.9997) for k in GATES}
GATES_PRECISE.update({"rtcamp5": (0.9997, 0.9994), "rtcamp6_v1": (0.9998, 0.9996)})
CROP_SLACK = (0.0015, 0.006)
FRAC_OK = 0.9995   # the headline scene's gate, for the tests that render rtcamp6_v3_1


def _fractions(acc, ref):
    rel = np.abs(acc.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    return float((rel <= 1e-2).mean()), float((rel <= 1e-3).mean())


def _compare(acc, ref):
    return _fractions(acc, ref)[0], float(acc.mean()), float(ref.mean())


def _check_scene(name, acc, ref, what="", gates=None):
    f2, f3 = _fractions(acc, ref)
    g2, g3 = (gates or GATES)[name]
    print("parity %s %s: within 1e-2 %.5f (gate %.4f), within 1e-3 %.5f (gate %.4f), mean gpu %.6g oracle %.6g" % (name, what, f2, g2, f3, g3, acc.mean(), ref.mean()))
    assert np.isfinite(acc).all()
    assert f2 >= g2 and f3 >= g3, (name, what, f2, f3)
    return f2, f3


def test_isaac64_raw_outputs_bit_exact(gpu, scenes, orc):
    sc, _ = scenes("cornell_mini")
    gpu.upload_scene(sc)
    # the strips: 4 + nc < 0 on their far side, where the reference's `as usize` saturates to a zero seed word (test_isaac64.py)
    for (w, h, sampling) in [(480, 270, 1), (33, 17, 7), (1920, 1080, 1024), (64, 1, 9), (1, 64, 3), (257, 3, 4000000000)]:
        gpu.set_resolution(w, h)
        n = min(w * h * 4, 256)
        first = (w * h * 4 - n) if sampling == 7 else 0
        got = gpu.debug_draws(sampling, first, n, 64)
        for i in range(0, n, 17):
            p = first + i
            pix, sub = p >> 2, p & 3
            ref = orc.path_draws(w, h, pix % w, pix // w, sub & 1, sub >> 1, sampling, 64)
            assert np.array_equal(got[i], ref), (w, h, sampling, p)


def test_path_draws_match_oracle_after_lens_rejection(gpu, scenes, orc):
    sc, _ = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    w, h, sampling = 64, 36, 3
    gpu.set_resolution(w, h)
    import ctypes as C
    out = np.empty((h, w, 4, 20), dtype=np.float32)
    rc = gpu.L.hr_debug_path_draws(gpu._h, sampling, C.c_void_p(out.ctypes.data))
    assert rc == 0, gpu.L.hr_last_error()
    rng = np.random.default_rng(1)
    for _ in range(200):
        x, y, sub = int(rng.integers(w)), int(rng.integers(h)), int(rng.integers(4))
        raw = orc.path_draws(w, h, x, y, sub & 1, sub >> 1, sampling, 64)
        f = [orc.u64_to_f64(v) for v in raw]
        j = 0
        while not ((2 * f[2 * j] - 1) ** 2 + (2 * f[2 * j + 1] - 1) ** 2 < 1.0):
            j += 1
        exp = [2 * f[2 * j] - 1, 2 * f[2 * j + 1] - 1] + f[2 * j + 2:2 * j + 20]
        assert np.array_equal(out[y, x, sub], np.asarray(exp, dtype=np.float64).astype(np.float32)), (x, y, sub)
    # precise shading: the records' twin holds what the rounding took away (seed_seg_kernel<.., LO>, isaac_core.h draw_lo_f32) — EVERY path of
    # a larger frame, the fix-up kernel's paths (five rejected lens attempts: record_from_window<LO>) among them
    assert gpu.L.hr_debug_path_draw_residuals(gpu._h, sampling, C.c_void_p(out.ctypes.data)) != 0      # fp32 shading in force: no twin
    w, h = 320, 180
    gpu.set_resolution(w, h)
    gpu.set_option("precise_shading", 1)
    try:
        hi = np.empty((h, w, 4, 20), dtype=np.float32)
        lo = np.empty((h, w, 4, 20), dtype=np.float32)
        assert gpu.L.hr_debug_path_draws(gpu._h, sampling, C.c_void_p(hi.ctypes.data)) == 0, gpu.L.hr_last_error()
        assert gpu.L.hr_debug_path_draw_residuals(gpu._h, sampling, C.c_void_p(lo.ctypes.data)) == 0, gpu.L.hr_last_error()
    finally:
        gpu.set_option("precise_shading", -1)
    deep = 0
    for y in range(0, h, 3):
        for x in range(w):
            for sub in range(4):
                raw = orc.path_draws(w, h, x, y, sub & 1, sub >> 1, sampling, 64)
                f = (np.asarray(raw, dtype=np.uint64) & np.uint64((1 << 52) - 1)).astype(np.float64) * 2.0 ** -52      # rand 0.4.3 next_f64: the low 52 bits (orc.u64_to_f64)
                j = 0
                while not ((2 * f[2 * j] - 1) ** 2 + (2 * f[2 * j + 1] - 1) ** 2 < 1.0):
                    j += 1
                deep += j >= 5
                exact = f[2 * j + 2:2 * j + 20]
                assert np.array_equal(hi[y, x, sub, 2:], exact.astype(np.float32)), (x, y, sub)
                assert np.array_equal(lo[y, x, sub, 2:], (exact - hi[y, x, sub, 2:].astype(np.float64)).astype(np.float32)), (x, y, sub, j)
    assert deep >= 3        # (4.6e-4 of 76,800 paths: ~35)


@pytest.mark.parametrize("name", ["rtcamp6_v3_1", "cornell_mini", "spheres"])
def test_closest_hit_matches_oracle(gpu, scenes, name):
    sc, o = scenes(name)
    gpu.upload_scene(sc)
    rng = np.random.default_rng(7)
    n = 4000
    cam = sc.desc.camera
    eye = np.array(cam.eye.tuple())
    org = eye + rng.normal(size=(n, 3)) * 0.3
    tgt = rng.uniform(-2.5, 2.5, size=(n, 3)) * np.array([1.0, 0.6, 1.0]) + np.array([0, 0.8, 0])
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays32 = np.concatenate([org, d], axis=1).astype(np.float32)
    got, gel = gpu.debug_intersect(rays32)
    ref, rel = o.intersect(rays32.astype(np.float64))
    same_hit = got[:, 0] == ref[:, 0]
    assert same_hit.mean() > 0.999
    both = same_hit & (ref[:, 0] == 1)
    assert (gel[both] == rel[both]).mean() > 0.998
    ok = both & (gel == rel)
    terr = np.abs(got[ok, 1] - ref[ok, 1]) / np.maximum(1.0, ref[ok, 1])
    # grazing sphere hits are ill-conditioned (t = -b - sqrt(d) with d -> 0): bound the bulk tightly, the tail loosely
    assert np.quantile(terr, 0.99) < 2e-5 and terr.max() < 1e-3, (np.quantile(terr, 0.99), terr.max())
    nerr = np.abs(got[ok, 5:8] - ref[ok, 5:8]).max(axis=1)
    # normals: a small sphere far from the origin divides an fp32 position error (~|o| * 1e-7) by its radius;
    # grazing hits are ill-conditioned on top of that
    assert np.median(nerr) < (1e-5 if name == "rtcamp6_v3_1" else 1e-4) and np.quantile(nerr, 0.999) < 2e-3


def _query_rays(sc, n, seed):
    rng = np.random.default_rng(seed)
    eye = np.array(sc.desc.camera.eye.tuple())
    org = eye + rng.normal(size=(n, 3)) * 0.3
    tgt = rng.uniform(-2.5, 2.5, size=(n, 3)) * np.array([1.0, 0.6, 1.0]) + np.array([0, 0.8, 0])
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([org, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("name,builder,quant", [("rtcamp6_v3_1", 0, 1), ("rtcamp6_v3_1", 0, 0), ("rtcamp6_v3_1", 2, 1), ("rtcamp6_dodeca", 0, 1),
                                                ("cornell_mini", 0, 1), ("spheres", 0, 1), ("spheres", 1, 1)])
def test_production_traversal_closest_hit_matches_oracle(gpu, scenes, name, builder, quant):
    """hr_debug_trace = the render kernel's own traversal (traverse_wave: 16-byte quantised records, box / leaf phases, two parked
    leaves, closest-hit culling) as a query.  Against the oracle's closest hit (bvh.rs:213-290 + scene.rs:385-401) with the
    tolerances of test_closest_hit_matches_oracle, and BIT-identical to the scalar walk of hr_debug_intersect: the order in which
    a wave visits nodes and tests leaves must not change what a ray hits."""
    sc, o = scenes(name)
    gpu.set_option("bvh_builder", builder)
    gpu.set_option("quant_nodes", quant)
    try:
        gpu.upload_scene(sc)
        n = 6000 + 37   # not a multiple of 64: the last wave is ragged
        rays32 = _query_rays(sc, n, 11)
        got, gel = gpu.debug_trace(rays32)
        scalar, sel = gpu.debug_intersect(rays32)
    finally:
        gpu.set_option("bvh_builder", -1)
        gpu.set_option("quant_nodes", 1)
    assert np.array_equal(gel, sel)
    assert np.array_equal(got.view(np.uint32), scalar.view(np.uint32))
    ref, rel = o.intersect(rays32.astype(np.float64))
    same_hit = got[:, 0] == ref[:, 0]
    assert same_hit.mean() > 0.999
    both = same_hit & (ref[:, 0] == 1)
    assert (gel[both] == rel[both]).mean() > 0.998
    ok = both & (gel == rel)
    terr = np.abs(got[ok, 1] - ref[ok, 1]) / np.maximum(1.0, ref[ok, 1])
    assert np.quantile(terr, 0.99) < 2e-5 and terr.max() < 1e-3, (np.quantile(terr, 0.99), terr.max())


@pytest.mark.parametrize("builder", [0, 1, 2])
def test_triangle_test_boundary_rules_on_the_gpu(gpu, ha, orc, builder):
    """The GPU twin of tests/test_emu_parity.py::test_triangle_test_boundary_rules: edges, vertices, t == 0, det == 0 and a degenerate
    triangle through the render kernel's traversal (hr_debug_trace) and the scalar walk (hr_debug_intersect), for every builder — the
    same decisions and the same distances, to the bit, as the reference's Cramer's rule (oracle, f64) on geometry whose numbers are
    exact in both (bvh.rs:266-290)."""
    import ctypes as C
    from test_emu_parity import TRI_EDGE_RAYS, _triangle_edge_geometry, _triangle_scene
    verts, faces = _triangle_edge_geometry()
    d, keep = _triangle_scene(ha, verts, faces)

    class Holder:
        pass
    h = Holder()
    h.desc_ptr = C.pointer(d)
    h.keep = keep
    o = orc.OracleScene(C.addressof(d))
    ref, rel = o.intersect(TRI_EDGE_RAYS.astype(np.float64))
    gpu.set_option("bvh_builder", builder)
    try:
        gpu.upload_scene(h)
        got, gel = gpu.debug_trace(TRI_EDGE_RAYS)
        scalar, sel = gpu.debug_intersect(TRI_EDGE_RAYS)
    finally:
        gpu.set_option("bvh_builder", -1)
    assert np.array_equal(got.view(np.uint32), scalar.view(np.uint32)) and np.array_equal(gel, sel)
    assert np.array_equal(got[:, 0], ref[:, 0].astype(np.float32)), (got[:, 0], ref[:, 0])
    assert np.array_equal(got[:, 0], np.array([1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 0, 0, 0], dtype=np.float32))
    hit = ref[:, 0] == 1
    assert np.array_equal(got[hit, 1], ref[hit, 1].astype(np.float32)), (got[hit, 1], ref[hit, 1])


@pytest.mark.parametrize("builder", [0, 2])
def test_triangle_test_across_scales_on_the_gpu(gpu, ha, orc, builder):
    """The GPU twin of tests/test_emu_parity.py::test_triangle_test_across_scales: triangles of edge length 1e-3 .. 1e3 up to 1e3 from
    the origin through the render kernel's traversal — the oracle's hits, distances to a few fp32 roundings of the quantities involved."""
    import ctypes as C
    from test_emu_parity import _hit_scale, _scaled_triangle_soup, _triangle_scene
    verts, faces, rays, expect = _scaled_triangle_soup()
    d, keep = _triangle_scene(ha, verts, faces)

    class Holder:
        pass
    h = Holder()
    h.desc_ptr = C.pointer(d)
    h.keep = keep
    o = orc.OracleScene(C.addressof(d))
    ref, rel = o.intersect(rays.astype(np.float64))
    gpu.set_option("bvh_builder", builder)
    try:
        gpu.upload_scene(h)
        got, gel = gpu.debug_trace(rays)
        scalar, sel = gpu.debug_intersect(rays)
    finally:
        gpu.set_option("bvh_builder", -1)
    assert np.array_equal(got.view(np.uint32), scalar.view(np.uint32)) and np.array_equal(gel, sel)
    assert np.array_equal(got[:, 0], ref[:, 0].astype(np.float32))
    hit = ref[:, 0] == 1
    assert hit[expect].mean() > 0.95 and np.array_equal(gel[hit], rel[hit])
    terr = np.abs(got[hit, 1] - ref[hit, 1]) / _hit_scale(verts, faces, rays[hit].astype(np.float64), ref[hit, 1])
    assert terr.max() < 2e-6 and np.quantile(terr, 0.9) < 1e-7, (terr.max(), np.quantile(terr, 0.9))


@pytest.mark.parametrize("name", ["rtcamp6_v3_1", "rtcamp6_v2", "tbf3"])
def test_production_traversal_shadow_rays_match_oracle(gpu, scenes, name):
    """Shadow rays through the render kernel's traversal WITH its two exact work savers (search limited to the sample distance
    + 0.03, stop at the first hit more than 0.02 in front of the sample): the visibility verdict must be the one of
    renderer.rs:280 — the reference's unbounded closest hit lies within |dp|^2 < 4e-4 of the light sample (vector.rs:89-91)."""
    sc, o = scenes(name)
    gpu.upload_scene(sc)
    els = [sc.desc.elements[i] for i in range(sc.desc.num_elements)]
    lights = [e for e in els if e.kind == 0 and max(e.material.emission.color.tuple()) > 0]
    assert lights
    rays32 = _query_rays(sc, 5000, 5)
    prim, _ = o.intersect(rays32.astype(np.float64))
    hit = prim[:, 0] == 1
    pos, nrm = prim[hit, 2:5], prim[hit, 5:8]
    rng = np.random.default_rng(3)
    origin = pos + 1e-4 * nrm                                  # material.rs: ray origin offset along the normal
    lt = lights[0]
    u = rng.normal(size=origin.shape)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    sample = np.array(lt.center.tuple()) + (lt.radius + 1e-4) * u   # scene.rs:92-101: a point on the sphere of radius r + OFFSET
    sv = sample - origin
    sl = np.linalg.norm(sv, axis=1)
    srays = np.concatenate([origin, sv / sl[:, None]], axis=1).astype(np.float32)
    # both sides see the fp32 ray; the sample point of that ray at the fp32 distance
    sl32 = sl.astype(np.float32)
    got, _ = gpu.debug_trace(srays, sl32)
    ref, _ = o.intersect(srays.astype(np.float64))
    d64 = srays[:, 3:6].astype(np.float64)
    sample32 = srays[:, 0:3].astype(np.float64) + d64 * sl32[:, None].astype(np.float64)
    ref_visible = (ref[:, 0] == 1) & (((ref[:, 2:5] - sample32) ** 2).sum(axis=1) < 4e-4)
    gpu_visible = (got[:, 0] == 1) & ((got[:, 1].astype(np.float64) - sl32) ** 2 < 4e-4)
    # a hit whose distance to the sample is within fp32 rounding of the 0.02 threshold may land on either side
    margin = np.abs(np.sqrt(((ref[:, 2:5] - sample32) ** 2).sum(axis=1)) - 0.02) < 1e-4
    agree = (ref_visible == gpu_visible) | margin
    assert ref_visible.sum() > 100 and (~ref_visible).sum() > 100      # the sample exercises both verdicts
    assert agree.mean() > 0.9995, (agree.mean(), int((~agree).sum()))
    vis = ref_visible & gpu_visible
    terr = np.abs(got[vis, 1] - ref[vis, 1]) / np.maximum(1.0, ref[vis, 1])
    # the light is a sphere and most of its samples are seen at a grazing angle (t = -b - sqrt(d), d -> 0): bulk tight, tail loose
    assert np.quantile(terr, 0.99) < 2e-5 and terr.max() < 1e-3, (np.quantile(terr, 0.99), terr.max())


@pytest.mark.parametrize("name,w,h,s", [("rtcamp6_v3_1", 320, 180, 4), ("cornell_mini", 96, 64, 4), ("cornell_mini", 320, 200, 8), ("spheres", 256, 144, 2),
                                         ("rtcamp6_dodeca", 195, 111, 2), ("rtcamp6_v3", 256, 144, 3), ("simple", 256, 144, 3),
                                         ("material_examples", 256, 144, 3), ("rtcamp6_v1", 256, 144, 2), ("rtcamp6_v2", 192, 108, 1),
                                         ("rtcamp5", 256, 144, 2), ("tbf3", 256, 144, 2)])
@pytest.mark.parametrize("precise", [0, 1])
def test_radiance_accumulator_matches_oracle(gpu, scenes, name, w, h, s, precise):
    sc, o = scenes(name)
    gpu.upload_scene(sc)
    gpu.set_resolution(w, h)
    gpu.set_option("batch", 3)
    gpu.set_option("precise_shading", precise)      # (0 = fp32 shading pinned: the automatic choice turns precise shading on for scenes without meshes)
    try:
        gpu.render(1, s + 1)
        acc = gpu.read_accumulator()
    finally:
        gpu.set_option("batch", 0)   # back to automatic
        gpu.set_option("precise_shading", -1)
    ref, _ = o.render(w, h, 1, s + 1, threads=0)
    _check_scene(name, acc, ref, "%dx%dx%d%s" % (w, h, s, " precise" if precise else ""), GATES_PRECISE if precise else GATES)
    m_gpu, m_ref = float(acc.mean()), float(ref.mean())
    assert abs(m_gpu - m_ref) <= 2e-3 * max(1.0, abs(m_ref)), (m_gpu, m_ref)


# Per-path accounting (round 4; tests/path_parity.py; measured values: profiles/r04_parity_report.json "per_path").  Per scene:
#   divergent_ppm        most paths per million that may take another branch than the oracle's (event log or hit element / triangle differs)
#   over_ppm             most paths per million that took the oracle's branches and still differ by more than 1e-3 x max(1, |oracle|)
#   flat_over_ppm        the same among the paths that never bounced off a sphere
# The headline scene and the Cornell box have no same-branch path beyond 1e-3 at all (same_max); in the scenes full of r = 0.1 spheres the
# same-branch tail is chain amplification (every sphere bounce multiplies the ray's fp32 position error by ~2 t / r), visible in the
# report as "over_1e-3_by_sphere_bounces_ppm": nothing at zero or one bounce in the sphere-only scene.
# Round 5: "same branch" now includes every discrete decision of a main ray (pt_core.h PathLog: the face of a cuboid hit and the cube-map face of
# the sky lookup are in the hash — a sky-seam flip, filed as "same branch, off by 0.585" at 7680x4320 in round 4, is a DIVERGENT path now), and
# every scene has a bound on its worst same-branch path: same_max = 2.5 x the worst path of samplings 1 .. 8 - 12 at this size
# (tools/same_branch_tail.py, profiles/r05_same_branch_tail.txt: rtcamp6_v2 0.097, rtcamp5 0.077, spheres 0.031, tbf3 0.015, rtcamp6_v1 0.0009).
# What such a path is: a chain of refractions / r = 0.1 mirror spheres hands the fp32 ray's rounding on, amplified, and the path ends with a sky
# or texture lookup a few texels from the oracle's — on the same face and surface, hence continuous: the error is bounded by the local contrast
# of the environment map (<= ~0.1 of a radiance of 1), where a divergent path is bounded by nothing.
PATH_LIMITS = {
    #                 w,   h,  divergent_ppm, over_ppm, flat_over_ppm, same_max
    "rtcamp6_v3_1": (320, 180, 60.0, 15.0, 10.0, 1e-3),
    "cornell_mini": (160, 100, 60.0, 0.0, 0.0, 1e-3),
    "spheres": (256, 144, 60.0, 300.0, 0.0, 0.08),
    "rtcamp6_v2": (192, 108, 900.0, 2400.0, 300.0, 0.25),
    "rtcamp5": (192, 108, 600.0, 800.0, 300.0, 0.2),
    "tbf3": (192, 108, 200.0, 600.0, 300.0, 0.04),
    "rtcamp6_v1": (192, 108, 150.0, 10.0, 10.0, 2.5e-3),
}


# Option precise_shading (round 6: the split pipeline's shading kernel computes a bounce's geometry in f64 and carries the ray as fp32 +
# residual; csrc/wf_core.h wf_surface_f64).  What it is for: the same-branch tail of the refraction-chain and small-sphere scenes.  Limits =
# the targets the round-5 verdict set (rtcamp6_v2 <= 300, rtcamp5 <= 100, tbf3 <= 80, spheres <= 30 ppm beyond 1e-3; measured at 480x270,
# samplings 1 and 2, profiles/r06_precise_parity.txt: 3.9 - 7.7, 25 - 33, 17 - 29, 29 - 31), with room for the one or two paths that a
# test-sized image turns into 7 - 12 ppm each; same_max = 2 x the worst path measured there.  What is left is the fp32 rounding of the
# DRAWS (the hand-off record holds them rounded once: a diffuse bounce's direction is off by ~4e-7 whatever the arithmetic after it).
# Precise shading, samplings 1 .. 8 at these sizes (tools/ab/precise_limits.sh, profiles/r06_precise_limits.txt): no scene has more than ONE path
# beyond 1e-3 on the oracle's branches in any sampling (one path = 4 - 16 ppm at these sizes); worst same-branch path 3.6e-5 (rtcamp6_v1) ...
# 4.4e-3 (rtcamp5: a roughness texel's border).  over_ppm = three paths, same_max = 2.5 x the worst of the eight samplings.
PATH_LIMITS_PRECISE = {
    #                 w,   h,  divergent_ppm, over_ppm, flat_over_ppm, same_max
    "rtcamp6_v3_1": (320, 180, 40.0, 13.1, 13.1, 1e-3),
    "cornell_mini": (160, 100, 50.0, 47.0, 47.0, 9e-3),
    "spheres": (256, 144, 30.0, 20.5, 0.0, 8.5e-3),
    "rtcamp6_v2": (192, 108, 400.0, 36.5, 36.5, 1e-3),
    "rtcamp5": (192, 108, 75.0, 36.5, 36.5, 0.011),
    "tbf3": (192, 108, 40.0, 36.5, 36.5, 8e-3),
    "rtcamp6_v1": (192, 108, 40.0, 12.5, 12.5, 1e-4),
}


@pytest.mark.parametrize("name", sorted(PATH_LIMITS_PRECISE))
def test_per_path_parity_accounting_precise_shading(gpu, scenes, name):
    """test_per_path_parity_accounting with option precise_shading: the path log comes from the split pipeline's LOG instantiation (the
    event log rides in the path's state), its radiances are what hr_render accumulates in that mode, and the same-branch tail meets the
    tighter limits above."""
    gpu.set_option("precise_shading", 1)
    try:
        _per_path_accounting(gpu, scenes, name, PATH_LIMITS_PRECISE, "precise")
    finally:
        gpu.set_option("precise_shading", -1)


@pytest.mark.parametrize("name", sorted(PATH_LIMITS))
def test_per_path_parity_accounting(gpu, scenes, name):
    """fp32 shading (pinned: option precise_shading 0; the automatic choice would shade `spheres`, a scene without meshes, in f64)."""
    gpu.set_option("precise_shading", 0)
    try:
        _per_path_accounting(gpu, scenes, name, PATH_LIMITS, "fp32 shading")
    finally:
        gpu.set_option("precise_shading", -1)


def _per_path_accounting(gpu, scenes, name, limits, label):
    """Path by path instead of pixel by pixel: hr_debug_path_log (the render kernel's LOG instantiation — same traversal, same path state
    machine) against the oracle's path log.  (i) the logged radiances ARE what hr_render accumulates (bit for bit, in the accumulate
    kernel's order); (ii) a path that took the oracle's branches traced the same number of rays; (iii) divergent paths and same-branch
    outliers stay below the stated parts per million, each divergent path classified by the first event that differs; (iv) the fraction
    of accumulator channels within 1e-3 that the per-path figures predict for a 4-sampling render is met by a real one."""
    import path_parity
    w, h, div_ppm, over_ppm, flat_ppm, same_max = limits[name]
    sc, o = scenes(name)
    gpu.upload_scene(sc)
    gpu.set_resolution(w, h)
    g = gpu.debug_path_log(1)
    # (i) one sampling rendered the normal way: 0 + ((s0 + s1) + (s2 + s3)) per pixel and channel, in fp32
    gpu.clear()
    gpu.render(1, 2)
    acc = gpu.read_accumulator()
    rad = g[0]
    want = (rad[:, :, 0] + rad[:, :, 1]) + (rad[:, :, 2] + rad[:, :, 3])
    assert np.array_equal(acc, want.astype(np.float32)), "the path log's radiances are not what hr_render accumulates"
    a = path_parity.account(g, o.path_log(w, h, 1))
    sb = a["same_branch"]
    print("per-path %s [%s]: %d paths, divergent %.1f ppm %s; same-branch over 1e-3: %.1f ppm (by sphere bounces %s, no sphere %.1f ppm), max %.3g, p99.9 %.3g" % (
        name, label, a["paths"], a["divergent_ppm"], a["divergent_by_class_ppm"], sb["over_1e-3_floor1_ppm"], sb["over_1e-3_by_sphere_bounces_ppm"],
        sb["no_sphere_bounce"]["over_1e-3_floor1_ppm"], sb["max_rel_floor1"], sb["p999_rel_floor1"]))
    assert sb["rays_equal"]                                                    # (ii)
    assert a["divergent_ppm"] <= div_ppm, a                                    # (iii)
    assert sb["over_1e-3_floor1_ppm"] <= over_ppm and sb["no_sphere_bounce"]["over_1e-3_floor1_ppm"] <= flat_ppm, sb
    assert sb["max_rel_floor1"] <= same_max, sb
    oq = sb["other_texel_quad"]
    print("per-path %s: %.1f ppm of the same-branch paths interpolated a texture between other texels than the oracle (worst of them off by %.3g)" % (name, oq["ppm"], oq["max_rel_floor1"]))
    assert abs(a["mean_radiance"]["gpu"] - a["mean_radiance"]["oracle"]) <= 2e-3 * a["mean_radiance"]["oracle"]
    # (iv) the pixel gates derived instead of measured: a channel of an S-sampling accumulator can only be off by more than 1e-3 if one of
    # the pixel's 4 S paths is (divergent or a same-branch outlier); with p = that probability per path, at least (1 - p)^(4 S) of the
    # channels are clean.  Checked against a real 4-sampling render with a factor 2 on p for the sampling-to-sampling scatter.
    S = 4
    p_bad = 1e-6 * (a["divergent_radiance"]["over_1e-3_floor1_ppm"] + sb["over_1e-3_floor1_ppm"])
    predicted = (1.0 - min(1.0, 2.0 * p_bad + 2e-5)) ** (4 * S)
    gpu.clear()
    gpu.render(1, S + 1)
    acc4 = gpu.read_accumulator()
    ref4, _ = o.render(w, h, 1, S + 1, threads=0)
    f2, f3 = _fractions(acc4, ref4)
    print("per-path %s: predicted fraction of channels within 1e-3 at %d samplings >= %.5f, measured %.5f (GATES: %.4f)" % (name, S, predicted, f3, GATES[name][1]))
    assert f3 >= predicted, (name, f3, predicted)


def test_per_path_parity_at_full_size(gpu, scenes):
    """The same accounting on EVERY path of one sampling of the headline configuration (1920x1080: 8,294,400 paths; the report in
    profiles/ holds samplings 1 and 1000, this test takes another one): tens of ppm divergent, a handful of same-branch paths beyond
    1e-3 in eight million, equal ray counts on every same-branch path."""
    import path_parity
    sc, o = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    gpu.set_resolution(1920, 1080)
    a = path_parity.account(gpu.debug_path_log(517), o.path_log(1920, 1080, 517))
    sb = a["same_branch"]
    print("per-path full size: %d paths, divergent %.2f ppm %s; same-branch over 1e-3: %.2f ppm, max %.3g, p99.9 %.3g" % (
        a["paths"], a["divergent_ppm"], a["divergent_by_class_ppm"], sb["over_1e-3_floor1_ppm"], sb["max_rel_floor1"], sb["p999_rel_floor1"]))
    assert a["paths"] == 1920 * 1080 * 4 and sb["rays_equal"]
    assert a["divergent_ppm"] <= 40.0 and sb["over_1e-3_floor1_ppm"] <= 3.0 and sb["max_rel_floor1"] <= 0.05 and sb["p999_rel_floor1"] <= 5e-5, a
    assert abs(a["mean_radiance"]["gpu"] - a["mean_radiance"]["oracle"]) <= 2e-4 * a["mean_radiance"]["oracle"]



@pytest.mark.parametrize("seed,builder,precise", [(1, 0, 0), (2, 0, 0), (3, 0, 0), (4, 0, 0), (5, 2, 0), (6, 1, 0), (7, 0, 0), (8, 2, 0), (1, 0, 1), (3, 0, 1), (5, 2, 1), (6, 1, 1)])
def test_random_scenes_path_by_path(gpu, ha, orc, seed, builder, precise):
    """Fuzz tier (tests/random_scenes.py): every element kind x surface type x textured / constant albedo, emission and roughness,
    overlapping and nested primitives, three NEE emitters of any surface type, an emissive cuboid, square and round lenses — combinations
    the reference's eight scenes do not contain — on the host-built and the device-built trees.  Path by path against the oracle: the logged
    radiances are what hr_render accumulates, same-branch paths trace the same number of rays, few divergent paths and same-branch
    outliers, no systematic difference; and the accumulator of two samplings within the usual tolerance."""
    import path_parity
    import random_scenes
    sc = random_scenes.build(ha, seed)
    o = orc.OracleScene(sc.desc_ptr)
    w, h = 160, 90
    try:
        gpu.set_option("bvh_builder", builder)
        gpu.upload_scene(sc)
    finally:
        gpu.set_option("bvh_builder", -1)
    gpu.set_resolution(w, h)
    gpu.set_option("precise_shading", precise)
    try:
        g = gpu.debug_path_log(1)
        gpu.clear()
        gpu.render(1, 2)
        acc1 = gpu.read_accumulator().copy()
        gpu.clear()
        gpu.render(1, 3)
        acc = gpu.read_accumulator().copy()
    finally:
        gpu.set_option("precise_shading", -1)
    rad = g[0]
    assert np.array_equal(acc1, ((rad[:, :, 0] + rad[:, :, 1]) + (rad[:, :, 2] + rad[:, :, 3])).astype(np.float32))
    a = path_parity.account(g, o.path_log(w, h, 1))
    sb = a["same_branch"]
    print("random scene %d (builder %d, precise %d): divergent %.0f ppm %s, same-branch over 1e-3 %.0f ppm, max %.3g, mean %.6g / %.6g" % (
        seed, builder, precise, a["divergent_ppm"], a["divergent_by_class_ppm"], sb["over_1e-3_floor1_ppm"], sb["max_rel_floor1"], a["mean_radiance"]["gpu"], a["mean_radiance"]["oracle"]))
    # gates at ~3 x what the campaigns measure (profiles/r05_fuzz_campaign_1000.txt: divergent <= 291 ppm, same-branch beyond 1e-3 <= 35 ppm;
    # until round 5 both gates stood at 2,000)
    assert sb["rays_equal"] and a["divergent_ppm"] <= 900.0 and sb["over_1e-3_floor1_ppm"] <= 120.0, a
    assert abs(a["mean_radiance"]["gpu"] - a["mean_radiance"]["oracle"]) <= 3e-3 * a["mean_radiance"]["oracle"], a["mean_radiance"]
    ref, _ = o.render(w, h, 1, 3, threads=0)
    f2, f3 = _fractions(acc, ref)
    assert np.isfinite(acc).all() and f2 >= 0.998 and f3 >= 0.99, (seed, f2, f3)


def test_golden_accumulator(gpu, scenes):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rtcamp6_64x36_s2.npz"))
    sc, _ = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    gpu.set_resolution(64, 36)
    gpu.render(1, 3)
    frac, m_gpu, m_ref = _compare(gpu.read_accumulator(), g["acc"].astype(np.float64))
    assert frac >= 0.999 and abs(m_gpu - m_ref) <= 2e-3 * max(1.0, m_ref)   # 6,912 channels: one flipped path is 4e-4


def test_sharding_and_batching_are_exact_partitions(gpu, scenes):
    """Size-independent properties: samplings are independent units, so any partition of the sampling
    range (batch size, stride sharding as used across GPUs) must give the same sum up to fp32 order."""
    sc, _ = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    gpu.set_resolution(120, 68)
    gpu.set_option("batch", 8)
    gpu.render(1, 9)
    whole = gpu.read_accumulator().astype(np.float64)
    gpu.clear()
    gpu.set_option("batch", 1)
    for rank in range(4):
        gpu.render(1 + rank, 9, 4)
    parts = gpu.read_accumulator().astype(np.float64)
    assert np.abs(whole - parts).max() <= 1e-4 * max(1.0, np.abs(whole).max())
    # determinism of a repeat
    gpu.clear()
    gpu.set_option("batch", 8)
    gpu.render(1, 9)
    again = gpu.read_accumulator().astype(np.float64)
    assert np.abs(whole - again).max() <= 1e-5 * max(1.0, np.abs(whole).max())
    gpu.set_option("batch", 0)   # back to automatic


def _crop_parity(gpu, o, name, W, H, S, crops, size=64, gates=None, crop_slack=None):
    """GPU render of the whole W x H image, S samplings, against the oracle on `crops` (top-left corners, size x size pixels)
    at the SAME full-image coordinates (seeds depend on them): every crop must pass the scene's gates, and the union of the
    crops agrees in the mean to 5e-3."""
    gpu.set_resolution(W, H)
    gpu.set_option("batch", 0)
    gpu.clear()
    gpu.render(1, S + 1)
    acc = gpu.read_accumulator()
    assert np.isfinite(acc).all() and (acc >= 0).all()
    got_all, ref_all, fr = [], [], []
    for (x0, y0) in crops:
        ref = o.render_region(W, H, x0, y0, size, size, 1, S + 1, threads=0)
        got = acc[y0:y0 + size, x0:x0 + size].astype(np.float64)
        f2, f3 = _fractions(got, ref)
        print("parity %s %dx%dx%d crop (%d,%d): within 1e-2 %.5f, within 1e-3 %.5f, mean gpu %.6g oracle %.6g" % (name, W, H, S, x0, y0, f2, f3, got.mean(), ref.mean()))
        fr.append((f2, f3))
        got_all.append(got)
        ref_all.append(ref)
    g2, g3 = gates or GATES[name]
    slack = crop_slack or CROP_SLACK
    # a single crop holds 12,288 channels (one decorrelated path is 2.4e-4 of them) and may sit on the scene's hardest spot
    # (refracting dodecahedron, grazing sphere rims): per crop the gates are relaxed by CROP_SLACK, pooled they hold as they are
    for (x0, y0), (f2, f3) in zip(crops, fr):
        assert f2 >= g2 - slack[0] and f3 >= g3 - slack[1], (name, x0, y0, f2, f3)
    p2, p3 = _fractions(np.concatenate(got_all), np.concatenate(ref_all))
    print("parity %s %dx%dx%d all crops: within 1e-2 %.5f (gate %.4f), within 1e-3 %.5f (gate %.4f)" % (name, W, H, S, p2, g2, p3, g3))
    assert p2 >= g2 and p3 >= g3, (name, p2, p3)
    g, r = np.concatenate(got_all).mean(), np.concatenate(ref_all).mean()
    assert abs(g - r) <= 5e-3 * r, (g, r)
    return acc


def test_config3_full_size_crops(gpu, scenes):
    """BASELINE config 3 (rtcamp6, 1920x1080) at its full size: exact path count, no RNG-window overflow, and oracle parity on
    five 64x64 crops at full-image coordinates: sky, textured floor + armadillo, bunny wire silhouette, picture-frame edge
    against the sky (mirror), and the image's bottom-right corner."""
    sc, o = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    gpu.set_option("counters", 1)
    acc = _crop_parity(gpu, o, "rtcamp6_v3_1", 1920, 1080, 8, [(160, 130), (800, 760), (700, 250), (1500, 300), (1856, 1016)])
    st = gpu.stats()
    gpu.set_option("counters", 0)
    assert st["paths"] == 1920 * 1080 * 4 * 8 and st["rng_overflow"] == 0
    assert (acc.sum(axis=2) > 0).mean() > 0.9   # black floor texels (albedo 0) and dark sky stay exactly 0
    assert 2.5 < (st["rays"] + st["shadow_culled"]) / st["paths"] < 3.6          # 3.05 rays per path (scene.intersect calls of the reference: traced + culled)
    assert 2.0 < st["rays"] / st["paths"] < 2.9          # what the kernel traces of them


def test_post_chain_matches_oracle(gpu, scenes, orc):
    sc, o = scenes("cornell_mini")
    gpu.upload_scene(sc)
    for (w, h) in [(96, 64), (8, 5), (1, 1), (3, 1)]:
        ref, _ = o.render(w, h, 1, 3, threads=0)
        gpu.set_resolution(w, h)
        gpu.write_accumulator(ref.astype(np.float32))
        img = gpu.resolve(2)
        exp = orc.resolve(ref.astype(np.float32).astype(np.float64), 2)
        diff = np.abs(img.astype(int) - exp.astype(int))
        assert diff.max() <= 1, (w, h, diff.max())
        assert (diff == 0).mean() > 0.99 or w * h < 50


def test_error_paths(ha):
    r = ha.Renderer(0)
    with pytest.raises(ha.HipError) as e:
        r.render(1, 2)
    assert e.value.code == -3          # no scene
    sc = ha.Scene("cornell_mini")
    r.upload_scene(sc)
    with pytest.raises(ha.HipError) as e:
        r.render(1, 2)
    assert e.value.code == -4          # no target
    raw = np.zeros((4, 4, 4, 8), dtype=np.uint32)
    assert r.L.hr_debug_path_log(r._h, 1, raw.ctypes.data) == -4 and r.L.hr_debug_path_log(r._h, 1, None) == -1
    r.set_resolution(4, 4)
    r.render(5, 5)                     # empty range is a no-op
    assert not r.read_accumulator().any()
    with pytest.raises(ha.HipError):
        r.set_option("nonsense", 1)
    for key, bad in [("batch", 65), ("max_leaf", 0), ("bvh_builder", 3), ("rng_window", 32), ("russian_roulette", 1), ("trace_boost", 5)]:
        with pytest.raises(ha.HipError):
            r.set_option(key, bad)
    for key, bad in [("adv_den", 0), ("leaf_den", 100), ("min_waves", 9), ("seed_mode", 5), ("seed_split", 10), ("nonsense", 1)]:
        with pytest.raises(ha.HipError):
            r.set_debug_option(key, bad)
    # the measurement knobs are not reachable through the product call: a host cannot ship a garbage image by key string
    for key in ("debug_skip", "seed_prof", "seed_mode", "seed_split", "seed_prio", "init_prio", "adv_den", "kchunk", "trace_wgs", "node_unroll"):
        with pytest.raises(ha.HipError):
            r.set_option(key, 1)
    with pytest.raises(ha.HipError):
        r.render_debug(7)
    with pytest.raises(ha.HipError):
        r.debug_draws(1, 0, 4, 65)       # window larger than the stored tail
    # non-finite geometry is refused at upload (tests/test_emu_parity.py has the whole list); the scene uploaded before stays usable
    import ctypes as C
    import random_scenes
    bad = random_scenes.build(ha, 5, spheres=3, cuboids=2, meshes=1)
    d = C.cast(bad.desc_ptr, C.POINTER(ha.SceneDesc)).contents
    next(d.elements[i] for i in range(d.num_elements) if d.elements[i].kind == ha.SPHERE).center.x = float("nan")
    with pytest.raises(ha.HipError) as e:
        r.upload_scene(bad)
    assert e.value.code == -1 and "not finite" in str(e.value)
    r.clear()
    r.render(1, 2)
    assert np.isfinite(r.read_accumulator()).all() and r.read_accumulator().any()
    r.close()


def test_bind_accumulator_checks_the_callers_pointer(ha, scenes):
    """hr_bind_accumulator takes a caller's device pointer (e.g. a torch tensor the host all-reduces itself): what can be checked of it is
    checked before plain stores go there — device memory, of the context's device, float-aligned, W x H x 3 floats inside its allocation.
    A refused binding leaves the previous one in place; a good one renders what the internal accumulator renders."""
    import torch
    sc, _ = scenes("cornell_mini")
    r = ha.Renderer(0)
    try:
        r.upload_scene(sc)
        r.set_resolution(64, 40)
        r.render(1, 3)
        own = r.read_accumulator().copy()
        host = np.zeros((40, 64, 3), dtype=np.float32)
        small = torch.zeros((8,), dtype=torch.float32, device="cuda:0")
        good = torch.zeros((40, 64, 3), dtype=torch.float32, device="cuda:0")
        for what, ptr, msg in [("a host pointer", host.ctypes.data, "not device memory"), ("a wild pointer", 4096, "not device memory"),
                               ("a misaligned pointer", good.data_ptr() + 2, "aligned")]:
            with pytest.raises(ha.HipError) as e:
                r.bind_accumulator(ptr)
            assert e.value.code == -1 and msg in str(e.value), (what, str(e.value))
        assert np.array_equal(r.read_accumulator(), own)         # still the internal accumulator
        r.bind_accumulator(good.data_ptr())
        r.clear()
        r.render(1, 3)
        r.synchronize()
        assert np.array_equal(good.cpu().numpy(), own)
        r.bind_accumulator(0)
        # a tensor of the wrong shape: 32 bytes somewhere in one of torch's 2-MiB small-block segments against 1024 x 1024 x 3 floats
        r.set_resolution(1024, 1024)
        with pytest.raises(ha.HipError) as e:
            r.bind_accumulator(small.data_ptr())
        assert e.value.code == -1 and "too small" in str(e.value), str(e.value)
    finally:
        r.close()


def test_no_device_memory_growth_over_context_cycles(ha):
    """create -> upload (host and device builders in turn) -> render -> replace the scene in place -> render -> destroy, twenty times: the
    device's free memory after every cycle is what it was after the first (the HIP runtime keeps its own pools; nothing of ours grows)."""
    import torch
    a, b = ha.Scene("rtcamp6_v3_1"), ha.Scene("cornell_mini")

    def free():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    marks = []
    for i in range(20):
        r = ha.Renderer(0)
        r.set_option("bvh_builder", i % 3)
        r.upload_scene(a if i % 2 else b)
        r.set_resolution(640 + i, 360)
        r.render(1, 9)
        r.resolve(8)
        r.upload_scene(b if i % 2 else a)
        r.set_resolution(320, 200 + i)
        r.render(1, 3)
        r.read_accumulator()
        r.close()
        marks.append(free())
    assert max(marks[1:]) - min(marks[1:]) <= 64 << 20, [m >> 20 for m in marks]


def test_two_contexts_driven_from_two_host_threads(ha):
    """A context is not thread-safe, but two contexts may be driven by two host threads at once (the one-process multi-GPU host does
    exactly that when it wants to): the accumulators are bit for bit what each context renders alone, and each thread reads its own
    hr_last_error."""
    import threading
    sa, sb = ha.Scene("rtcamp6_v3_1"), ha.Scene("cornell_mini")

    def work(scene, w, h, n, out, key):
        r = ha.Renderer(0)
        try:
            r.upload_scene(scene)
            r.set_resolution(w, h)
            for i in range(n):
                r.render(1 + 4 * i, 5 + 4 * i)
            out[key] = r.read_accumulator()
            try:
                r.set_option("nonsense-" + key, 1)
            except ha.HipError as e:
                out[key + "-err"] = str(e)
        finally:
            r.close()

    alone, both = {}, {}
    work(sa, 320, 180, 6, alone, "a")
    work(sb, 200, 120, 9, alone, "b")
    ta = threading.Thread(target=work, args=(sa, 320, 180, 6, both, "a"))
    tb = threading.Thread(target=work, args=(sb, 200, 120, 9, both, "b"))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert np.array_equal(alone["a"], both["a"]) and np.array_equal(alone["b"], both["b"]) and alone["a"].any() and alone["b"].any()
    assert "nonsense-a" in both["a-err"] and "nonsense-b" in both["b-err"]


def test_bound_accumulator_is_exclusive(scenes, ha):
    """hr_bind_accumulator: a caller-owned accumulator belongs to ONE context (the launch's radiance is added with plain loads and stores);
    a second context that tries to bind the same buffer is refused, and the buffer is free again once the first lets go of it."""
    import torch
    sc, _ = scenes("cornell_mini")
    a, b = ha.Renderer(0), ha.Renderer(0)
    try:
        for r in (a, b):
            r.upload_scene(sc)
            r.set_resolution(48, 27)
        buf = torch.zeros((27, 48, 3), dtype=torch.float32, device="cuda:0")
        a.bind_accumulator(buf.data_ptr())
        a.bind_accumulator(buf.data_ptr())                     # binding it again to the same context is fine
        with pytest.raises(ha.HipError) as e:
            b.bind_accumulator(buf.data_ptr())
        assert e.value.code == -1 and "already bound" in str(e.value)
        a.render(1, 3)
        a.synchronize()
        assert float(buf.sum()) > 0 and np.array_equal(a.read_accumulator(), buf.cpu().numpy())
        a.bind_accumulator(None)                               # back to the internal buffer: the caller's is free
        b.bind_accumulator(buf.data_ptr())
        a.set_resolution(48, 27)
        b.set_resolution(48, 27)                               # unbinds too
        a.bind_accumulator(buf.data_ptr())
    finally:
        a.close()
        b.close()


def test_cli_drop_in(tmp_path, scenes, orc):
    """The host driver with the reference's flag surface (main.rs:1230-1256): -w -h -s -t -i; writes result.png,
    NNN.png and result.txt with the reference's log lines; the image matches the oracle's post chain."""
    import os
    import subprocess
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "hanamaru-renderer_amd", "hanamaru-hip")
    assert os.path.exists(exe), "hanamaru-hip not built (run __graft_entry__.build())"
    r = subprocess.run([exe, "-w", "96", "-h", "54", "-s", "6", "-t", "1000", "-i", "1000", "--assets", os.path.join(root, "assets")],
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    for line in ["resolution: 96x54.", "max sampling: 6x4 spp.", "time limit: 1000.00 sec.", "reached max sampling", "output final image: 000.png",
                 "sampled: 6x4 spp."]:
        assert line in r.stdout, (line, r.stdout)
    txt = open(tmp_path / "result.txt").read()
    assert "init scene:" in txt and "sampled: 6x4 spp." in txt and txt.splitlines()[-1].startswith("total ")
    img = np.asarray(Image.open(tmp_path / "result.png"))
    assert img.shape == (54, 96, 3) and np.array_equal(img, np.asarray(Image.open(tmp_path / "000.png")))
    _, o = scenes("rtcamp6_v3_1")
    ref, _ = o.render(96, 54, 1, 7, threads=0)
    exp = orc.resolve(ref, 6)
    d = np.abs(img.astype(int) - exp.astype(int))
    assert (d <= 2).mean() > 0.99 and abs(img.mean() - exp.mean()) < 0.5


def _run_cli(tmp_path, args, timeout=600):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "hanamaru-renderer_amd", "hanamaru-hip")
    assert os.path.exists(exe), "hanamaru-hip not built (run __graft_entry__.build())"
    r = subprocess.run([exe] + [str(a) for a in args] + ["--assets", os.path.join(root, "assets")], cwd=str(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout
    return r.stdout


def _crops_match_oracle(img, o, orc, W, H, samplings, crops, size=16):
    """An image the CLI wrote against orc.resolve of the oracle's accumulator of the same sampling count, on crops at full-image
    coordinates: the post chain is a per-pixel map followed by a 3x3 filter, so the interior of a crop resolved alone equals the image."""
    for (x0, y0) in crops:
        ref = o.render_region(W, H, x0, y0, size, size, 1, samplings + 1, threads=0)
        exp = orc.resolve(ref, samplings)[1:-1, 1:-1].astype(int)
        got = img[y0 + 1:y0 + size - 1, x0 + 1:x0 + size - 1].astype(int)
        d = np.abs(got - exp)
        assert (d <= 2).mean() > 0.99 and (d <= 1).mean() > 0.97 and abs(got.mean() - exp.mean()) < 0.6, (x0, y0, samplings, d.max(), (d <= 1).mean())


def test_cli_time_limit_stops_the_render(tmp_path, scenes, orc):
    """report_progress's time-limit branch (renderer.rs:222-231) in the host driver: a sampling budget that cannot be reached within -t.
    The render stops early ("reached time limit"), the final image takes number 000 and equals result.png, "remain:" is printed, the
    sampling count reported is a whole number of chunks below -s — and the image is the resolve of exactly that many samplings (oracle)."""
    import re
    from PIL import Image
    W, H, S, B = 160, 90, 1000000, 4
    out = _run_cli(tmp_path, ["-w", W, "-h", H, "-s", S, "-t", "0.15", "-i", "1000", "--batch", B])
    assert "reached time limit" in out and "reached max sampling" not in out and "output final image: 000.png" in out and "output progress image" not in out
    remain = float(re.search(r"remain: (-?[0-9.]+) sec\.", out).group(1))
    n = int(re.search(r"sampled: (\d+)x4 spp\.", out).group(1))
    lines = re.findall(r"rendering: (\d+)x4 sampled \(last ([0-9.]+) sec\)\. total: ([0-9.]+) sec \(([0-9.]+) %\)\.", out)
    assert 2 * B <= n < S and n % B == 0 and int(lines[-1][0]) == n and [int(x[0]) for x in lines] == list(range(B, n + 1, B))
    # stopped before the limit, or within the chunks that were already in flight (--inflight, default 8) when it came into sight: a chunk is
    # only issued if it is predicted to finish in time (renderer.rs:222-231 asked for the moment it would finish), and before the first
    # report nothing is known — the first chunk of a cold process carries the code-object load and may alone be longer than this tiny limit
    # (a launch holds several reports — "launches of N reports" — and is timed as a whole: `last` is the launch's time over N)
    per_launch = int(re.search(r"launches of (\d+) reports", out).group(1))
    longest = max(float(x[1]) for x in lines) * per_launch
    assert remain < 0.15 and float(lines[-1][2]) <= 0.15 + 8 * 1.1 * longest + 0.05, (lines[-1], longest)
    img = np.asarray(Image.open(tmp_path / "result.png"))
    assert img.shape == (H, W, 3) and np.array_equal(img, np.asarray(Image.open(tmp_path / "000.png"))) and not (tmp_path / "001.png").exists()
    assert ("sampled: %dx4 spp." % n) in open(tmp_path / "result.txt").read()
    _, o = scenes("rtcamp6_v3_1")
    _crops_match_oracle(img, o, orc, W, H, n, [(40, 30), (100, 60)])


def test_cli_progress_images_at_the_report_interval(tmp_path, scenes, orc):
    """report_progress's interval branch (renderer.rs:243-251): with -i 0 an image is due at every report.  NNN.png with the counter bumped
    only for progress images, the final image takes the next number and equals result.png, and every image holds exactly the samplings
    of the "rendering:" line before it (the chunk in flight is awaited and reported first) — each checked against the oracle's resolve
    of that many samplings.  The default (--batch 1) is the reference's own cadence: one "rendering:" line per sampling."""
    import re
    from PIL import Image
    W, H = 96, 54
    out = _run_cli(tmp_path, ["-w", W, "-h", H, "-s", 24, "-t", "1000", "-i", "0", "--batch", 8])
    events = re.findall(r"rendering: (\d+)x4 sampled|output (progress|final) image: (\d+)\.png|(reached max sampling)|(reached time limit)", out)
    seq = [("r", int(e[0])) if e[0] else (e[1][0], int(e[2])) if e[1] else ("max" if e[3] else "time", 0) for e in events]
    assert seq == [("r", 8), ("p", 0), ("r", 16), ("p", 1), ("r", 24), ("max", 0), ("f", 2)], seq
    _, o = scenes("rtcamp6_v3_1")
    i0, i1, i2 = [np.asarray(Image.open(tmp_path / ("%03d.png" % k))) for k in range(3)]
    assert np.array_equal(i2, np.asarray(Image.open(tmp_path / "result.png"))) and not (tmp_path / "003.png").exists() and not np.array_equal(i0, i1)
    _crops_match_oracle(i0, o, orc, W, H, 8, [(10, 8), (60, 30)])
    _crops_match_oracle(i1, o, orc, W, H, 16, [(10, 8), (60, 30)])
    _crops_match_oracle(i2, o, orc, W, H, 24, [(10, 8), (60, 30)])
    # the reference's cadence is the DEFAULT: one line per sampling, and `-s 5 -i 0` prints and writes exactly what the reference's
    # report_progress does (renderer.rs:205-251): a progress image after every non-final report, the final image is 004.png
    sub = tmp_path / "b1"
    sub.mkdir()
    out = _run_cli(sub, ["-w", W, "-h", H, "-s", 5, "-t", "1000", "-i", "0"])
    events = re.findall(r"rendering: (\d+)x4 sampled \(last|output (progress|final) image: (\d+)\.png|(reached max sampling)", out)
    seq = [("r", int(e[0])) if e[0] else (e[1][0], int(e[2])) if e[1] else ("max", 0) for e in events]
    assert seq == [("r", 1), ("p", 0), ("r", 2), ("p", 1), ("r", 3), ("p", 2), ("r", 4), ("p", 3), ("r", 5), ("max", 0), ("f", 4)], seq
    assert out.index("reached max sampling") < out.index("output final image: 004.png") < out.index("remain: ")
    for k in range(5):
        _crops_match_oracle(np.asarray(Image.open(sub / ("%03d.png" % k))), o, orc, W, H, k + 1, [(30, 20)])
    assert np.array_equal(np.asarray(Image.open(sub / "004.png")), np.asarray(Image.open(sub / "result.png"))) and not (sub / "005.png").exists()
    # an interval that passes now and then while eight launches are in flight: every image is written after the samplings in flight were
    # reported, and holds exactly the samplings of the "rendering:" line before it
    sub = tmp_path / "pipe"
    sub.mkdir()
    # (round 6: a launch holds 64 samplings at this size — the lines still come one per sampling — so the render is made long enough
    # for the interval to pass between launches several times)
    out = _run_cli(sub, ["-w", 320, "-h", 180, "-s", 4000, "-t", "1000", "-i", "0.1"])
    events = re.findall(r"rendering: (\d+)x4 sampled \(last|output (progress|final) image: (\d+)\.png", out)
    seq = [("r", int(e[0])) if e[0] else (e[1][0], int(e[2])) for e in events]
    assert [x[1] for x in seq if x[0] == "r"] == list(range(1, 4001)) and seq[-1][0] == "f" and seq[-2] == ("r", 4000)
    prog = [(i, x[1]) for i, x in enumerate(seq) if x[0] == "p"]
    assert len(prog) >= 1 and [k for _, k in prog] == list(range(len(prog))) and seq[-1][1] == len(prog)
    i, k = prog[len(prog) // 2]
    assert seq[i - 1][0] == "r"
    _crops_match_oracle(np.asarray(Image.open(sub / ("%03d.png" % k))), o, orc, 320, 180, seq[i - 1][1], [(100, 60)])
    # an interval that never passes: no progress image, the final one is 000.png
    sub = tmp_path / "never"
    sub.mkdir()
    out = _run_cli(sub, ["-w", W, "-h", H, "-s", 9, "-t", "1000", "-i", "1000", "--batch", 2])
    assert "output progress image" not in out and "output final image: 000.png" in out and len(re.findall(r"rendering: ", out)) == 5


def test_no_systematic_bias_at_many_samplings(gpu, scenes):
    """32 samplings (128 paths per pixel): per-pixel Monte-Carlo noise is down by 5.7x, so a systematic difference
    between the fp32 kernels (hardware sin/cos/exp/log/rcp) and the f64 oracle would show in block means."""
    sc, o = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    w, h, s = 160, 90, 32
    gpu.set_resolution(w, h)
    gpu.set_option("batch", 8)
    gpu.render(1, s + 1)
    acc = gpu.read_accumulator().astype(np.float64) / (4 * s)
    gpu.set_option("batch", 0)   # back to automatic
    ref, _ = o.render(w, h, 1, s + 1, threads=0)
    ref /= 4 * s
    assert abs(acc.mean() - ref.mean()) <= 5e-4 * ref.mean()
    # 10x10-pixel block means: identical seeds -> identical paths except for rare fp32 branch flips
    ab = acc[:90, :160].reshape(9, 10, 16, 10, 3).mean(axis=(1, 3))
    rb = ref[:90, :160].reshape(9, 10, 16, 10, 3).mean(axis=(1, 3))
    rel = np.abs(ab - rb) / np.maximum(rb, 0.05)
    assert np.quantile(rel, 0.95) < 5e-3 and rel.max() < 5e-2, (np.quantile(rel, 0.95), rel.max())
    frac, _, _ = _compare(acc * 4 * s, ref * 4 * s)
    assert frac >= FRAC_OK


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_debug_renderer_gpu(gpu, scenes, mode):
    sc, o = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    w, h = 96, 54
    gpu.set_resolution(w, h)
    gpu.render_debug(mode)
    got = gpu.read_accumulator().astype(np.float64)
    ref = o.render_debug(w, h, mode)
    d = np.abs(got - ref)
    assert (d <= 2e-3 * np.maximum(1.0, np.abs(ref))).mean() > 0.99
    assert abs(got.mean() - ref.mean()) < 2e-3 * max(1.0, abs(ref.mean()))


def test_cli_checkpoint_resume_and_debug(tmp_path):
    """--checkpoint / --resume: 4 + 4 samplings == 8 samplings (same per-index seeds); -d writes the FocalPlane view."""
    import os
    import subprocess
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "hanamaru-renderer_amd", "hanamaru-hip")
    base = [exe, "-w", "64", "-h", "36", "-t", "1000", "-i", "1000", "--assets", os.path.join(root, "assets")]

    def run(args, sub):
        d = tmp_path / sub
        d.mkdir(exist_ok=True)
        r = subprocess.run(base + args, cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0, r.stdout
        return d, r.stdout
    d8, _ = run(["-s", "8"], "full")
    ck = str(tmp_path / "acc.bin")
    run(["-s", "4", "--checkpoint", ck], "half")
    dr, out = run(["-s", "8", "--resume", ck], "resumed")
    assert "resumed at 4x4 sampled" in out and "sampled: 8x4 spp." in out
    a, b = np.asarray(Image.open(d8 / "result.png")).astype(int), np.asarray(Image.open(dr / "result.png")).astype(int)
    assert np.abs(a - b).max() <= 1            # fp32 summation order only
    dd, out = run(["-d"], "debug")
    assert "sampled: 1x4 spp." in out
    img = np.asarray(Image.open(dd / "result.png"))
    assert img.shape == (36, 64, 3) and img.std() > 5


def _seed_mode_available(gpu, mode):
    """seed_mode 3 and 4 (measured experiments, slower than the default) are compiled only into `make EXPERIMENTS=1` builds; the
    product library answers HR_ERR_UNSUPPORTED (-6) for them."""
    import hanamaru_amd as ha
    try:
        gpu.set_debug_option("seed_mode", mode)
        return True
    except ha.HipError as e:
        assert e.code == -6 and mode >= 3, (mode, str(e))
        return False


def test_seed_kernels_are_bit_identical(gpu, scenes):
    """The five seed kernels — the three-run kernel (seed_mode 2, the default: the init sweep as three runs computed side by side from
    states the producer waves work out ahead), its phase-shifted four-run form (seed_mode 3: the halves half a period apart, barriers in
    the middle of the round, a paused ahead pass), its five-wave four-run form (seed_mode 4: roles by SIMD), the producer / consumer kernel with the ring of generator words (seed_mode 1, every
    seed_split) and the fused kernel (seed_mode 0) — must hand the trace kernel exactly the same draws: same raw tails -> the
    same accumulator, bit for bit (since round 3 a launch's radiance is summed in a fixed order: accumulate_kernel)."""
    sc, _ = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    outs = []
    try:
        for (w, h, s) in [(130, 71, 6), (640, 360, 9)]:   # ragged: tiles hang over the right and bottom edges; many groups per CU
            gpu.set_resolution(w, h)
            ref = None
            for mode, head in [(0, 16), (1, 16), (1, 8), (1, 12), (1, 24), (2, 16), (3, 16), (4, 16)]:   # 2: the three-run kernel (no state ring), 3: its phase-shifted four-run form
                if not _seed_mode_available(gpu, mode):      # modes 3 and 4: only in `make EXPERIMENTS=1` builds
                    continue
                gpu.set_debug_option("seed_mode", mode)
                gpu.set_debug_option("seed_split", head)
                gpu.clear()
                gpu.render(1, s + 1)
                acc = gpu.read_accumulator().astype(np.float64)
                if ref is None:
                    ref = acc
                    assert ref.sum() > 0
                else:
                    assert np.array_equal(ref, acc), (w, h, mode, head, np.abs(ref - acc).max())
    finally:
        gpu.set_debug_option("seed_mode", 2)
        gpu.set_debug_option("seed_split", 16)


def test_priority_governor_does_not_change_results(gpu, scenes):
    """Option trace_boost (five levels from "the seed kernel's producer waves first" to "the trace kernel's box and leaf phases first";
    -1 = decided on the device from the kernels' time stamps) only moves issue slots between the two kernels: the accumulator is the same,
    bit for bit — and so is a second render of the same samplings (no atomics: every path leaves its radiance in its own record and
    accumulate_kernel sums a pixel's records in a fixed order)."""
    sc, _ = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    gpu.set_resolution(320, 180)
    outs = []
    try:
        for boost in (0, 1, 2, 3, 4, -1, -1):
            gpu.set_option("trace_boost", boost)
            gpu.clear()
            gpu.render(1, 13)
            outs.append(gpu.read_accumulator().astype(np.float64))
        with pytest.raises(Exception):
            gpu.set_option("trace_boost", 5)
    finally:
        gpu.set_option("trace_boost", -1)
    assert outs[0].sum() > 0
    for o in outs[1:]:
        assert np.array_equal(outs[0], o), np.abs(outs[0] - o).max()


def test_nee_culls_do_not_change_a_bit(gpu, scenes):
    """nee_setup (pt_core.h) does not trace a shadow ray whose contribution is known to be exactly zero before it starts: the sample lies on
    the far side of the emitter (sample_on_surface draws over the whole sphere, scene.rs:92-101 — more than half of the samples), the surface
    is GGX and the emitter below its horizon (material.rs:64-67 returns 0).  The reference traces
    those rays and discards them (renderer.rs:279-280); with debug option nee_cull 0 so does the kernel — same accumulator, bit for bit, on
    every scene type; and the path log still counts the reference's scene.intersect calls."""
    try:
        for name, w, h, s in [("rtcamp6_v3_1", 400, 225, 4), ("rtcamp6_v2", 320, 180, 2), ("tbf3", 320, 180, 2), ("rtcamp5", 320, 180, 2), ("rtcamp6_v1", 256, 144, 2),
                              ("spheres", 320, 180, 2), ("cornell_mini", 200, 128, 4), ("material_examples", 256, 144, 2)]:
            sc, _ = scenes(name)
            gpu.upload_scene(sc)
            gpu.set_resolution(w, h)
            out = {}
            for cull in (7, 0):       # the mask of shortcuts in force: 7 = all (default), 0 = every shadow ray traced
                gpu.set_debug_option("nee_cull", cull)
                gpu.set_option("counters", 1)
                gpu.clear()
                gpu.render(1, 1 + s)
                st = gpu.stats()
                gpu.set_option("counters", 0)
                out[cull] = (gpu.read_accumulator().copy(), st, gpu.debug_path_log(1))
            (a, sa, la), (b, sb, lb) = out[7], out[0]
            assert a.sum() > 0 and np.array_equal(a, b), (name, np.abs(a - b).max())
            assert sb["shadow_culled"] == 0 and sa["shadow_culled"] > 0
            assert sa["rays"] + sa["shadow_culled"] == sb["rays"], (name, sa["rays"], sa["shadow_culled"], sb["rays"])
            for x, y in zip(la, lb):      # radiance, ray counts (culled rays included), events, hashes of every path
                assert np.array_equal(x, y), name
            print("nee culls %s: %.3f of %.3f rays per path not traced (%.1f %% of the node tests)" %
                  (name, sa["shadow_culled"] / sa["paths"], sb["rays"] / sb["paths"], 100.0 * (1.0 - sa["node_tests"] / sb["node_tests"])))
    finally:
        gpu.set_debug_option("nee_cull", 7)
        gpu.set_option("counters", 0)


def test_kernel_variants_render_the_same_bits(gpu, scenes):
    """The instrumented build (counters), the occupancy variants (min_waves 4 / 6), the walk on the 32-byte fp32 records (quant_nodes 0)
    and a device-built tree are other INSTRUCTION STREAMS for the same arithmetic: closest hits do not depend on the tree or the
    record format, every product is an explicit FMA or not one, and a launch's radiance is summed in a fixed order — so the accumulator
    is the same, bit for bit."""
    sc, _ = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    gpu.set_resolution(400, 225)

    def render():
        gpu.clear()
        gpu.render(1, 9)
        return gpu.read_accumulator().copy()
    ref = render()
    assert ref.sum() > 0
    try:
        gpu.set_option("counters", 1)
        assert np.array_equal(ref, render()), "instrumented build"
        gpu.set_option("counters", 0)
        for mw in (4, 6):
            gpu.set_debug_option("min_waves", mw)
            assert np.array_equal(ref, render()), ("min_waves", mw)
        gpu.set_debug_option("min_waves", 5)
        # the split pipeline (traversal kernel + shading kernel per path iteration, wf_kernels.h) is the megakernel cut at scene.intersect
        log_mega = gpu.debug_path_log(3)
        gpu.set_debug_option("trace_mode", 1)
        assert np.array_equal(ref, render()), "split pipeline"
        # ... and its LOG instantiation (the event log rides in the path's state) writes the megakernel's path log, word for word
        for a, b in zip(log_mega, gpu.debug_path_log(3)):
            assert np.array_equal(a, b), "split pipeline, path log"
        gpu.set_option("counters", 1)
        assert np.array_equal(ref, render()), "split pipeline, instrumented build"
        gpu.set_option("counters", 0)
        # option precise_shading has two homes too — path_advance<.., PREC> in the megakernel (the residuals parked in the path's record) and
        # the split pipeline's shading kernel (the residuals in the queued state): one function (prec_core.h shade_hit_f64), the same bits
        gpu.set_option("precise_shading", 1)
        prec_split = render()
        prec_split_log = gpu.debug_path_log(3)
        gpu.set_debug_option("trace_mode", 0)
        assert np.array_equal(prec_split, render()), "precise shading: megakernel vs split pipeline"
        assert not np.array_equal(prec_split, ref)
        for a, b in zip(prec_split_log, gpu.debug_path_log(3)):
            assert np.array_equal(a, b), "precise shading, path log"
        gpu.set_option("counters", 1)
        assert np.array_equal(prec_split, render()), "precise shading, instrumented build"
        gpu.set_option("counters", 0)
        gpu.set_option("precise_shading", -1)
        gpu.set_debug_option("trace_mode", 1)
        gpu.set_option("quant_nodes", 0)
        gpu.upload_scene(sc)
        assert np.array_equal(ref, render()), "split pipeline, fp32 node records"
        gpu.set_debug_option("trace_mode", 0)
        assert np.array_equal(ref, render()), "fp32 node records"
        gpu.set_option("quant_nodes", 1)
        gpu.set_option("bvh_builder", 2)
        gpu.upload_scene(sc)
        assert np.array_equal(ref, render()), "device-built tree"
    finally:
        gpu.set_option("counters", 0)
        gpu.set_option("precise_shading", -1)
        gpu.set_debug_option("trace_mode", -1)
        gpu.set_debug_option("min_waves", 5)
        gpu.set_option("quant_nodes", 1)
        gpu.set_option("bvh_builder", -1)


def test_priority_governor_decides_on_the_device(gpu, scenes):
    """The governor lives in device memory (GovDev, governor_kernel): inside ONE hr_render call — the host never waits — it judges the
    launches as they finish (all but the first and last of the burst, whose kernels ran alone for part of their time), and a fixed
    level is what the kernels run at."""
    sc, _ = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    gpu.set_resolution(1920, 1080)
    try:
        gpu.set_option("trace_boost", -1)
        st0 = gpu.stats()
        assert st0["governor_decisions"] == 0 and st0["governor_level"] == 0      # a new balance starts at "seed kernel first"
        gpu.clear()
        gpu.render(1, 1 + 4 * 24)                                                  # 24 launches of 4 samplings, one call
        gpu.synchronize()
        st = gpu.stats()
        assert st["trace_launches"] == 24                                          # (hr_clear zeroed the launch counters)
        assert 16 <= st["governor_decisions"] <= 23, st["governor_decisions"]     # not the first, not the last
        assert 0 <= st["governor_level"] <= 4 and st["governor_moves"] <= st["governor_decisions"]
        for level in (3, 0):
            gpu.set_option("trace_boost", level)
            gpu.clear()
            gpu.render(1, 9)
            gpu.synchronize()
            st = gpu.stats()
            assert st["governor_level"] == level and st["governor_decisions"] == 0 and st["governor_moves"] == 0
    finally:
        gpu.set_option("trace_boost", -1)


def test_wave_budget_governor(gpu, scenes):
    """The governor's second control (round 5): where the trace kernel is the faster kernel of the pair by a margin, part of its persistent
    workgroups leave at once (GovDev::budget) — the seed kernel beside it gains what their waves no longer take.  On the headline workload the
    budget settles between 2.5 and 3.5 workgroups per CU within a few launches; on a trace-bound scene every workgroup stays and the priority
    levels work as before; a fixed level pins the budget at "all"; and the accumulator does not depend on any of it (fixed summation order)."""
    sc, _ = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    gpu.set_resolution(1920, 1080)
    gpu.set_option("batch", 4)
    try:
        gpu.set_option("trace_boost", -1)
        gpu.clear()
        gpu.render(1, 97)                 # 24 launches in ONE call: the host never waits, everything is decided on the device
        st = gpu.stats()
        governed = gpu.read_accumulator().copy()
        cus = 256
        print("wave budget: headline settles at %d trace workgroups (%d moves, %d launches judged), level %d" % (st["governor_budget"], st["governor_budget_moves"], st["governor_decisions"], st["governor_level"]))
        assert st["governor_level"] == 0 and st["governor_budget_moves"] >= 2 and st["governor_decisions"] >= 12
        assert cus * 5 // 2 <= st["governor_budget"] <= cus * 7 // 2 and st["governor_budget"] % (cus // 4) == 0
        gpu.set_option("trace_boost", 0)  # a fixed level: no budget
        gpu.clear()
        gpu.render(1, 97)
        st0 = gpu.stats()
        assert st0["governor_budget"] == 0 and st0["governor_budget_moves"] == 0
        assert np.array_equal(governed, gpu.read_accumulator())
        gpu.set_option("trace_boost", -1)
        gpu.set_debug_option("trace_budget", 512)      # pinned from the host: same bits again
        gpu.clear()
        gpu.render(1, 97)
        assert np.array_equal(governed, gpu.read_accumulator())
        gpu.set_debug_option("trace_budget", 0)
        # a scene whose trace kernel is the slower one keeps every workgroup
        sc2, _ = scenes("rtcamp6_v2")
        gpu.upload_scene(sc2)
        gpu.clear()
        gpu.render(1, 65)
        st2 = gpu.stats()
        print("wave budget: rtcamp6_v2 keeps %s workgroups, level %d" % ("all" if st2["governor_budget"] == 0 else st2["governor_budget"], st2["governor_level"]))
        assert st2["governor_budget"] == 0 and st2["governor_level"] >= 2
    finally:
        gpu.set_option("trace_boost", -1)
        gpu.set_debug_option("trace_budget", 0)
        gpu.set_option("batch", 0)


def _run_bench(extra, env=None, launcher_ranks=0, port=29541):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--warmup", "0", "--spp-per-step", "2", "--width", "160", "--height", "90", "--no-cpu-baseline"]
    cmd = [sys.executable]
    if launcher_ranks:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher_ranks), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(root, "bench.py")] + extra + common
    e = dict(os.environ, HR_BENCH_CHECKSUM="1")
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)


def _bench_line(proc):
    import json
    import re
    assert proc.returncode == 0, proc.stderr[-3000:]
    mean = float(re.search(r"accumulator mean after all-reduce: ([0-9.eE+-]+)", proc.stderr).group(1))
    return json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1]), mean


def test_bench_gpus_n_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` exactly as the driver runs N = 1 — no torch.distributed.run: ONE process drives two contexts,
    hr_comm_init_local + hr_allreduce_accumulators.  On this 1-GPU box both contexts share device 0 (the library's same-device
    sum, labelled FALLBACK in config.parallelism); with two GPUs the same calls are one RCCL group all-reduce.  The summed
    accumulator must equal a single-rank run over the same sampling indices (samplings 1..8 either way)."""
    j1, m1 = _bench_line(_run_bench(["--steps", "4"]))
    j2, m2 = _bench_line(_run_bench(["--gpus", "2", "--steps", "2"]))
    assert abs(m1 - m2) <= 1e-6 * m1, (m1, m2)
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["scaling"] == "weak" and j2["value"] > 0
    import torch
    par = j2["config"]["parallelism"]
    assert "spp-sharded x2" in par and ("RCCL group all-reduce" in par if torch.cuda.device_count() >= 2 else "FALLBACK" in par), par
    # the bench line's roofline: frac IS SURVEY 8(d)'s figure, the loaded bytes can not exceed that booking, and the traversal-only workload is there
    r = j1["roofline"]
    assert r["bound"] == "l1_ta_issue" and r["bound_contract"] == "hbm" and r["pair_bound"] in ("seed_seg_kernel", "trace_kernel") and r["peak"] == 8000.0   # (pair_bound: the slower kernel of THIS run — at 160x90 either)
    assert 0 < r["frac"] == r["frac_survey_8d"] and 0 < r["loaded_bytes"]["frac"] <= r["frac"] and "frac_alone" not in r
    assert abs(r["achieved"] * 1e9 * r["avg_launch_ms"] * 1e-3 - r["algorithmic_bytes_per_launch"]) <= 2e-3 * r["algorithmic_bytes_per_launch"]
    assert 0 < r["l2"]["frac"] < 1 and 0 < r["l2"]["frac_alone"] < 1 and r["hbm_normalised_alone"] > 0   # (at 160x90 "alone" is not reliably the faster one)
    assert 0 < r["traversal_section"]["share_of_wave_cycles"] < 1 and r["traversal_only"]["Mrays_per_s"] > 0 and 0 < r["traversal_only"]["frac"] < 1
    assert "cpu_baseline" not in j1 and j1["config"]["estimator"].startswith("reference")
    # the workload string states what was run, not a constant
    assert j1["config"]["samplings_total"] == 8 and "x 8 samplings" in j1["config"]["workload"] and "4 steps x 2 samplings" in j1["config"]["workload"]
    assert j2["config"]["samplings_total"] == 8 and j2["config"]["paths_total"] == 160 * 90 * 4 * 8
    assert j1["post_chain"]["post_kernel_ms"] > 0 and j1["post_chain"]["resolution"] == [160, 90]
    # the timed region's split, per rank
    m = j2["multi_gpu"]
    assert m["accumulator_bytes"] == 160 * 90 * 3 * 4 and len(m["per_rank"]) == 2 and [x["rank"] for x in m["per_rank"]] == [0, 1]
    # the line proves its own exchange: what the communicators said about themselves, and parts == total
    assert j1["multi_gpu"]["rccl"]["nranks"] == 1 and j1["multi_gpu"]["rccl"]["path"] == "none" and j1["multi_gpu"]["exchange_verified"]
    assert j1["multi_gpu"]["checksum"]["rel_err"] == 0.0 and j1["multi_gpu"]["checksum"]["total"][0] > 0
    rc, ck = m["rccl"], m["checksum"]
    assert rc["nranks"] == 2 and rc["ranks_seen"] == [0, 1] and rc["allreduces_per_rank"] == [2] and m["exchange_verified"]       # warm-up + the timed one
    if torch.cuda.device_count() >= 2:
        assert rc["path"] == "rccl-group" and rc["devices_seen"] == [0, 1] and rc["version"] > 20000
    else:
        assert rc["path"] == "same-device-fallback" and rc["devices_seen"] == [0]
    assert ck["rel_err"] < 1e-6 and ck["totals_identical_on_all_ranks"] and abs(sum(ck["total"]) / (160 * 90 * 3) - m2) <= 1e-5 * m2
    assert all(x["paths"] == 160 * 90 * 4 * 4 and x["render_ms"] > 0 and x["allreduce_ms"] > 0 and x["seed_kernel_ms"] > 0 for x in m["per_rank"])
    assert m["render_ms"]["max"] >= m["render_ms"]["min"] > 0 and 0 < m["allreduce_share_of_timed_region"] < 1
    # three contexts, odd step count
    j3, m3 = _bench_line(_run_bench(["--gpus", "3", "--steps", "1", "--no-counters"]))
    _, m3ref = _bench_line(_run_bench(["--steps", "3", "--no-counters"]))
    assert j3["n_gpus"] == 3 and abs(m3 - m3ref) <= 1e-6 * m3ref


def test_bench_gpus_8_and_total_samplings_on_one_device(tmp_path):
    """The command lines of BASELINE configs 4 and 5 with everything but the node: `bench.py --gpus 8` without a launcher (ONE process,
    eight contexts, hr_comm_init_local + hr_allreduce_accumulators — eight same-device contexts here, one RCCL group on an 8-GPU
    node) and `--total-samplings S` (strong scaling: exactly samplings 1..S, sharded (s - 1) mod 8, last step clipped).  The summed
    accumulator must equal a single-context run over the same sampling indices."""
    j8, m8 = _bench_line(_run_bench(["--gpus", "8", "--steps", "2", "--no-counters"]))          # 2 steps x 2 samplings x 8 contexts = samplings 1..32
    _, m1 = _bench_line(_run_bench(["--steps", "16", "--no-counters"]))
    assert j8["n_gpus"] == 8 and j8["scaling"] == "weak" and abs(m8 - m1) <= 1e-6 * m1, (m8, m1)
    assert len(j8["multi_gpu"]["per_rank"]) == 8 and j8["config"]["samplings_total"] == 32
    # strong scaling: 21 samplings over 8 contexts in 2 steps -> 2 per context per step, the second step clipped (ranks 0..4 get 3, 5..7 get 2)
    js, ms = _bench_line(_run_bench(["--gpus", "8", "--steps", "2", "--total-samplings", "21", "--no-counters"]))
    jr, mr = _bench_line(_run_bench(["--total-samplings", "21", "--steps", "3", "--no-counters"]))
    assert js["scaling"] == "strong" and jr["scaling"] == "strong" and js["config"]["samplings_total"] == 21 == jr["config"]["samplings_total"]
    assert js["config"]["paths_total"] == 160 * 90 * 4 * 21 and abs(ms - mr) <= 1e-6 * mr, (ms, mr)
    assert sorted(x["paths"] // (160 * 90 * 4) for x in js["multi_gpu"]["per_rank"]) == [2, 2, 2, 3, 3, 3, 3, 3]
    assert "clipped to --total-samplings 21" in js["config"]["workload"]


def test_bench_multirank_path_on_one_gpu(tmp_path):
    """bench.py's launcher path (one process per rank under torch.distributed.run: sharding by sampling index, one all-reduce) run
    as two ranks on ONE GPU (HR_BENCH_ONE_DEVICE: gloo on a host copy, RCCL refuses two ranks per device): the summed accumulator
    must equal a single-rank run over the same sampling indices."""
    j1, m1 = _bench_line(_run_bench(["--steps", "4", "--no-counters"]))
    j2, m2 = _bench_line(_run_bench(["--gpus", "2", "--steps", "2", "--no-counters"], env={"HR_BENCH_ONE_DEVICE": "1"}, launcher_ranks=2))
    assert abs(m1 - m2) <= 1e-6 * m1
    assert j2["n_gpus"] == 2 and j2["scaling"] == "weak" and j2["value"] > 0 and "roofline" in j2
    assert j2["multi_gpu"]["checksum"] is None and "gloo" in j2["multi_gpu"]["rccl"]["source"]      # the aid is labelled as what it is


def _two_gpus():
    import torch
    return torch.cuda.device_count() >= 2


@pytest.mark.skipif("not _two_gpus()", reason="needs two GPUs (skipped on the 1-GPU box, live on any bigger one)")
def test_rccl_group_over_two_distinct_devices(scenes, ha):
    """hr_comm_init_local over two contexts on two DIFFERENT devices: ncclCommInitAll + one RCCL group all-reduce over xGMI (the path `python
    bench.py --gpus N` and the CLI take on a multi-GPU node).  Sharded by sampling index, the total both contexts receive must be the
    accumulator of one context that rendered every sampling (fp32 summation order aside), the communicators must say 2 ranks on devices 0 and
    1, and the parts' f64 sums must add up to the total's."""
    sc, _ = scenes("rtcamp6_v3_1")
    rs = [ha.Renderer(d) for d in (0, 1)]
    try:
        for r in rs:
            r.upload_scene(sc)
            r.set_resolution(320, 180)
        ha.comm_init_local(rs)
        infos = [r.comm_info() for r in rs]
        assert [i["path"] for i in infos] == ["rccl-group"] * 2 and [i["nranks"] for i in infos] == [2, 2]
        assert sorted(i["rank"] for i in infos) == [0, 1] and sorted(i["device"] for i in infos) == [0, 1] and infos[0]["rccl_version"] > 20000
        for k, r in enumerate(rs):
            r.render(1 + k, 9, 2)
        ha.allreduce_accumulators(rs)
        tots = [r.read_accumulator() for r in rs]
        assert np.array_equal(tots[0], tots[1])
        parts = np.sum([r.accumulator_sum(False) for r in rs], axis=0)
        total = np.array(rs[0].accumulator_sum(True))
        assert (np.abs(parts - total) <= 1e-6 * np.abs(total)).all() and total.min() > 0
        one = ha.Renderer(0)
        one.upload_scene(sc)
        one.set_resolution(320, 180)
        one.render(1, 9)
        ref = one.read_accumulator().astype(np.float64)
        one.close()
        assert np.abs(tots[0] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
        assert [r.comm_info()["allreduces"] for r in rs] == [1, 1]
        # a device listed twice next to another one is refused before RCCL sees it
        extra = ha.Renderer(0)
        try:
            with pytest.raises(ha.HipError):
                ha.comm_init_local([rs[0], rs[1], extra])
        finally:
            extra.close()
    finally:
        for r in rs:
            r.close()


@pytest.mark.skipif("not _two_gpus()", reason="needs two GPUs (skipped on the 1-GPU box, live on any bigger one)")
def test_bench_launcher_path_with_real_rccl_on_two_gpus(tmp_path):
    """bench.py exactly as the driver launches N = 2: torch.distributed.run, one process per GPU, hr_comm_init_rank from a broadcast
    ncclUniqueId, ncclAllReduce inside the library.  The line must prove its exchange (2 ranks, 2 devices, path rccl-rank, parts == total)
    and the total must be a single-rank run's over the same sampling indices."""
    j1, m1 = _bench_line(_run_bench(["--steps", "4", "--no-counters"]))
    j2, m2 = _bench_line(_run_bench(["--gpus", "2", "--steps", "2", "--no-counters"], launcher_ranks=2))
    assert abs(m1 - m2) <= 1e-6 * m1
    rc, ck = j2["multi_gpu"]["rccl"], j2["multi_gpu"]["checksum"]
    assert rc["path"] == "rccl-rank" and rc["nranks"] == 2 and rc["ranks_seen"] == [0, 1] and rc["devices_seen"] == [0, 1] and rc["version"] > 20000
    assert ck["rel_err"] < 1e-6 and ck["totals_identical_on_all_ranks"] and j2["multi_gpu"]["exchange_verified"]


def test_config5_4k_crops(gpu, scenes):
    """BASELINE config 5 (rtcamp6 + fractal dodecahedron, 3840x2160) at its full size on one GPU: exact path count, the hand-off
    buffer cap shrinking the launch, and oracle parity on 64x64 crops at full-image coordinates: the dodecahedron (deep
    BVH, refraction), the bunny's ear silhouette, the picture frame's edge against the sky, floor + armadillo, open sky."""
    sc, o = scenes("rtcamp6_dodeca")
    gpu.upload_scene(sc)
    gpu.set_option("counters", 1)
    W, H, S = 3840, 2160, 4
    acc = _crop_parity(gpu, o, "rtcamp6_dodeca", W, H, S, [(1950, 150), (1850, 520), (3040, 640), (1620, 1560), (380, 320)])
    st = gpu.stats()
    assert st["paths"] == W * H * 4 * S and st["rng_overflow"] == 0 and st["trace_launches"] == 1
    gpu.set_option("max_tail_gib", 4)                    # one 4K sampling of hand-off records is 4.25 GB: the cap shrinks the launch to 1
    gpu.clear()
    gpu.render(1, 3)
    st = gpu.stats()
    gpu.set_option("max_tail_gib", 20)
    gpu.set_option("counters", 0)
    assert st["paths"] == W * H * 4 * 2 and st["trace_launches"] == 2
    capped = gpu.read_accumulator()
    gpu.clear()
    gpu.render(1, 3)
    whole = gpu.read_accumulator()
    assert np.abs(capped.astype(np.float64) - whole).max() <= 1e-4 * max(1.0, float(np.abs(whole).max()))


def test_config5_4k_crops_precise_shading(gpu, scenes):
    """The same crops with option precise_shading: the split pipeline at 3840x2160 (its queues are sized for the worst case and kept under
    max_tail_gib: with a cap of 8 GiB a launch holds ONE 4K sampling, 33 M paths), exact path count, parity on the crops, and the same
    accumulator (to the summation order) whatever the launch size."""
    sc, o = scenes("rtcamp6_dodeca")
    gpu.upload_scene(sc)
    W, H, S = 3840, 2160, 2
    gpu.set_option("precise_shading", 1)
    gpu.set_option("counters", 1)
    gpu.set_option("max_tail_gib", 8)                     # this scene's queues take 10.6 GB per 4K sampling (one emitter): a launch per sampling
    try:
        a = _crop_parity(gpu, o, "rtcamp6_dodeca", W, H, S, [(1950, 150), (1850, 520), (3040, 640)])
        st = gpu.stats()
        assert st["paths"] == W * H * 4 * S and st["rng_overflow"] == 0 and st["trace_launches"] == 2
        gpu.set_option("counters", 0)
        gpu.set_option("max_tail_gib", 64)                # room for both samplings in one launch
        gpu.clear()
        gpu.render(1, S + 1)
        whole = gpu.read_accumulator()      # (another launch size is another summation order of a pixel's records: equal to fp32 rounding)
        assert np.abs(a.astype(np.float64) - whole).max() <= 1e-4 * max(1.0, float(np.abs(whole).max()))
    finally:
        gpu.set_option("max_tail_gib", 20)
        gpu.set_option("counters", 0)
        gpu.set_option("precise_shading", -1)


def test_precise_shading_defaults_and_finite_radiance(gpu, scenes):
    """Option precise_shading = -1 (default): ON for scenes without triangle meshes (in the megakernel), OFF for mesh scenes; pinned on, a
    mesh scene takes the split pipeline (same bits, faster there).  And the regression of round 6's one NaN: rtcamp5's roughness map has texels
    of ~1e-8, where the reference's own GGX half-vector expression (material.rs:264-265) rounds to sqrt(-4e-16); sampling 192 of the 1080p
    frame holds such a path."""
    sc, _ = scenes("spheres")
    gpu.upload_scene(sc)
    assert gpu.stats()["shading_in_force"] == 1
    gpu.set_option("russian_roulette", 3)
    assert gpu.stats()["shading_in_force"] == 0          # the roulette estimator has no f64 instantiation: the automatic choice stands back
    gpu.set_option("russian_roulette", 0)
    sc, _ = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    assert gpu.stats()["shading_in_force"] == 0
    gpu.set_option("precise_shading", 1)
    assert gpu.stats()["shading_in_force"] == 2          # a mesh scene: the split form is the faster one
    gpu.set_option("precise_shading", -1)
    sc, _ = scenes("rtcamp5")
    gpu.upload_scene(sc)
    assert gpu.stats()["shading_in_force"] == 0
    gpu.set_option("precise_shading", 1)
    try:
        assert gpu.stats()["shading_in_force"] == 2      # any mesh: the split form
        gpu.set_resolution(1920, 1080)
        gpu.clear()
        gpu.render(191, 194)
        split = gpu.read_accumulator().copy()
        assert np.isfinite(split).all()
        gpu.set_debug_option("trace_mode", 0)
        assert gpu.stats()["shading_in_force"] == 1      # pinned: the megakernel form, the same bits
        gpu.clear()
        gpu.render(191, 194)
        assert np.array_equal(split, gpu.read_accumulator())
    finally:
        gpu.set_debug_option("trace_mode", -1)
        gpu.set_option("precise_shading", -1)


@pytest.mark.parametrize("precise", [0, 1])
def test_ggx_nee_sample_exactly_at_the_horizon_adds_nothing(gpu, scenes, precise):
    """The non-finite pixel tools/finite_soak.py found (every scene at 1920x1080 x 1,024 samplings, both shading modes): cornell_mini, sampling
    732, pixel (1394, 371) — a GGX hit whose NEE sample lies EXACTLY on the horizon of the shaded face after fp32 rounding (l.n = +0: 0 / 0 in
    material.rs:64-89's expression; tests/test_emu_parity.py has the path on the host).  The sampling is finite and the pixel is the oracle's."""
    sc, o = scenes("cornell_mini")
    gpu.upload_scene(sc)
    gpu.set_resolution(1920, 1080)
    gpu.set_option("precise_shading", precise)
    try:
        gpu.clear()
        gpu.render(732, 733)
        acc = gpu.read_accumulator().copy()
    finally:
        gpu.set_option("precise_shading", -1)
    assert np.isfinite(acc).all()
    ref = o.render_region(1920, 1080, 1394, 371, 1, 1, 732, 733, threads=1)[0, 0]
    assert np.abs(acc[371, 1394].astype(np.float64) - ref).max() <= 1e-4, (acc[371, 1394], ref)


@pytest.mark.parametrize("seed,precise", [(6438, 0), (6438, 1), (6219, 0)])
def test_random_scenes_rendered_long_stay_finite(gpu, ha, seed, precise):
    """tools/finite_fuzz.py (1,000 random scenes x 640x360 x 256 samplings x both shading modes) found two scenes with non-finite pixels: a
    near-mirror GGX from a roughness map's ~0 texel whose NEE term overflows fp32 (seed 6438: +inf where the reference's f64 carries 1e29), and
    one NaN of an approximate reciprocal (seed 6219, fp32 shading).  accumulate_kernel takes a path's radiance in through path_radiance_in: the
    image stays finite, the overflowing path's pixel is white."""
    import random_scenes
    kw = dict(spheres=2, cuboids=1, meshes=1) if seed % 4 == 2 else dict(spheres=0, cuboids=6, meshes=2)
    sc = random_scenes.build(ha, seed, **kw)
    gpu.set_option("bvh_builder", seed % 3)
    try:
        gpu.upload_scene(sc)
    finally:
        gpu.set_option("bvh_builder", -1)
    gpu.set_resolution(640, 360)
    gpu.set_option("precise_shading", precise)
    try:
        gpu.clear()
        gpu.render(1, 257)
        acc = gpu.read_accumulator().copy()
        img = gpu.resolve(256)
    finally:
        gpu.set_option("precise_shading", -1)
    assert np.isfinite(acc).all() and acc.max() < 1e34
    assert img.shape == (360, 640, 3)


def test_config5_full_length_through_the_cli(tmp_path, scenes, orc):
    """BASELINE config 5 end to end at its FULL length on one GPU: `hanamaru-hip --scene rtcamp6_dodeca -w 3840 -h 2160 -s 1024` —
    3.4e10 paths (half a minute), the reference's log lines, and the 4K PNG that comes out of accumulate -> Reinhard -> gamma ->
    bilateral -> u8 compared with the oracle's post chain of its own 1,024-sampling accumulator on crops at full-image coordinates
    (the dodecahedron, the bunny's ear, the frame against the sky, floor + armadillo, open sky)."""
    import re
    from PIL import Image
    W, H, S = 3840, 2160, 1024
    out = _run_cli(tmp_path, ["--scene", "rtcamp6_dodeca", "-w", W, "-h", H, "-s", S, "-t", "100000", "-i", "100000"], timeout=900)
    assert "reached max sampling" in out and ("sampled: %dx4 spp." % S) in out and "output final image: 000.png" in out
    m = re.search(r"gpu: ([0-9.]+) Mpaths/s wall", out)
    img = np.asarray(Image.open(tmp_path / "result.png"))
    assert img.shape == (H, W, 3)
    print("config 5 full length through the CLI: %s Mpaths/s wall, image mean %.3f" % (m.group(1) if m else "?", img.mean()))
    assert m and float(m.group(1)) > 600.0         # a sanity floor far below the bench's rate (scene set-up and PNG excluded by the CLI's own clock)
    _, o = scenes("rtcamp6_dodeca")
    _crops_match_oracle(img, o, orc, W, H, S, [(1970, 170), (1870, 540), (3060, 660), (1640, 1580), (400, 340)])


def test_config4_full_length_through_the_cli(tmp_path, scenes, orc):
    """BASELINE config 4 at its FULL length (1920x1080 x 4,096 samplings = 3.4e10 paths) through the CLI with the samplings sharded over
    two contexts (`--gpu-ids 0,0`: all this box has is one device, so both shards run on it and hr_allreduce_accumulators sums them with
    its same-device kernel instead of RCCL — the sharding, the strides and the sum are those of the 8-GPU run): the PNG against the
    oracle's post chain of its own 4,096-sampling accumulator on crops."""
    import re
    from PIL import Image
    W, H, S = 1920, 1080, 4096
    out = _run_cli(tmp_path, ["-w", W, "-h", H, "-s", S, "-t", "100000", "-i", "100000", "--gpu-ids", "0,0"], timeout=900)
    assert "reached max sampling" in out and ("sampled: %dx4 spp." % S) in out
    m = re.search(r"gpu: ([0-9.]+) Mpaths/s wall", out)
    img = np.asarray(Image.open(tmp_path / "result.png"))
    assert img.shape == (H, W, 3)
    print("config 4 full length through the CLI, two shards on one device: %s Mpaths/s wall, image mean %.3f" % (m.group(1) if m else "?", img.mean()))
    assert m and float(m.group(1)) > 600.0
    _, o = scenes("rtcamp6_v3_1")
    _crops_match_oracle(img, o, orc, W, H, S, [(940, 330), (1250, 420), (700, 760), (100, 100)])


def test_config5_4k_post_chain(gpu, scenes, orc):
    """BASELINE config 5's post chain at its full size: hr_resolve (tonemap_gamma_kernel + bilateral_quantise_kernel, renderer.rs:64-90) at
    3840x2160 against orc.resolve on the SAME accumulator — the rendered one, then a synthetic high-dynamic-range one (values from 0 to
    1e4 next to each other: the bilateral filter's range weight and the Reinhard white point at work).  <= 1 LSB everywhere, > 99 % exact,
    and the same on the border rows and columns, where filter.rs:32-58's u32 arithmetic wraps (x - 1 at x = 0) before it clamps."""
    sc, _ = scenes("rtcamp6_dodeca")
    gpu.upload_scene(sc)
    W, H = 3840, 2160
    gpu.set_resolution(W, H)
    gpu.clear()
    gpu.render(1, 3)
    acc = gpu.read_accumulator()
    rng = np.random.default_rng(5)
    syn = (np.exp(rng.normal(0.0, 2.5, size=(H, W, 3))) * 4.0).astype(np.float32)
    syn[::97, ::89] = 0.0
    syn[5::211, 7::193] = 1e4
    syn[0, :, 0] = 50.0                                     # bright border row / column next to dark neighbours
    syn[:, -1, 2] = 0.0
    ms0 = gpu.stats()["post_kernel_ms"]
    for what, a, s in (("rendered", acc, 2), ("synthetic", syn, 1)):
        gpu.write_accumulator(a)
        img = gpu.resolve(s)
        exp = orc.resolve(a.astype(np.float64), s)
        d = np.abs(img.astype(np.int16) - exp.astype(np.int16))
        border = np.concatenate([d[0].ravel(), d[-1].ravel(), d[:, 0].ravel(), d[:, -1].ravel()])
        print("4K post chain (%s accumulator): max diff %d LSB, exact %.5f, border exact %.5f, image mean %.3f / %.3f" % (
            what, d.max(), (d == 0).mean(), (border == 0).mean(), img.mean(), exp.mean()))
        assert img.shape == (H, W, 3) and d.max() <= 1 and (d == 0).mean() > 0.99, (what, d.max(), (d == 0).mean())
        for edge in (d[0], d[-1], d[:, 0], d[:, -1], d[:2, :2], d[-2:, -2:]):
            assert edge.max() <= 1 and (edge == 0).mean() > 0.98, what
        assert img.std() > 10
    st = gpu.stats()
    print("4K post chain: %.3f ms per hr_resolve (two kernels, HIP events)" % ((st["post_kernel_ms"] - ms0) / 2))
    assert 0 < st["post_kernel_ms"] - ms0 < 200.0


def test_config2_spheres_full_size_crops(gpu, scenes):
    """BASELINE config 2 (spheres only, Diffuse + Specular, 1920x1080 x 64 samplings) at its full size: path count, no
    triangle tests, and oracle parity on 64x64 crops: a sphere's silhouette, two overlapping spheres, cloud / sky only,
    the bottom-left image corner (spheres cut by the border), the inside of a sphere."""
    sc, o = scenes("spheres")
    gpu.upload_scene(sc)
    gpu.set_option("counters", 1)
    # 64 samplings = 256 paths per pixel: a pixel is off by more than 1e-3 as soon as ONE of them took another branch (grazing
    # sphere rims), so the 1e-3 fraction falls with the sampling count while the 1e-2 fraction rises — gates for this count
    _crop_parity(gpu, o, "spheres", 1920, 1080, 64, [(900, 190), (400, 160), (1150, 590), (0, 1016), (760, 900)], gates=(0.9996, 0.9985), crop_slack=(0.0008, 0.003))
    st = gpu.stats()
    gpu.set_option("counters", 0)
    assert st["paths"] == 1920 * 1080 * 4 * 64 and st["rng_overflow"] == 0 and st["tri_tests"] == 0 and st["sphere_tests"] > 0


@pytest.mark.parametrize("precise", [0, 1])
def test_matches_the_reference_binarys_committed_render(gpu, scenes, precise):
    """tests/golden/reference_rtcamp6_1000x4spp.png is the image the REFERENCE binary produced (committed in its repository
    as rtcamp6_1000x4spp.png, README.md:19): default scene, 1920x1080, `-s 1000`.  It is the one output of the real Rust
    program available here, so it pins everything at once: per-path ISAAC-64 seeding incl. the u64->f64 conversion, the
    scene, the JPEG/PNG decoders, the estimator, tone map, bilateral filter and quantisation.  Identical seeds mean
    identical Monte-Carlo noise, so the images agree far better than two independent 4000-spp renders would."""
    import os
    from PIL import Image
    ref = np.asarray(Image.open(os.path.join(os.path.dirname(__file__), "golden", "reference_rtcamp6_1000x4spp.png")).convert("RGB")).astype(np.float64)
    assert ref.shape == (1080, 1920, 3)
    sc, _ = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    gpu.set_resolution(1920, 1080)
    gpu.set_option("batch", 0)
    gpu.set_option("precise_shading", precise)      # (automatic = fp32 shading on this scene; 1: the split pipeline with f64 bounces)
    try:
        gpu.clear()
        gpu.render(1, 1001)
        img = gpu.resolve(1000).astype(np.float64)
    finally:
        gpu.set_option("precise_shading", -1)
    d = np.abs(img - ref)
    psnr = 10 * np.log10(255.0 ** 2 / (d ** 2).mean())
    print("vs the reference binary's image, precise %d: PSNR %.2f dB, mean abs diff %.4f, exact channels %.5f, within 1 LSB %.6f, max %d" % (precise, psnr, d.mean(), (d == 0).mean(), (d <= 1).mean(), d.max()))
    # measured (profiles/r06_reference_image_compare.txt): fp32 shading PSNR 76.2 dB, 99.858 % of the channels identical, 99.9974 % within 1 LSB,
    # max 6; precise shading 77.8 dB, 99.895 %, 99.9990 %, max 3.  (The ORACLE's image is byte-identical to it: tests/test_oracle.py.)  What is
    # left with precise shading is the fp32 accumulator and post chain at quantisation borders, and the divergent paths.
    if precise:
        assert psnr > 74.0 and (d == 0).mean() > 0.998 and (d <= 1).mean() > 0.99995 and d.max() <= 6
    else:
        assert psnr > 70.0 and (d == 0).mean() > 0.997 and (d <= 1).mean() > 0.9998 and d.max() <= 12


@pytest.mark.parametrize("name,max_leaf", [("rtcamp6_v3_1", 4), ("rtcamp6_dodeca", 4), ("spheres", 2), ("cornell_mini", 1)])
def test_device_bvh_build_is_interchangeable(gpu, scenes, name, max_leaf):
    """Options bvh_builder = 1 (LBVH) and 2 (PLOC) build the tree on the GPU (csrc/gpu_bvh.h) and emit it in the trace kernel's
    16-byte preorder format.  Closest hits do not depend on the tree: against the host-SAH tree over the same fp32 primitives
    the hit flag and t are identical bit for bit (pt_core.h spells its FMAs out, so a primitive test returns the same bits in
    every leaf slot); the radiance accumulator then differs only through equal-t ties between adjacent triangles and the
    atomics' summation order.  The PLOC tree must cost about as many node tests as the host SAH tree, the LBVH clearly more."""
    sc, o = scenes(name)
    rng = np.random.default_rng(5)
    n = 20000
    eye = np.array(sc.desc.camera.eye.tuple())
    org = eye + rng.normal(size=(n, 3)) * 0.3
    tgt = rng.uniform(-2.5, 2.5, size=(n, 3)) * np.array([1.0, 0.6, 1.0]) + np.array([0, 0.8, 0])
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], axis=1).astype(np.float32)
    res = {}
    try:
        for builder in (0, 1, 2):
            gpu.set_option("bvh_builder", builder)
            gpu.set_option("max_leaf", max_leaf)
            gpu.upload_scene(sc)
            st = gpu.stats()
            if builder:
                nprim = st["triangles"] + st["spheres"] + st["cuboids"]
                assert st["bvh_nodes"] % 2 == 1 and nprim // max_leaf <= st["bvh_nodes"] <= 4 * nprim and 0 < st["bvh_build_ms"] < 50.0   # (split clipping: more references than triangles)
            else:
                assert st["bvh_build_ms"] == 0
            hit, el = gpu.debug_intersect(rays)
            gpu.set_resolution(160, 90)
            gpu.set_option("counters", 1)
            gpu.clear()
            gpu.render(1, 3)
            acc = gpu.read_accumulator().astype(np.float64)
            st = gpu.stats()
            gpu.set_option("counters", 0)
            res[builder] = (hit, el, acc, st["node_tests"] / max(1, st["rays"]), st["bvh_build_ms"])
    finally:
        gpu.set_option("bvh_builder", -1)
        gpu.set_option("max_leaf", 4)
        gpu.set_option("counters", 0)
    h0, e0, a0, nt0, _ = res[0]
    ref, _ = o.render(160, 90, 1, 3, threads=0, counters=True)
    for builder in (1, 2):
        h1, e1, a1, nt1, ms = res[builder]
        assert np.array_equal(h0[:, 0], h1[:, 0])
        hit = h0[:, 0] == 1
        dt = np.abs(h0[hit, 1].astype(np.float64) - h1[hit, 1]) / np.maximum(1.0, h0[hit, 1])
        print("%s builder %d vs host tree: exact t %.5f, max rel dt %.3g, same element %.5f, node tests per ray %.1f vs %.1f, build %.2f ms" %
              (name, builder, (dt == 0).mean(), dt.max(), (e0[hit] == e1[hit]).mean(), nt1, nt0, ms))
        assert dt.max() == 0
        assert (e0[hit] == e1[hit]).mean() > 0.999
        assert np.isfinite(a1).all()
        frac, m1, m0 = _compare(a1, a0)
        assert frac >= GATES[name][0] and abs(m1 - m0) <= 2e-3 * max(1e-3, abs(m0))
        frac, m1, mr = _compare(a1, ref)
        assert frac >= GATES[name][0] and abs(m1 - mr) <= 2e-3 * max(1e-3, abs(mr))
    if name.startswith("rtcamp6"):
        assert res[2][3] <= 1.15 * nt0 and res[2][3] < res[1][3]      # PLOC: within 15 % of the host SAH tree, better than the LBVH


def _heightfield_scene(ha, cells):
    """A terrain of cells x cells x 2 triangles over [-4, 4]^2 as ONE mesh element, borrowing camera / skybox / images from a scene."""
    import ctypes as C
    g = np.linspace(-4.0, 4.0, cells + 1)
    xx, zz = np.meshgrid(g, g, indexing="xy")
    yy = 0.35 * np.sin(1.7 * xx) * np.cos(2.3 * zz) + 0.1 * np.sin(9.0 * xx + 4.0 * zz)
    verts = np.ascontiguousarray(np.stack([xx, yy, zz], axis=-1).reshape(-1, 3), dtype=np.float64)
    i0 = (np.arange(cells)[:, None] * (cells + 1) + np.arange(cells)[None, :]).reshape(-1)
    faces = np.ascontiguousarray(np.concatenate([np.stack([i0, i0 + 1, i0 + cells + 2], axis=1), np.stack([i0, i0 + cells + 2, i0 + cells + 1], axis=1)]), dtype=np.uint64)
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

    class Holder:   # what Renderer.upload_scene wants, plus everything that must stay alive
        pass
    h = Holder()
    h.desc_ptr = C.pointer(d)
    h.keep = (base, el, verts, faces, d)
    return h, verts, faces


def test_device_builders_at_a_million_primitives(gpu, ha):
    """SURVEY.md §8f rank 1 at scale: both device builders take a 10^6-triangle mesh (the multi-workgroup PLOC: nearest neighbour,
    role count, hipCUB scan, merge + compaction per iteration, cluster count kept on the device) — build time from HIP events
    below 20 ms for LBVH and reported for PLOC; closest hits of the two trees are bit-identical and agree with a brute-force
    check of the hit triangle on a sample of rays."""
    sc, verts, faces = _heightfield_scene(ha, 707)    # 707 x 707 x 2 = 999,698 triangles: the largest terrain below 2^20 primitives, i.e. the 20-bit-index / 42-bit-Morton form of the sort key
    rng = np.random.default_rng(2)
    n = 4096
    org = np.stack([rng.uniform(-3.5, 3.5, n), rng.uniform(1.0, 2.0, n), rng.uniform(-3.5, 3.5, n)], axis=1)
    tgt = np.stack([rng.uniform(-3.9, 3.9, n), np.zeros(n), rng.uniform(-3.9, 3.9, n)], axis=1)
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], axis=1).astype(np.float32)
    res = {}
    try:
        for builder in (1, 2):
            gpu.set_option("bvh_builder", builder)
            gpu.upload_scene(sc)
            gpu.upload_scene(sc)                      # second build: allocator and code caches warm
            st = gpu.stats()
            assert st["triangles"] == faces.shape[0]
            res[builder] = (gpu.debug_trace(rays), st["bvh_build_ms"], st["bvh_nodes"])
            print("device builder %d: %d triangles -> %d records in %.2f ms" % (builder, st["triangles"], st["bvh_nodes"], st["bvh_build_ms"]))
    finally:
        gpu.set_option("bvh_builder", -1)
    (h1, e1), ms1, _ = res[1]
    (h2, e2), ms2, _ = res[2]
    assert np.array_equal(h1.view(np.uint32), h2.view(np.uint32)) and np.array_equal(e1, e2)
    assert (h1[:, 0] == 1).mean() > 0.95          # the rays aim at the terrain (a few leave over its edge)
    assert ms1 < 20.0 and ms2 < 20.0, (ms1, ms2)      # measured: 7.6 ms and 8.8 ms
    # the hit point lies on the surface: its height equals the heightfield's (bilinear cell height within the cell's range)
    hit = h1[:, 0] == 1
    pos = h1[hit, 2:5].astype(np.float64)
    yy = 0.35 * np.sin(1.7 * pos[:, 0]) * np.cos(2.3 * pos[:, 2]) + 0.1 * np.sin(9.0 * pos[:, 0] + 4.0 * pos[:, 2])
    assert np.abs(pos[:, 1] - yy).max() < 2e-3     # the mesh samples the function every 0.0113 units: chord error ~1e-4


def test_builders_at_four_million_triangles(gpu, ha, orc):
    """Past the old 2^20-primitive cap (round 4: the leaf word holds 24 index bits and 4 count bits, the device builders' sort key 24 index
    bits + a 36-bit Morton code): a 4,004,450-triangle terrain through the host SAH builder and both device builders.  Closest hits of the
    three trees are bit-identical (walked by the production traversal on the 16-byte records — byte offsets beyond 2^28 — and by the
    scalar walk on the 32-byte ones), and they are the oracle's hits (f64, the reference's own median-split BVH over the same mesh)."""
    sc, verts, faces = _heightfield_scene(ha, 1415)
    assert faces.shape[0] == 4004450 > (1 << 20) * 3
    rng = np.random.default_rng(3)
    n = 4096
    org = np.stack([rng.uniform(-3.5, 3.5, n), rng.uniform(1.0, 2.0, n), rng.uniform(-3.5, 3.5, n)], axis=1)
    tgt = np.stack([rng.uniform(-3.9, 3.9, n), np.zeros(n), rng.uniform(-3.9, 3.9, n)], axis=1)
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], axis=1).astype(np.float32)
    res = {}
    try:
        # the default (bvh_builder = -1) picks the device PLOC build for a scene of this size: the upload — flatten, copy, build — takes about a
        # second where the host's one-thread SAH build alone takes 25 (scenes below 200,000 primitives keep the host tree: checked on rtcamp6)
        import time
        gpu.set_option("bvh_builder", -1)
        t0 = time.perf_counter()
        gpu.upload_scene(sc)
        t_auto = time.perf_counter() - t0
        st = gpu.stats()
        auto_hits = (gpu.debug_trace(rays), gpu.debug_intersect(rays))
        print("4 M triangles, default options: builder %d, hr_upload_scene %.2f s (device build %.1f ms)" % (st["bvh_builder_used"], t_auto, st["bvh_build_ms"]))
        assert st["bvh_builder_used"] == 2 and t_auto < 4.0
        small = ha.Scene("rtcamp6_v3_1")
        gpu.upload_scene(small)
        assert gpu.stats()["bvh_builder_used"] == 0
        for builder in (1, 2, 0):
            gpu.set_option("bvh_builder", builder)
            gpu.upload_scene(sc)
            st = gpu.stats()
            assert st["triangles"] == faces.shape[0] and st["bvh_nodes"] > faces.shape[0] // 4 and st["bvh_builder_used"] == builder
            res[builder] = (gpu.debug_trace(rays), gpu.debug_intersect(rays))
            print("builder %d: %d triangles -> %d records per octant (%.0f MB of 16-byte records), device build %.2f ms" % (
                builder, st["triangles"], st["bvh_nodes"], (st["bvh_nodes"] + 1) * 8 * 16 / 1e6, st["bvh_build_ms"]))
            assert (st["bvh_nodes"] + 1) * 8 * 16 > (1 << 28)          # the quantised records' byte offsets pass the old 2^28 limit
    finally:
        gpu.set_option("bvh_builder", -1)
    (h0, e0), (s0, se0) = res[0]
    res[3] = auto_hits
    for b in (1, 2, 3):
        (h, e), (s_, se) = res[b]
        assert np.array_equal(h.view(np.uint32), h0.view(np.uint32)) and np.array_equal(e, e0), b      # production traversal, 16-byte records
        assert np.array_equal(s_.view(np.uint32), s0.view(np.uint32)) and np.array_equal(se, se0), b    # scalar walk, 32-byte records
    assert np.array_equal(h0[:, :2].view(np.uint32), s0[:, :2].view(np.uint32))                         # the two walks agree on hit / distance
    hit = h0[:, 0] == 1
    assert hit.mean() > 0.95
    o = orc.OracleScene(sc.desc_ptr)
    ref, rel = o.intersect(rays.astype(np.float64))
    assert np.array_equal(ref[:, 0] == 1, hit) and np.array_equal(rel[hit], e0[hit])
    t_err = np.abs(h0[hit, 1].astype(np.float64) - ref[hit, 1]) / np.maximum(1.0, ref[hit, 1])
    n_err = np.abs(h0[hit, 5:8].astype(np.float64) - ref[hit, 5:8]).max(axis=1)
    print("4 M triangles vs oracle: |dt| max %.2e, normal error median %.2e, 99.9 %% %.2e" % (t_err.max(), np.median(n_err), np.quantile(n_err, 0.999)))
    assert t_err.max() <= 2e-5 and np.median(n_err) < 1e-5 and np.quantile(n_err, 0.99) < 2e-3


def test_hundreds_of_launches_in_one_call(gpu, scenes):
    """One hr_render call that issues 400 launches (batch = 1): the event pairs of finished launches are retired on the way without
    stopping the pipeline, the kernel times of all 400 launches end up in the statistics, and the accumulator equals four calls
    of 100 samplings."""
    sc, _ = scenes("cornell_mini")
    gpu.upload_scene(sc)
    gpu.set_resolution(48, 32)
    gpu.set_option("batch", 1)
    try:
        gpu.clear()
        gpu.render(1, 401)
        one = gpu.read_accumulator().astype(np.float64)
        st = gpu.stats()
        assert st["seed_launches"] == 400 and st["trace_launches"] == 400 and st["trace_kernel_ms"] > 0 and st["seed_kernel_ms"] > 0
        gpu.clear()
        for k in range(4):
            gpu.render(1 + 100 * k, 101 + 100 * k)
        four = gpu.read_accumulator().astype(np.float64)
    finally:
        gpu.set_option("batch", 0)
    assert np.abs(one - four).max() <= 1e-5 * max(1.0, np.abs(one).max())


def test_mark_and_wait_keep_the_pipeline_running(gpu, scenes):
    """hr_mark / hr_wait: waiting for an earlier marker must not disturb later work, and the result equals a plain render."""
    sc, _ = scenes("cornell_mini")
    gpu.upload_scene(sc)
    gpu.set_resolution(200, 120)
    gpu.clear()
    gpu.render(1, 13)
    ref = gpu.read_accumulator().astype(np.float64)
    gpu.clear()
    tickets = []
    for s in (1, 5, 9):
        gpu.render(s, s + 4)
        tickets.append(gpu.mark())
        if len(tickets) >= 2:
            gpu.wait(tickets[-2])          # one chunk stays in flight
    gpu.wait(tickets[0])                    # waiting again for an old ticket is a no-op
    gpu.wait(tickets[-1])
    acc = gpu.read_accumulator().astype(np.float64)
    assert np.abs(acc - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_seed_kernels_agree_on_odd_shapes(gpu, scenes):
    """Tiny and ragged images, sampling ranges with strides, one or many groups per workgroup: the producer / consumer seed kernel
    and the fused one must feed the trace kernel the same draws (the same accumulator, bit for bit)."""
    sc, _ = scenes("cornell_mini")
    gpu.upload_scene(sc)
    rng = np.random.default_rng(12)
    shapes = [(1, 1), (3, 2), (5, 17), (64, 1), (257, 3), (100, 100)] + [(int(rng.integers(1, 400)), int(rng.integers(1, 300))) for _ in range(4)]
    try:
        for (w, h) in shapes:
            begin, stride = int(rng.integers(1, 50)), int(rng.integers(1, 4))
            end = begin + stride * int(rng.integers(1, 9))
            gpu.set_resolution(w, h)
            outs = []
            for mode in (0, 1, 2, 3, 4):
                if not _seed_mode_available(gpu, mode):
                    continue
                gpu.set_debug_option("seed_mode", mode)
                gpu.clear()
                gpu.render(begin, end, stride)
                outs.append(gpu.read_accumulator().astype(np.float64))
            for o in outs[1:]:
                assert np.isfinite(o).all()
                assert np.array_equal(outs[0], o), (w, h, begin, end, stride, np.abs(outs[0] - o).max())
    finally:
        gpu.set_debug_option("seed_mode", 2)


def test_ragged_sizes_and_large_sampling_indices_path_by_path(gpu, scenes):
    """Edge cases of the loop bounds against the ORACLE (the test above compares kernels with each other): images smaller than a tile,
    widths and heights that are not multiples of the 4x4 tile, single rows and columns, and sampling indices up to the end of the u32
    range (the index is a seed word, renderer.rs:165-168) — every path of every case takes the oracle's branches or is one of a handful
    of classified divergences, and the same-branch radiances agree as they do at 1920x1080.
    The strips (64x1, 1x64, 257x3) found a real difference when this test was written: normalized_coord is divided by min(w, h)
    (renderer.rs:53-54), so 4 + nc is negative on the far side of an image more than four times as wide as high, and the seed word
    `((4.0 + nc) * k) as usize` saturates to 0 in Rust — the oracle's C cast wrapped and the GPU's clamped, so 4 - 36 % of a strip's
    paths drew other random numbers.  Both now spell the Rust rule out (tests/test_isaac64.py::test_seed_words_saturate_on_strips)."""
    import path_parity
    sc, o = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    total = bad = 0
    worst = 0.0
    for (w, h), sampling in [((1, 1), 1), ((3, 2), 7), ((5, 17), 4294967295), ((2, 5), 65536), ((7, 19), 4000000000), ((33, 17), 123456789), ((130, 67), 2147483648),
                             ((97, 41), 4294967294), ((64, 1), 65536), ((1, 64), 4000000000), ((257, 3), 123456789)]:
        gpu.set_resolution(w, h)
        g = gpu.debug_path_log(sampling)
        a = path_parity.account(g, o.path_log(w, h, sampling))
        assert g[0].shape == (h, w, 4, 3) and np.isfinite(g[0]).all() and a["paths"] == w * h * 4 and a["same_branch"]["rays_equal"]
        if sampling < 4294967295:     # hr_render takes [begin, end): the last u32 index is reachable by the path log only
            gpu.clear()
            gpu.render(sampling, sampling + 1)
            rad = g[0]
            assert np.array_equal(gpu.read_accumulator(), ((rad[:, :, 0] + rad[:, :, 1]) + (rad[:, :, 2] + rad[:, :, 3])).astype(np.float32))
        total += a["paths"]; bad += a["divergent"]
        worst = max(worst, a["same_branch"]["max_rel_floor1"])
    print("ragged sizes: %d paths, %d divergent, worst same-branch error %.3g" % (total, bad, worst))
    assert bad <= 4 and worst <= 2e-3, (total, bad, worst)


def test_cli_multi_device_in_one_process(tmp_path):
    """hanamaru-hip --gpu-ids a,b: device r renders every N-th sampling, hr_allreduce_accumulators sums them for each image (RCCL
    between distinct devices; contexts that share one device — all a single-GPU box can offer — are summed by a kernel on it).
    With two contexts on device 0 the image must equal the single-context image (same samplings, other summation order)."""
    import subprocess
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "hanamaru-renderer_amd", "hanamaru-hip")
    imgs = []
    for name, extra in [("one", []), ("two", ["--gpu-ids", "0,0"])]:
        d = tmp_path / name
        d.mkdir()
        out = subprocess.run([exe, "-w", "160", "-h", "90", "-s", "21", "-t", "1000", "--batch", "4", "--scene", "cornell_mini", "--assets", os.path.join(root, "assets")] + extra,
                             cwd=d, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert "sampled: 21x4 spp." in out.stdout
        if extra:
            assert "devices: 2." in out.stdout
        imgs.append(np.asarray(Image.open(d / "result.png")).astype(int))
    diff = np.abs(imgs[0] - imgs[1])
    assert diff.max() <= 1 and (diff == 0).mean() > 0.999


def test_russian_roulette_is_unbiased_and_off_by_default(gpu, scenes):
    """Option "russian_roulette" (north_star names it; the reference has none, renderer.rs:174-200, so it is off by default and
    every parity test runs without it).  On: paths die with probability 1 - q from the chosen iteration on and survivors are
    re-weighted by 1 / q — fewer rays per path, more noise, the SAME expectation: the image mean over 128 samplings agrees with the
    reference estimator's (GPU without roulette, itself parity-checked) and with the f64 oracle's, block means show no drift."""
    sc, o = scenes("rtcamp6_v3_1")
    gpu.upload_scene(sc)
    w, h, S = 160, 90, 128
    gpu.set_resolution(w, h)

    def render(rr):
        gpu.set_option("russian_roulette", rr)
        gpu.set_option("counters", 1)
        gpu.clear()
        gpu.render(1, S + 1)
        acc = gpu.read_accumulator().astype(np.float64)
        st = gpu.stats()
        gpu.set_option("counters", 0)
        gpu.set_option("russian_roulette", 0)
        return acc, st["rays"] / st["paths"]

    ref, rays_ref = render(0)
    again, _ = render(0)
    assert np.abs(ref - again).max() <= 1e-4 * max(1.0, np.abs(ref).max())      # off means off: only the summation order varies
    for start in (2, 4):
        rr, rays_rr = render(start)
        assert np.isfinite(rr).all() and (rr >= 0).all()
        assert rays_rr < (0.9 if start == 2 else 0.97) * rays_ref, (start, rays_rr, rays_ref)
        assert abs(rr.mean() - ref.mean()) <= 4e-3 * ref.mean(), (start, rr.mean(), ref.mean())
        # 10 x 10-pixel blocks: relative differences scatter around zero (no region is systematically darker or brighter)
        bl = lambda a: a[:h // 10 * 10, :w // 10 * 10].reshape(h // 10, 10, w // 10, 10, 3).sum(axis=(1, 3, 4))
        rel = (bl(rr) - bl(ref)) / bl(ref)
        assert abs(rel.mean()) < 5e-3 and np.median(np.abs(rel)) < 0.03 and np.abs(rel).max() < 0.6, (start, rel.mean(), np.median(np.abs(rel)), np.abs(rel).max())
        print("russian roulette from iteration %d: rays per path %.3f -> %.3f, image mean %.6g vs %.6g, block drift %.2e" % (start, rays_ref, rays_rr, rr.mean(), ref.mean(), rel.mean()))
    # and against the f64 oracle itself, same 128 samplings (the roulette adds variance: at 16 samplings the means differ by 1.5 %)
    orc_acc, _ = o.render(w, h, 1, S + 1, threads=0)
    rr3, _ = render(3)
    assert abs(rr3.mean() - orc_acc.mean()) <= 4e-3 * orc_acc.mean(), (rr3.mean(), orc_acc.mean())


def test_cuboid_far_from_the_origin_stays_finite(gpu, ha, orc):
    """The cuboid face cascade compares the hit position with the face planes at an absolute EPS of 1e-4 (scene.rs:160-182); in
    fp32 a position 2000 units from the origin is off by more than that, no face matches, and a zero normal would put NaNs
    into the accumulator.  The kernel then takes the nearest face: the render of the `simple` scene moved 2000 units away stays
    finite and keeps the f64 oracle's brightness."""
    sc = ha.Scene("simple")
    d = sc.desc
    shift = (2000.0, 0.0, -2000.0)
    for i in range(d.num_elements):
        e = d.elements[i]
        for v in (e.center, e.aabb_min, e.aabb_max):
            v.x += shift[0]; v.y += shift[1]; v.z += shift[2]
    d.camera.eye.x += shift[0]; d.camera.eye.y += shift[1]; d.camera.eye.z += shift[2]
    o = orc.OracleScene(sc.desc_ptr)
    gpu.upload_scene(sc)
    gpu.set_resolution(160, 90)
    gpu.clear()
    gpu.render(1, 5)
    acc = gpu.read_accumulator()
    assert np.isfinite(acc).all() and (acc >= 0).all()
    ref, _ = o.render(160, 90, 1, 5, threads=0)
    assert abs(acc.mean() - ref.mean()) <= 0.002 * ref.mean(), (acc.mean(), ref.mean())
    f2, f3 = _fractions(acc, ref)
    print("simple + 2000 units: within 1e-2 %.4f, within 1e-3 %.4f, mean gpu %.6g oracle %.6g" % (f2, f3, acc.mean(), ref.mean()))
    assert f2 >= 0.99 and f3 >= 0.95, (f2, f3)      # fp32 resolves 1.2e-4 at 2000 units: the ray offsets of 1e-4 are at the rounding limit


def test_rccl_allreduce_of_the_accumulator(gpu, scenes, ha):
    """The multi-GPU exchange of the boundary (hr_comm_* / hr_allreduce_accumulator): RCCL is loaded, a communicator of world
    size 1 is created on this GPU from an ncclUniqueId and ncclAllReduce runs on the accumulator.  With one rank the total must
    equal the rank's own accumulator bit for bit; the total is what hr_read_accumulator / hr_resolve see until the next render,
    and the rank's own accumulator is left untouched (it keeps accumulating)."""
    sc, _ = scenes("cornell_mini")
    gpu.upload_scene(sc)
    gpu.set_resolution(96, 54)
    gpu.clear()
    gpu.render(1, 4)
    own = gpu.read_accumulator()
    uid = ha.comm_unique_id()
    assert len(uid) == ha.COMM_ID_BYTES and any(uid)
    assert gpu.comm_info()["path"] == "none" and gpu.comm_info()["nranks"] == 0
    gpu.comm_init_rank(uid, 1, 0)
    try:
        ci = gpu.comm_info()                   # asked of RCCL: ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion
        assert ci["path"] == "rccl-rank" and ci["nranks"] == 1 and ci["rank"] == 0 and ci["device"] == 0 and ci["rccl_version"] > 20000 and ci["allreduces"] == 0
        assert not gpu.total_device_ptr()
        with pytest.raises(ha.HipError):
            gpu.accumulator_sum(True)          # no total yet
        gpu.allreduce_accumulator()
        assert gpu.comm_info()["allreduces"] == 1
        s_own, s_tot = np.array(gpu.accumulator_sum(False)), np.array(gpu.accumulator_sum(True))
        assert np.array_equal(s_own, s_tot) and np.allclose(s_own, own.astype(np.float64).sum(axis=(0, 1)), rtol=1e-12)
        assert gpu.total_device_ptr() and gpu.total_device_ptr() != gpu.L.hr_accumulator_device_ptr(gpu._h)
        tot = gpu.read_accumulator()
        assert np.array_equal(tot, own) and own.sum() > 0
        img = gpu.resolve(3)
        assert img.std() > 1
        gpu.render(4, 5)                       # a new sampling invalidates the total ...
        assert not gpu.total_device_ptr()
        own2 = gpu.read_accumulator()          # ... and lands on top of the rank's own part
        assert (own2 >= own).all() and own2.sum() > own.sum()
        gpu.allreduce_accumulator()
        assert np.array_equal(gpu.read_accumulator(), own2)
        with pytest.raises(ha.HipError):
            gpu.comm_init_rank(uid, 1, 1)      # rank outside the world
    finally:
        gpu.comm_destroy()
    with pytest.raises(ha.HipError):
        gpu.allreduce_accumulator()            # no communicator any more


def test_one_rccl_per_process(gpu, ha):
    """PyTorch maps its own RCCL build (torch/lib/librccl.so) with libtorch_hip; under torch.distributed.run its NCCL backend runs on it.
    The library must run its collective on THAT object, not load /opt/rocm's librccl beside it: hr_comm_library names what it resolved
    ncclAllReduce from, and afterwards the process still maps exactly one librccl."""
    import torch  # noqa: F401  (what bench.py and the launcher path import first)

    def mapped():
        return sorted({line.split()[-1] for line in open("/proc/self/maps") if "librccl" in line})
    before = mapped()
    path, reused = ha.comm_library()
    after = mapped()
    assert os.path.exists(path) and "librccl" in os.path.basename(path)
    assert len(after) == 1 and os.path.realpath(after[0]) == os.path.realpath(path), (before, after, path)
    if before:      # torch had mapped one: that is the one in use
        assert reused and os.path.realpath(before[0]) == os.path.realpath(path)
    # the launcher path at world size 1 runs on it
    gpu.comm_init_rank(ha.comm_unique_id(), 1, 0)
    try:
        assert gpu.comm_info()["nranks"] == 1 and mapped() == after
    finally:
        gpu.comm_destroy()


def test_rccl_group_path_of_one_process_driving_its_gpus(scenes, ha):
    """The one-process form of the exchange — hr_comm_init_local + hr_allreduce_accumulators, what `python bench.py --gpus N` and the
    CLI's --gpus use on a multi-GPU node — with ONE context: n = 1 is not "all contexts on one device" (that needs two), so the
    call takes the RCCL branch: ncclCommInitAll over the device list, then ncclGroupStart / ncclAllReduce / ncclGroupEnd.  The
    total must equal the context's own accumulator bit for bit; the communicator can be replaced and destroyed."""
    sc, _ = scenes("cornell_mini")
    r = ha.Renderer(0)
    try:
        r.upload_scene(sc)
        r.set_resolution(96, 54)
        r.render(1, 4)
        own = r.read_accumulator()
        ha.comm_init_local([r])
        ci = r.comm_info()
        assert ci["path"] == "rccl-group" and ci["nranks"] == 1 and ci["rank"] == 0 and ci["rccl_version"] > 20000
        assert not r.total_device_ptr()
        ha.allreduce_accumulators([r])
        assert r.total_device_ptr() and r.total_device_ptr() != r.L.hr_accumulator_device_ptr(r._h)
        assert np.array_equal(r.read_accumulator(), own) and own.sum() > 0
        r.render(4, 6)
        assert not r.total_device_ptr()
        own2 = r.read_accumulator()
        ha.allreduce_accumulators([r])                      # the communicator outlives a collective
        assert np.array_equal(r.read_accumulator(), own2) and own2.sum() > own.sum()
        assert r.resolve(5).std() > 1                       # hr_resolve reads the total
        ha.comm_init_local([r])                             # a second init replaces the communicator (ncclCommDestroy + ncclCommInitAll)
        ha.allreduce_accumulators([r])
        assert np.array_equal(r.read_accumulator(), own2)
        r.comm_destroy()
        with pytest.raises(ha.HipError):
            ha.allreduce_accumulators([r])
        # a device listed twice next to another one is refused before RCCL sees it — only checkable with >= 2 devices; here: a null context
        with pytest.raises(ha.HipError):
            ha.comm_init_local([])
    finally:
        r.close()


def test_same_device_group_sum(scenes, ha):
    """hr_comm_init_local over contexts on ONE device + hr_allreduce_accumulators: every context ends up with the same total =
    the sum of the parts, i.e. (sharded by sampling index) the accumulator of one context that rendered all samplings."""
    sc, _ = scenes("cornell_mini")
    rs = [ha.Renderer(0) for _ in range(3)]
    try:
        for r in rs:
            r.upload_scene(sc)
            r.set_resolution(80, 45)
        ha.comm_init_local(rs)
        assert [r.comm_info()["path"] for r in rs] == ["same-device-fallback"] * 3 and [r.comm_info()["rank"] for r in rs] == [0, 1, 2]
        assert rs[0].comm_info()["nranks"] == 3 and rs[0].comm_info()["rccl_version"] == 0
        for k, r in enumerate(rs):
            r.render(1 + k, 10, 3)
        ha.allreduce_accumulators(rs)
        parts = np.sum([r.accumulator_sum(False) for r in rs], axis=0)
        assert (np.abs(parts - np.array(rs[1].accumulator_sum(True))) <= 1e-6 * parts).all()
        tots = [r.read_accumulator().astype(np.float64) for r in rs]
        assert np.array_equal(tots[0], tots[1]) and np.array_equal(tots[0], tots[2])
        one = ha.Renderer(0)
        one.upload_scene(sc)
        one.set_resolution(80, 45)
        one.render(1, 10)
        ref = one.read_accumulator().astype(np.float64)
        one.close()
        assert np.abs(tots[0] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    finally:
        for r in rs:
            r.close()


def test_rtcamp5_gpu_render_against_the_references_committed_image(gpu, scenes):
    """The reference repository ships a render of init_scene_rtcamp5 (rtcamp5.png; golden fixture = its 480x270 downscale).  The
    GPU image of the same scene (textured emitter, TIFF floor, 43 collision-placed diamonds, refraction index 2.42) must show the
    same picture; the floor texture of that older render differs, hence correlation rather than PSNR (unrelated scene: 0.64)."""
    from PIL import Image
    sc, _ = scenes("rtcamp5")
    gpu.upload_scene(sc)
    gpu.set_resolution(960, 540)
    gpu.clear()
    gpu.render(1, 65)
    img = Image.fromarray(gpu.resolve(64)).resize((240, 135), Image.BOX)
    here = os.path.dirname(os.path.abspath(__file__))
    ref = Image.open(os.path.join(here, "golden", "reference_rtcamp5_480x270.png")).resize((240, 135), Image.BOX)
    a, b = np.asarray(img).astype(float), np.asarray(ref).astype(float)
    corr = np.corrcoef(a.ravel(), b.ravel())[0, 1]
    print("rtcamp5 vs reference image: correlation %.4f, mean abs diff %.2f" % (corr, np.abs(a - b).mean()))
    assert corr > 0.9 and np.abs(a - b).mean() < 18.0
