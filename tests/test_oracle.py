"""The CPU oracle against every known answer available for this path.

The reference has no tests and cannot be built here, so the pins are: the BVH shapes and the
path statistics an independent restatement produced during the survey (Appendix C.3 / D.2 — two
implementations agreeing), hand-derived values of the post chain, and the committed golden accumulators."""
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_bvh_shapes_match_survey(scenes):
    _, o = scenes("rtcamp6_v3_1")
    exp = {1: (4095, 2048, 11, {3: 2022, 4: 26}, (-1.332764, 0.003133, -0.540999, 1.062427, 2.494685, 1.340048)),
           2: (7, 4, 2, {3: 4}, (-0.680204, 0.0, -3.730405, 2.680204, 2.7, -2.269595)),
           3: (63, 32, 5, {3: 16, 4: 16}, (-0.871415, 0.0, -3.822478, 2.871415, 3.0, -2.177522)),
           5: (511, 256, 8, {3: 24, 4: 232}, (-0.423932, 0.0, 1.810497, 0.425070, 1.0, 2.585678))}
    for el, (nodes, leaves, depth, hist, aabb) in exp.items():
        st = o.bvh_stats(el)
        assert (st["nodes"], st["leaves"], st["max_depth"]) == (nodes, leaves, depth)
        assert {k: v for k, v in enumerate(st["leaf_hist"]) if v} == hist
        assert np.allclose(st["aabb"], aabb, atol=1e-6)
    for el in range(6, 11):
        st = o.bvh_stats(el)
        assert (st["nodes"], st["leaves"], st["max_depth"]) == (511, 256, 8)
    assert o.bvh_stats(0) is None and o.bvh_stats(4) is None
    top = o.bvh_stats(-1)
    assert (top["nodes"], top["leaves"], top["max_depth"]) == (5, 3, 2)
    assert o.top_leaves() == [[2, 3, 8, 7, 9], [0, 4, 1], [6, 10, 5]]
    assert o.num_emissions() == 1


def test_dodecahedron_bvh(scenes):
    _, o = scenes("rtcamp6_dodeca")
    st = o.bvh_stats(11)
    assert (st["nodes"], st["leaves"], st["max_depth"]) == (4095, 2048, 11)
    assert {k: v for k, v in enumerate(st["leaf_hist"]) if v} == {3: 992, 4: 1056}


def test_path_statistics_match_survey_c1(scenes):
    """BASELINE config 1 (480x270, -s 1): every aggregate of SURVEY.md Appendix D.2, exactly."""
    _, o = scenes("rtcamp6_v3_1")
    _, c = o.render(480, 270, 1, 2, threads=0, counters=True)
    assert c["paths"] == 518400
    assert (c["rays_primary"], c["rays_bounce"], c["rays_shadow"]) == (518400, 576504, 486762)
    assert c["surface_hits"] == 622558 and c["draws"] == 3510822
    assert c["tex_samples"] == 269796 and c["sky_lookups"] == 473863
    assert c["top_node_tests"] == 7888444 and c["mesh_roots"] == 7473052
    assert c["mesh_node_tests"] == 161042086
    assert (c["tri_tests"], c["tri_accepted"]) == (77005150, 1515082)
    assert c["sphere_tests"] == 1571724 and c["cuboid_tests"] == 1571724
    assert c["rays_per_path_hist"][1:19] == [177987, 49532, 192395, 15726, 30107, 7296, 13377, 4016, 7842, 2974, 4283, 1342, 2657, 840, 1898,
                                             1129, 1904, 3095]
    dh = c["draws_per_path_hist"]
    assert [dh[k] for k in range(4, 34, 2)] == [157850, 207019, 74426, 32908, 16354, 9241, 5693, 3998, 8564, 1870, 371, 83, 16, 4, 3]
    # algorithmic bytes of the reference-order traversal: 15,897 B/path
    alg = 32 * (c["top_node_tests"] + c["mesh_node_tests"]) + 36 * c["tri_tests"] + 16 * c["sphere_tests"] + 24 * c["cuboid_tests"]
    assert round(alg / c["paths"]) == 15897


@pytest.mark.parametrize("name,w,h,s,fn", [("rtcamp6_v3_1", 64, 36, 2, "rtcamp6_64x36_s2.npz"), ("cornell_mini", 48, 32, 2, "cornell_mini_48x32_s2.npz"),
                                           ("spheres", 48, 27, 1, "spheres_48x27_s1.npz")])
def test_golden_accumulators(scenes, orc, name, w, h, s, fn):
    g = np.load(os.path.join(GOLD, fn))
    _, o = scenes(name)
    acc, c = o.render(w, h, 1, s + 1, threads=0, counters=True)
    assert np.array_equal(acc.astype(np.float32), g["acc"])            # the oracle is deterministic
    assert np.array_equal(orc.resolve(acc, s), g["rgb8"])
    assert [c[k] for k in orc.COUNTER_FIELDS] == g["counters"].tolist()


def test_render_is_additive_and_thread_independent(scenes):
    _, o = scenes("cornell_mini")
    a, _ = o.render(40, 30, 1, 4, threads=1)
    b, _ = o.render(40, 30, 1, 4, threads=5)
    assert np.array_equal(a, b)
    parts = np.zeros_like(a)
    for r in range(3):
        o.render(40, 30, 1 + r, 4, stride=3, threads=2, acc=parts)
    assert np.allclose(parts, a, rtol=0, atol=1e-12)
    px = o.calc_pixel(40, 30, 7, 9, 1, 0, 2) + 0
    assert np.isfinite(px).all()


def test_reinhard_gamma_known_values(orc):
    # tonemap.rs:22-27 by hand: c = 1.5*hdr, L = .22r + .707g + .071b, out = sat(c (L/900 + 1)/(L + 1)), then ^(1/2.2)
    def ref(hdr):
        c = 1.5 * np.asarray(hdr, dtype=np.float64)
        lum = 0.22 * c[0] + 0.707 * c[1] + 0.071 * c[2]
        return np.clip(c * (lum / 900.0 + 1.0) / (lum + 1.0), 0, 1) ** (1 / 2.2)
    acc = np.array([[[4.0, 4.0, 4.0], [0.0, 0.0, 0.0], [400.0, 0.4, 0.04]]])      # one sampling: scale 1/4
    _, stage = orc.resolve(acc, 1, want_stage=True)
    for i, hdr in enumerate([(1, 1, 1), (0, 0, 0), (100, 0.1, 0.01)]):
        assert np.allclose(stage[0, i], ref(hdr), rtol=1e-14, atol=0)
    lum = 1.5 * (0.22 + 0.707 + 0.071)
    assert stage[0, 0, 0] == pytest.approx((1.5 * (lum / 900 + 1) / (lum + 1)) ** (1 / 2.2), rel=1e-14)


def _bilateral_numpy(img):
    """filter.rs:32-58 with the wrapping-u32 edge rules spelled out."""
    h, w, _ = img.shape
    out = np.zeros_like(img)

    def gauss(x, s):
        return math.exp(-(x * x) / (2 * s * s)) / (2 * math.pi * s * s)
    for y in range(h):
        for x in range(w):
            cs = img[y, x].sum()
            acc, wp = np.zeros(3), 0.0
            for i in range(3):
                for j in range(3):
                    nx = (x - ((1 - i) & 0xffffffff)) & 0xffffffff
                    ny = (y - ((1 - j) & 0xffffffff)) & 0xffffffff
                    nx, ny = min(nx, w - 1), min(ny, h - 1)
                    dx, dy = (x - nx) & 0xffffffff, (y - ny) & 0xffffffff
                    dist = math.sqrt((dx * dx + dy * dy) & 0xffffffff)
                    wgt = gauss((img[ny, nx].sum() - cs) / 3.0, 1.0) * gauss(dist, 16.0)
                    acc += img[ny, nx] * wgt
                    wp += wgt
            out[y, x] = acc / wp
    return out


@pytest.mark.parametrize("w,h", [(9, 6), (1, 1), (2, 7), (5, 1)])
def test_bilateral_edge_rules(orc, w, h):
    rng = np.random.default_rng(w * 31 + h)
    acc = rng.uniform(0, 3, size=(h, w, 3)) * 4
    img, stage = orc.resolve(acc, 1, want_stage=True)
    exp = _bilateral_numpy(stage)
    exp8 = (255.0 * np.clip(exp, 0, 1)).astype(np.uint8)        # color.rs:10-16 truncation
    assert np.array_equal(img, exp8)


def test_texture_bilinear_and_edges(scenes, ha):
    sc, o = scenes("cornell_mini")
    im = sc.image(0).astype(np.float64) / 255.0
    hh, ww = im.shape[:2]

    def texel(x, y):                       # texture.rs:59-63 incl. the u32 wrap of (H - y - 1)
        x = min(x, ww - 1)
        yy = (hh - y - 1) & 0xffffffff
        return im[min(yy, hh - 1), x, :3]

    def ref(u, v):                         # texture.rs:29-49
        x, y = u * ww, v * hh
        x1, y1 = math.floor(x), math.floor(y)
        x2, y2 = x1 + 1.0, y1 + 1.0
        ix1, ix2, iy1, iy2 = max(int(x1), 0), max(int(x2), 0), max(int(y1), 0), max(int(y2), 0)
        g = (texel(ix1, iy1) * (x2 - x) * (y2 - y) + texel(ix2, iy1) * (x - x1) * (y2 - y) + texel(ix1, iy2) * (x2 - x) * (y - y1) +
             texel(ix2, iy2) * (x - x1) * (y - y1))
        return g ** 2.2
    for u, v in [(0.3, 0.7), (0.0, 0.0), (0.999, 0.999), (1.0, 1.0), (0.5, 1.0), (0.013, 0.51)]:
        assert np.allclose(o.image_bilinear(0, u, v), ref(u, v), rtol=1e-13, atol=1e-15)


def test_skybox_face_selection(scenes):
    sc, o = scenes("cornell_mini")
    # face images of cornell_mini are distinguishable by their blue channel: 200 - 20 f (scenes.cpp)
    inten = np.array(sc.desc.skybox.intensity.tuple())
    for f, d in enumerate([(1, .1, .2), (-1, .1, .2), (.1, 1, .2), (.1, -1, .2), (.1, .2, 1), (.1, .2, -1)]):
        rgb = o.skybox(np.array(d, dtype=np.float64))
        assert rgb[2] == pytest.approx(inten[2] * ((200 - 20 * f) / 255.0) ** 2.2, rel=1e-12)
    # ties fall through to the Z faces (strict '>' comparisons, scene.rs:300-318)
    assert o.skybox(np.array([1.0, 1.0, 0.5]))[2] == pytest.approx(inten[2] * ((200 - 80) / 255.0) ** 2.2, rel=1e-12)


def test_closest_hit_basics(scenes):
    sc, o = scenes("cornell_mini")       # element 5 = emissive sphere, centre (0.6, 1.9, 0.4), r 0.25 (scenes.cpp)
    rays = np.array([[0.6, 5, 0.4, 0, -1, 0],      # straight down onto it
                     [0.6, 5, 0.4, 0, 1, 0],       # up: miss
                     [0.6, 1.9, 0.4, 0, 1, 0],     # from INSIDE: spheres are never hit from inside (scene.rs:61-63)
                     [0.0, 3.0, 2.5, 0, -1, 0]],   # onto the floor cuboid (top face y = 0)
                    dtype=np.float64)
    out, el = o.intersect(rays)
    assert out[0, 0] == 1 and el[0] == 5 and out[0, 1] == pytest.approx(5 - 2.15) and np.allclose(out[0, 5:8], (0, 1, 0))
    assert out[1, 0] == 0 and el[1] == -1 and out[1, 1] == 1e100          # Intersection::empty distance (config.rs:9)
    assert out[2, 0] == 0 and el[2] == -1
    assert out[3, 0] == 1 and el[3] == 0 and out[3, 1] == pytest.approx(3.0) and np.allclose(out[3, 5:8], (0, 1, 0))


def _bilinear_numpy(im8, u, v):
    """texture.rs:29-49 + 59-63 on an RGBA8 image array (row 0 = top), gamma-space interpolation, then ^2.2."""
    im = im8.astype(np.float64) / 255.0
    hh, ww = im.shape[:2]

    def texel(x, y):
        x = min(x, ww - 1)
        yy = (hh - y - 1) & 0xffffffff
        return im[min(yy, hh - 1), x, :3]
    x, y = u * ww, v * hh
    x1, y1 = math.floor(x), math.floor(y)
    x2, y2 = x1 + 1.0, y1 + 1.0
    ix1, ix2, iy1, iy2 = max(int(x1), 0), max(int(x2), 0), max(int(y1), 0), max(int(y2), 0)
    g = (texel(ix1, iy1) * (x2 - x) * (y2 - y) + texel(ix2, iy1) * (x - x1) * (y2 - y) + texel(ix1, iy2) * (x2 - x) * (y - y1) +
         texel(ix2, iy2) * (x - x1) * (y - y1))
    return g ** 2.2


# ---- hand-derived pins for the code paths the reference's committed rtcamp6 image does not exercise (material.rs:139-199,
#      scene.rs:67-71, scene.rs:393-395): closed-form values, not oracle-vs-oracle comparisons

def test_refraction_closed_forms(orc):
    """PointMaterial::sample for Refraction (material.rs:154-199) at configurations with closed-form answers."""
    n = np.array([0.0, 1.0, 0.0])
    pos = np.array([0.3, 0.0, -0.2])
    ior = 1.5
    # normal incidence from outside: Fresnel r_s = r_p = ((1/ior - 1) / (1/ior + 1))^2 = 0.04
    fr = ((1 / ior - 1) / (1 / ior + 1)) ** 2
    assert fr == pytest.approx(0.04)
    ok, o, d, refl = orc.material_sample(2, ior, 0.0, 0.039, 0.5, pos, -(-n), n)   # view = direction TOWARDS the eye; sample negates it
    assert ok and np.allclose(d, n, atol=1e-15) and refl == 1.0 and np.allclose(o, pos + 1e-4 * n)          # r0 <= fr: mirror branch
    ok, o, d, refl = orc.material_sample(2, ior, 0.0, 0.041, 0.5, pos, n, n)
    assert ok and np.allclose(d, -n, atol=1e-15) and refl == pytest.approx((1 / ior) ** 2) and np.allclose(o, pos - 1e-4 * n)   # straight through
    # Snell at 45 degrees: sin(t) = sin(45) / 1.5
    view = np.array([math.sin(math.pi / 4), math.cos(math.pi / 4), 0.0])
    ok, o, d, refl = orc.material_sample(2, ior, 0.0, 0.999, 0.5, pos, view, n)
    st = math.sin(math.pi / 4) / ior
    assert np.allclose(d, [-st, -math.sqrt(1 - st * st), 0.0], atol=1e-14) and refl == pytest.approx(1 / ior ** 2)
    cos_i, cos_t = math.cos(math.pi / 4), math.sqrt(1 - st * st)
    rs = ((cos_i / ior - cos_t) / (cos_i / ior + cos_t)) ** 2
    rp = ((cos_t / ior - cos_i) / (cos_t / ior + cos_i)) ** 2
    just_below = 0.5 * (rs + rp) * (1 - 1e-9)
    ok, o, d, refl = orc.material_sample(2, ior, 0.0, just_below, 0.5, pos, view, n)
    assert np.allclose(d, [-view[0], view[1], 0.0], atol=1e-14) and refl == 1.0                           # the Fresnel coin at its edge
    # from inside beyond the critical angle asin(1/1.5) = 41.8 degrees: total internal reflection whatever r0 says
    inside = np.array([math.sin(math.radians(60)), -math.cos(math.radians(60)), 0.0])   # eye side is below the surface
    ok, o, d, refl = orc.material_sample(2, ior, 0.0, 0.999999, 0.5, pos, inside, n)
    assert ok and refl == 1.0 and np.allclose(d, [-inside[0], inside[1], 0.0], atol=1e-14) and np.allclose(o, pos - 1e-4 * n)
    # from inside at normal incidence: leaves with reflectance ior^2
    ok, o, d, refl = orc.material_sample(2, ior, 0.0, 0.5, 0.5, pos, -n, n)
    assert np.allclose(d, n, atol=1e-15) and refl == pytest.approx(ior ** 2) and np.allclose(o, pos + 1e-4 * n)


def test_ggx_family_closed_forms(orc):
    """GGX / GGXRefraction at roughness 0 collapse to the mirror / to plain Refraction (material.rs:113-149, 260-269): the half
    vector is the normal exactly (cos_theta = sqrt((1 - r1) / (1 - r1)) = 1), G = 1, so reflectance = Schlick(v.n)."""
    n = np.array([0.0, 0.0, 1.0])
    pos = np.zeros(3)
    view = np.array([math.sin(0.7), 0.0, math.cos(0.7)])
    for r0, r1 in [(0.1, 0.2), (0.83, 0.6)]:
        ok, o, d, refl = orc.material_sample(3, 0.8, 0.0, r0, r1, pos, view, n)
        assert ok and np.allclose(d, [-view[0], 0.0, view[2]], atol=1e-14)
        assert refl == pytest.approx(0.8 + 0.2 * (1 - math.cos(0.7)) ** 5, rel=1e-13)
        a = orc.material_sample(4, 1.5, 0.0, r0, r1, pos, view, n)
        b = orc.material_sample(2, 1.5, 0.0, r0, r1, pos, view, n)
        assert a[0] == b[0] and np.allclose(a[1], b[1], atol=1e-15) and np.allclose(a[2], b[2], atol=1e-14) and a[3] == pytest.approx(b[3], rel=1e-14)
    # Diffuse: r1 = 0 samples the normal itself, bsdf = 1/pi; GGX bsdf below the horizon is 0
    ok, o, d, refl = orc.material_sample(0, 0.0, 0.5, 0.37, 0.0, pos, view, n)
    assert ok and np.allclose(d, n, atol=1e-15) and refl == 1.0
    assert orc.material_bsdf(0, 0.0, 0.5, view, n, n) == pytest.approx(1 / math.pi)
    assert orc.material_bsdf(3, 0.8, 0.3, view, n, np.array([0.6, 0.0, -0.8])) == 0.0
    # GGX bsdf at the mirror configuration l = reflect: h = n, D = 1 / (pi alpha^2), G from Smith with l.n = v.n
    alpha2 = 0.3 ** 2                                    # roughness_to_alpha2: alpha = roughness, alpha2 = roughness^2 (material.rs:250-258)
    l = np.array([-view[0], 0.0, view[2]])
    vn = view[2]
    lam = 0.5 * math.sqrt(1 + alpha2 * (1 / vn ** 2 - 1)) - 0.5
    expect = (1 / (math.pi * alpha2)) * (1 / (1 + 2 * lam)) * (0.8 + 0.2 * (1 - vn) ** 5) / (4 * vn * vn)
    assert orc.material_bsdf(3, 0.8, 0.3, view, n, l) == pytest.approx(expect, rel=1e-12)


def test_sphere_uv_poles_and_seam(scenes):
    """Sphere::intersect's lat-long UVs (scene.rs:67-71): v = 1 - acos(n.y)/pi, u = 0.5 - sign(n.z) acos(n.x / |n.xz|) / 2pi."""
    sc, o = scenes("material_examples")          # sphere 2: centre (0, 0.4, 0), r 0.4 (main.rs material_examples)
    c, r = np.array([0.0, 0.4, 0.0]), 0.4
    cases = [((1, 0, 0), 0.5, 0.5),            # +x: acos(1) = 0; f64::signum(+0.0) = 1
             ((0, 0, 1), 0.25, 0.5), ((0, 0, -1), 0.75, 0.5),
             ((-1, 0, 1e-9), 0.0, 0.5), ((-1, 0, -1e-9), 1.0, 0.5),   # the seam at -x: u jumps from 0 to 1
             ((1, 1, 0), 0.5, 0.75), ((0, 1, 1e-6), 0.25, 1.0)]       # towards the north pole v -> 1
    for dirn, u, v in cases:
        nrm = np.array(dirn, dtype=np.float64)
        nrm /= np.linalg.norm(nrm)
        hit = o.intersect_material(c + 0.49 * nrm, -nrm)   # from inside the 0.2 gap to the neighbouring spheres
        assert hit["hit"] and hit["element"] == 2 and hit["distance"] == pytest.approx(0.49 - r, rel=1e-10)
        assert np.allclose(hit["normal"], nrm, atol=1e-12)
        assert hit["u"] == pytest.approx(u, abs=2e-7) and hit["v"] == pytest.approx(v, abs=2e-6), (dirn, hit["u"], hit["v"])
    # exactly at the pole n.xz = 0: 0/0 -> NaN u, as in the reference (the constant-colour materials of these spheres ignore it)
    hit = o.intersect_material(c + np.array([0, 3.0, 0]), np.array([0, -1.0, 0]))
    assert hit["hit"] and hit["v"] == 1.0 and math.isnan(hit["u"])
    assert hit["surface"] == 1 and np.array_equal(hit["albedo"], [1, 1, 1])


def test_textured_emitter_and_image_roughness(scenes):
    """scene.rs:389-396: emission / roughness come from texture.sample(u, v) * tint.  rtcamp5's sphere 3 is an EMISSIVE sphere
    textured with the earth image (tint (5,5,2)) and sphere 4 takes its roughness from the same image's red channel
    (main.rs:252-500); the expected values are computed here from the decoded texels, texture.rs:29-49 restated in numpy."""
    sc, o = scenes("rtcamp5")
    e3, e4 = sc.desc.elements[3], sc.desc.elements[4]
    assert e3.material.emission.image == 0 and e3.material.emission.color.tuple() == (5.0, 5.0, 2.0) and e4.material.roughness.image == 0
    im = sc.image(0)
    brightest = 0.0
    for el, e in ((3, e3), (4, e4)):
        c, r = np.array(e.center.tuple()), e.radius
        for dirn in [(1, 0, 0), (0, 0, 1), (0.3, 0.5, -0.8), (-0.7, -0.1, 0.2)]:
            nrm = np.array(dirn, dtype=np.float64)
            nrm /= np.linalg.norm(nrm)
            hit = o.intersect_material(c + (r + 0.02) * nrm, -nrm)    # from just outside: other objects stand close by
            assert hit["hit"] and hit["element"] == el
            v = 1.0 - math.acos(nrm[1]) / math.pi
            u = 0.5 - math.copysign(1.0, nrm[2]) * math.acos(nrm[0] / math.hypot(nrm[0], nrm[2])) / (2 * math.pi)
            assert hit["u"] == pytest.approx(u, abs=1e-9) and hit["v"] == pytest.approx(v, abs=1e-9)
            tex = _bilinear_numpy(im, hit["u"], hit["v"])
            if el == 3:
                assert np.allclose(hit["emission"], tex * np.array([5.0, 5.0, 2.0]), rtol=1e-12, atol=0)
                brightest = max(brightest, float(hit["emission"].max()))
            else:
                assert hit["roughness"] == pytest.approx(tex[0] * 1.0, rel=1e-12) and np.array_equal(hit["emission"], [0, 0, 0])
    assert brightest > 0.5     # the inverted earth map is black over the oceans: at least one probe landed on a lit texel
    # tbf3: four emitters share the image with different tints — a NEE-visible textured emitter list
    sc2, o2 = scenes("tbf3")
    assert o2.num_emissions() == 4
    for el in (3, 4, 5, 6):
        e = sc2.desc.elements[el]
        c = np.array(e.center.tuple())
        hit = o2.intersect_material(c + np.array([0.0, 0.0, e.radius + 0.02]), np.array([0.0, 0.0, -1.0]))
        assert hit["hit"] and hit["element"] == el and hit["u"] == pytest.approx(0.25) and hit["v"] == pytest.approx(0.5)
        assert np.allclose(hit["emission"], _bilinear_numpy(sc2.image(0), 0.25, 0.5) * np.array(e.material.emission.color.tuple()), rtol=1e-12)


@pytest.mark.parametrize("x0,y0", [(200, 100), (300, 930), (936, 562), (1300, 420),
                                   # round 4: every material of the scene — green armadillo (GGX), glass armadillo (Refraction), blue and magenta
                                   # armadillos, the bunny's wire (GGX, lit from inside), the frame, the mirror's picture of the lamps, the floor's lettering
                                   (420, 660), (820, 840), (1440, 700), (1180, 500), (900, 460), (1540, 300), (1200, 230), (760, 1000)])
def test_oracle_reproduces_the_reference_binarys_committed_render(scenes, orc, x0, y0):
    """tests/golden/reference_rtcamp6_1000x4spp.png is the output of the REAL reference binary (committed in its repository,
    README.md:19: default scene, 1920x1080, -s 1000).  The oracle renders small crops of that exact configuration (seeds
    depend on full-image coordinates) and must reproduce the 8-bit image: this pins, against the Rust program itself, the
    ISAAC-64 seeding and its u64->f64 conversion, the scene construction, the PNG/JPEG texture decoding (sky and floor
    crops), the estimator and the whole post chain.  (Interior of the crop only: the 3x3 bilateral needs neighbours.)"""
    from PIL import Image
    ref = np.asarray(Image.open(os.path.join(GOLD, "reference_rtcamp6_1000x4spp.png")).convert("RGB")).astype(int)
    _, o = scenes("rtcamp6_v3_1")
    rw, rh = 12, 8
    acc = o.render_region(1920, 1080, x0, y0, rw, rh, 1, 1001, threads=0)
    img = orc.resolve(acc, 1000).astype(int)
    d = np.abs(img[1:-1, 1:-1] - ref[y0 + 1:y0 + rh - 1, x0 + 1:x0 + rw - 1])
    assert d.max() <= 1 and (d == 0).mean() >= 0.98, (d.max(), (d == 0).mean())


def test_oracle_whole_frame_pin(scenes, orc):
    """Pin 1b (round 6): the oracle against the reference binary's committed render over the WHOLE frame.  tools/oracle_whole_frame_pin.py
    rendered all 1920 x 1080 pixels x 1,000 samplings with the oracle once, offline (hours of CPU), ran its post chain and recorded, as data:
    every channel where oracle - reference != 0, per-row counts, and the f64 accumulators of six row triples.  Here: (i) the record is
    consistent and says what DESIGN.md §6.1 quotes (not one of the 6,220,800 channels differs); (ii) segments of the recorded
    accumulator rows are re-derived with the oracle as it is built now — exact equality: the record IS this oracle's output; (iii) the middle
    row of every triple, resolved here, reproduces reference + recorded difference on all 1,920 pixels — accumulator, post chain and the
    difference list hang together."""
    from PIL import Image
    path = os.path.join(GOLD, "oracle_whole_frame_pin.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/oracle_whole_frame_pin.npz not generated yet (tools/oracle_whole_frame_pin.py, hours of CPU)")
    z = np.load(path)
    ref = np.asarray(Image.open(os.path.join(GOLD, "reference_rtcamp6_1000x4spp.png")).convert("RGB")).astype(np.int16)
    H, W = ref.shape[:2]
    assert (H, W) == (1080, 1920) and int(z["meta"][0]) == 1000
    diff = np.zeros_like(ref)
    yxc = z["diff_yxc"].astype(int)
    diff[yxc[:, 0], yxc[:, 1], yxc[:, 2]] = z["diff_val"]
    # (i)
    assert np.array_equal((diff == 0).reshape(H, -1).sum(axis=1), z["rows_exact"]) and np.array_equal((np.abs(diff) <= 1).reshape(H, -1).sum(axis=1), z["rows_within1"])
    identical, within1, worst = float((diff == 0).mean()), float((np.abs(diff) <= 1).mean()), int(np.abs(diff).max())
    print("oracle vs the reference binary's render, whole frame: %.4f %% identical, %.4f %% within 1 LSB, worst %d LSB, %d differing channels" % (100 * identical, 100 * within1, worst, len(z["diff_val"])))
    assert len(z["diff_val"]) == 0 and identical == 1.0 and worst == 0        # measured: byte-identical over the whole frame
    oracle_img = ref + diff
    # (ii) + (iii)
    _, o = scenes("rtcamp6_v3_1")
    rows, idx = z["acc_rows"], z["acc_row_index"].astype(int)
    rng = np.random.RandomState(6)
    for t in range(0, len(idx), 3):
        assert idx[t] + 1 == idx[t + 1] and idx[t + 1] + 1 == idx[t + 2]
        for k in (0, 1, 2):      # a 16-pixel segment of every recorded row, at a random column
            x0 = int(rng.randint(0, W - 16))
            seg = o.render_region(W, H, x0, int(idx[t + k]), 16, 1, 1, 1001, threads=0)
            assert np.array_equal(seg[0], rows[t + k, x0:x0 + 16]), (idx[t + k], x0)
        img = orc.resolve(np.ascontiguousarray(rows[t:t + 3]), 1000).astype(np.int16)
        assert np.array_equal(img[1], oracle_img[idx[t + 1]]), idx[t + 1]
