"""CPU tier: host/scenes.cpp against the numbers of the reference's scene builders.

tests/golden/scene_literals.json holds what tools/extract_scene_literals.py read out of the reference's `init_scene_*` functions (numbers, enum
tags, asset names, matrix factors, loop bounds, gen_range ranges and — for values only known at run time — expression trees over the loop's
draws).  This file REPLAYS that data into a scene in Python — Camera::new (camera.rs:45-64), Matrix44 (matrix.rs), ObjLoader (loader.rs:12-59),
hsv_to_rgb (color.rs:51-61), ISAAC-64 gen_range draws and the AABB-collision rejection of Scene::add_with_check_collisions
(scene.rs:366-376, bvh.rs:14-18) — and compares the result, element by element and field by field, with the hr_scene_desc the host library
builds.  GPU and oracle are fed that same hr_scene_desc, so without this a wrong literal in host/scenes.cpp would be invisible to every
parity test.  The ISAAC-64 outputs come from the oracle library (itself pinned by rand's known-answer vectors, tests/test_isaac64.py)."""
import copy
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "scene_literals.json")))
SURFACE = {"Diffuse": 0, "Specular": 1, "Refraction": 2, "GGX": 3, "GGXRefraction": 4}
# scene name of the host library -> function of the reference
SCENES = {"simple": "simple", "material_examples": "material_examples", "rtcamp5": "rtcamp5", "tbf3": "tbf3", "rtcamp6_v1": "rtcamp6_v1",
          "rtcamp6_v2": "rtcamp6_v2", "rtcamp6_v3": "rtcamp6_v3", "rtcamp6_v3_1": "rtcamp6_v3_1"}


# ---- the reference's arithmetic, restated (each operation in the reference's order, so that results are equal to the last bit)
def mat_mul(a, b):      # matrix.rs:162-176
    return [[a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j] + a[i][3] * b[3][j] for j in range(4)] for i in range(4)]


def mat_factor(op, a):  # matrix.rs:11-78
    if op == "scale_linear":
        op, a = "scale", [a[0]] * 3
    if op == "scale":
        return [[a[0], 0, 0, 0], [0, a[1], 0, 0], [0, 0, a[2], 0], [0, 0, 0, 1.0]]
    if op == "translate":
        return [[1.0, 0, 0, a[0]], [0, 1.0, 0, a[1]], [0, 0, 1.0, a[2]], [0, 0, 0, 1.0]]
    s, c = math.sin(a[0]), math.cos(a[0])
    if op == "rotate_x":
        return [[1.0, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1.0]]
    if op == "rotate_y":
        return [[c, 0, s, 0], [0, 1.0, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1.0]]
    raise ValueError(op)


def load_obj(path, m):  # loader.rs:12-59 + matrix.rs:178-189
    v, f = [], []
    for line in open(path):
        sp = line.rstrip("\n").rstrip("\r").split(" ")
        if sp[0] == "v":
            x, y, z = float(sp[1]), float(sp[2]), float(sp[3])
            v.append([x * m[r][0] + y * m[r][1] + z * m[r][2] + m[r][3] for r in range(3)])
        elif sp[0] == "f":
            idx = [int(t.split("/")[0]) - 1 for t in sp[1:] if t != ""]
            f.append(idx[:3])
            if len(sp) == 5:
                f.append([idx[0], idx[2], idx[3]])
    return np.array(v, dtype=np.float64), np.array(f, dtype=np.int64)


def hsv_to_rgb(h, s, v):  # color.rs:51-61
    sat = lambda x: min(max(x, 0.0), 1.0)
    hue = [sat(abs(h * 6.0 - 3.0) - 1.0), sat(2.0 - abs(h * 6.0 - 2.0)), sat(2.0 - abs(h * 6.0 - 4.0))]
    return [((c - 1.0) * s + 1.0) * v for c in hue]


def camera_new(c):        # camera.rs:45-64
    eye, tgt, up = np.array(c["eye"]), np.array(c["target"]), np.array(c["up"])
    up = up / math.sqrt(float(up @ up))                      # `.normalize()` on the literal
    norm = lambda a: a / math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])
    cross = lambda a, b: np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])
    fw = norm(tgt - eye)
    right = norm(cross(fw, up))
    upv = norm(cross(right, fw))
    hh = math.tan(c["fov"] * (math.pi / 180.0))
    return {"eye": eye, "forward": fw, "right": right, "up": upv, "plane_half_right": right * hh * c["focus"], "plane_half_up": upv * hh * c["focus"],
            "lens_radius": 0.5 * c["aperture"], "focus_distance": c["focus"], "lens_shape": {"Square": 0, "Circle": 1}[c["lens"]]}


def u64_to_f64(v):
    """rand 0.4.3's `Rand for f64`: the low 52 bits as the mantissa of a float in [1, 2), minus 1"""
    return np.array([0x3FF0000000000000 | (int(v) & 0x000FFFFFFFFFFFFF)], dtype=np.uint64).view(np.float64)[0] - 1.0


class Draws:
    """StdRng::from_seed(seed) (ISAAC-64) + gen_range(lo, hi) of rand 0.3 / 0.4: lo + (hi - lo) * next_f64"""
    def __init__(self, orc, seed, n=20000):
        s = (C.c_uint64 * len(seed))(*seed)
        self.raw = np.zeros(n, dtype=np.uint64)
        assert orc.lib().orc_isaac64(s, len(seed), 0, self.raw.ctypes.data, n) == 0
        assert all(orc.lib().orc_u64_to_f64(int(r)) == u64_to_f64(r) for r in self.raw[:64])     # the conversion the oracle's path generators use
        self.k = 0

    def gen_range(self, lo, hi):
        v = lo + (hi - lo) * float(u64_to_f64(self.raw[self.k]))
        self.k += 1
        return v


def ev(t, d, count):
    """a number, or an expression tree over this attempt's draws `d` and the loop counter"""
    if not isinstance(t, list):
        return t
    op = t[0]
    if op == "draw":
        return d[t[1]]
    if op == "count":
        return float(count)
    if op == "neg":
        return -ev(t[1], d, count)
    if op == "to_radians":
        return ev(t[1], d, count) * (math.pi / 180.0)
    a, b = ev(t[1], d, count), ev(t[2], d, count)
    return a + b if op == "+" else a - b if op == "-" else a * b if op == "*" else a / b


def tex_value(t, d, count):
    """(colour, image path or None)"""
    if t["t"] == "white":
        return [1.0, 1.0, 1.0], None
    if t["t"] == "black":
        return [0.0, 0.0, 0.0], None
    if t["t"] == "color":
        return [ev(x, d, count) for x in t["color"]], None
    if t["t"] == "hsv":
        rgb = hsv_to_rgb(*[ev(x, d, count) for x in t["hsv"]])
        return [c * t["scale"] for c in rgb] if "scale" in t else rgb, None
    return [ev(x, d, count) for x in t.get("color", [1.0, 1.0, 1.0])], t["image"]


def realise(e, assets, d=(), count=0):
    """fixture element -> concrete element (numbers only), with its AABB (scene.rs:82-87, 187, bvh.rs:91-99)"""
    out = {"kind": e["kind"], "material": {"surface": SURFACE[e["material"]["surface"]], "param": e["material"]["param"]}}
    for k in ("albedo", "emission", "roughness"):
        out["material"][k] = tex_value(e["material"][k], d, count)
    if e["kind"] == "sphere":
        out["center"] = [ev(x, d, count) for x in e["center"]]
        out["radius"] = ev(e["radius"], d, count)
        out["aabb"] = ([c - out["radius"] for c in out["center"]], [c + out["radius"] for c in out["center"]])
    elif e["kind"] == "cuboid":
        out["min"], out["max"] = [ev(x, d, count) for x in e["min"]], [ev(x, d, count) for x in e["max"]]
        out["aabb"] = (out["min"], out["max"])
    else:
        m = None
        for f in e["matrix"]:
            fm = mat_factor(f[0], [ev(x, d, count) for x in f[1:]])
            m = fm if m is None else mat_mul(m, fm)
        out["verts"], out["faces"] = load_obj(os.path.join(assets, e["model"]), m)
        used = out["verts"][np.unique(out["faces"])]
        out["aabb"] = (used.min(axis=0).tolist(), used.max(axis=0).tolist())
    return out


def collides(a, b):   # bvh.rs:14-18, strict
    return all(a[0][k] < b[1][k] and a[1][k] > b[0][k] for k in range(3))


def replay(fx, assets, orc):
    """the element list Scene.elements ends up with (main.rs: the vec! literal, then placement loops and scene.add calls in source order)"""
    elems = []
    rng = Draws(orc, fx["seed"]) if fx["seed"] else None
    attempts = []
    for step in fx["order"]:
        if step == "fixed":
            elems += [realise(e, assets) for e in fx["fixed"]]
        elif step.startswith("add:"):
            elems.append(realise(fx["added"][int(step[4:])], assets))
        else:
            lp = fx["loops"][int(step[5:])]
            count, tries = 0, 0
            while count < lp["count"]:
                d = [rng.gen_range(lo, hi) for lo, hi in lp["draws"]]      # every attempt consumes all its draws, in evaluation order
                cand = realise(lp["element"], assets, d, count)
                tries += 1
                assert tries < 100000
                if not any(collides(e["aabb"], cand["aabb"]) for e in elems):
                    elems.append(cand)
                    count += 1
            attempts.append(tries)
    return elems, attempts


def v3(v):
    return [v.x, v.y, v.z]


def compare(fx, desc, ha, assets, orc):
    """raises AssertionError at the first field of the host library's scene that differs from the replayed reference data"""
    cam = camera_new(fx["camera"])
    for k in ("eye", "forward", "right", "up", "plane_half_right", "plane_half_up"):
        np.testing.assert_allclose(v3(getattr(desc.camera, k)), cam[k], rtol=0, atol=1e-15, err_msg="camera." + k)
    assert desc.camera.lens_radius == cam["lens_radius"] and desc.camera.focus_distance == cam["focus_distance"] and desc.camera.lens_shape == cam["lens_shape"]
    assert v3(desc.skybox.intensity) == fx["skybox"]["intensity"]
    for k, face in enumerate(fx["skybox"]["faces"]):
        assert np.array_equal(_image(desc, desc.skybox.face_image[k]), ha.decode_image(os.path.join(assets, fx["skybox"]["dir"], face))), "skybox face %d" % k
    elems, attempts = replay(fx, assets, orc)
    assert desc.num_elements == len(elems), (desc.num_elements, len(elems))
    compare_elements(elems, desc, ha, assets)
    return elems, attempts


def compare_elements(elems, desc, ha, assets):
    for i, want in enumerate(elems):
        e = desc.elements[i]
        where = "element %d (%s)" % (i, want["kind"])
        assert e.kind == {"sphere": 0, "cuboid": 1, "mesh": 2}[want["kind"]], where
        if want["kind"] == "sphere":
            assert v3(e.center) == want["center"] and e.radius == want["radius"], (where, v3(e.center), want["center"], e.radius, want["radius"])
        elif want["kind"] == "cuboid":
            assert v3(e.aabb_min) == want["min"] and v3(e.aabb_max) == want["max"], where
        else:
            got = np.ctypeslib.as_array(C.cast(e.vertexes, C.POINTER(C.c_double)), shape=(e.num_vertexes, 3))
            faces = np.ctypeslib.as_array(e.faces, shape=(e.num_faces, 3))
            assert got.shape == want["verts"].shape and np.array_equal(faces.astype(np.int64), want["faces"]), where
            np.testing.assert_allclose(got, want["verts"], rtol=0, atol=1e-13, err_msg=where)
        m = e.material
        assert m.surface == want["material"]["surface"], where
        if want["material"]["param"] is not None:
            assert m.param == want["material"]["param"], (where, m.param)
        for k in ("albedo", "emission", "roughness"):
            t = getattr(m, k)
            col, img = want["material"][k]
            np.testing.assert_allclose(v3(t.color), col, rtol=0, atol=1e-15, err_msg="%s %s" % (where, k))
            assert (t.image >= 0) == (img is not None), (where, k)
            if img is not None:
                assert np.array_equal(_image(desc, t.image), ha.decode_image(os.path.join(assets, img))), (where, k, img)


def _image(desc, i):
    im = desc.images[i]
    return np.ctypeslib.as_array(im.rgba, shape=(im.height, im.width, 4))


@pytest.mark.parametrize("name", sorted(SCENES))
def test_scene_builder_matches_the_reference_literals(ha, orc, name):
    sc = ha.Scene(name)
    elems, attempts = compare(FIX[SCENES[name]], sc.desc, ha, ha.ASSET_ROOT, orc)
    print(name, len(elems), "elements; attempts per placement loop:", attempts)
    if name == "rtcamp6_v2":    # the 100 + 5 sphere generator; rejections happen, so the draw order matters
        assert len(elems) == 106 and attempts[0] > 100


def test_spheres_scene_is_the_rtcamp6_v2_generator_with_two_materials(ha, orc):
    """BASELINE config 2 is build-defined: camera, skybox, the 100 + 5 sphere generator and its five emitters are rtcamp6_v2's (same draws, same
    placements); GGX is replaced by Diffuse / Specular alternating with the accepted-sphere index, and the dodecahedron is left out."""
    fx = copy.deepcopy(FIX["rtcamp6_v2"])
    fx["order"] = [s for s in fx["order"] if not s.startswith("add:")]
    elems, _ = replay(fx, ha.ASSET_ROOT, orc)
    sc = ha.Scene("spheres")
    assert sc.desc.num_elements == len(elems) == 105
    for i, want in enumerate(elems):
        e = sc.desc.elements[i]
        assert e.kind == 0 and v3(e.center) == want["center"] and e.radius == want["radius"], i
        if i < 100:
            assert e.material.surface == (0 if i % 2 == 0 else 1) and v3(e.material.albedo.color) == pytest.approx(want["material"]["albedo"][0], abs=1e-15)
        else:
            assert e.material.surface == 0 and v3(e.material.emission.color) == pytest.approx(want["material"]["emission"][0], abs=1e-15)


def test_config5_scene_is_the_headline_scene_plus_the_dodecahedron(ha, orc):
    """BASELINE config 5 is build-defined: rtcamp6_v3_1 — camera, skybox, all eleven elements as the reference's literals
    give them — plus models/fractal_dodecahedron.obj with the Refraction-1.5 material of main.rs:910-915."""
    fx = FIX["rtcamp6_v3_1"]
    elems, _ = replay(fx, ha.ASSET_ROOT, orc)
    sc = ha.Scene("rtcamp6_dodeca")
    assert sc.desc.num_elements == len(elems) + 1 == 12
    compare_elements(elems, sc.desc, ha, ha.ASSET_ROOT)
    extra = sc.desc.elements[11]
    assert extra.kind == 2 and extra.num_faces == 7200 and extra.material.surface == 2 and extra.material.param == 1.5
    cam = camera_new(fx["camera"])
    np.testing.assert_allclose(v3(sc.desc.camera.eye), cam["eye"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("path,delta", [(("simple", "fixed", 1, "radius"), 1e-9), (("rtcamp6_v3_1", "added", 3, "matrix", 0, 1), 1e-9), (("rtcamp6_v3_1", "camera", "eye", 0), 1e-9), (("simple", "camera", "fov"), 1e-7), (("material_examples", "fixed", 3, "material", "param"), 1e-6),
                                        (("tbf3", "skybox", "intensity", 2), 1e-6), (("rtcamp6_v2", "loops", 0, "draws", 1, 0), 1e-6),
                                        (("rtcamp5", "fixed", 0, "matrix", 1, 1), 1e-7), (("rtcamp6_v1", "fixed", 0, "material", "emission", "color", 0), 1e-6)])
def test_a_perturbed_literal_is_caught(ha, orc, path, delta):
    """the point of the fixture: change ONE number of it by a hair and the comparison with host/scenes.cpp fails"""
    fx = copy.deepcopy(FIX[path[0]])
    node = fx
    for k in path[1:-1]:
        node = node[k]
    assert isinstance(node[path[-1]], float), (path, node[path[-1]])
    node[path[-1]] += delta
    sc = ha.Scene(path[0])
    with pytest.raises(AssertionError):
        compare(fx, sc.desc, ha, ha.ASSET_ROOT, orc)


@pytest.mark.skipif(not os.path.exists("/root/reference/src/main.rs"), reason="the reference is only present in the build container")
def test_fixture_is_what_the_extractor_reads_from_the_reference():
    """provenance of tests/golden/scene_literals.json: re-running tools/extract_scene_literals.py on the reference's main.rs gives the committed
    data, number for number (build container only — the GPU box has no /root/reference)"""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "extract_scene_literals.py"), "/root/reference/src/main.rs"], stdout=subprocess.PIPE, text=True, check=True).stdout
    fresh = json.loads(out)
    for name, data in FIX.items():
        if name.startswith("_"):
            continue
        assert fresh[name] == data, name
    assert sorted(k for k, v in fresh.items() if isinstance(v, dict) and "error" in v) == sorted(FIX["_not_extracted"])
