"""Host layer (loader.rs / camera.rs / matrix.rs / scene builders / image codecs restated in C++)."""
import math
import os

import numpy as np
import pytest

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def test_png_decode_is_exact(ha):
    from PIL import Image
    p = os.path.join(ASSETS, "textures/2d/magic-circle3.png")
    assert np.array_equal(ha.decode_image(p), np.asarray(Image.open(p)))


def test_tiff_lzw_decode_is_exact(ha, tmp_path):
    """LZW + horizontal-predictor RGB strips (the reference's MarbleFloorTiles2 floor texture) and an uncompressed RGBA file."""
    from PIL import Image
    p = os.path.join(ASSETS, "textures/2d/MarbleFloorTiles2/TexturesCom_MarbleFloorTiles2_1024_c_diffuse.tiff")
    assert np.array_equal(ha.decode_image(p), np.asarray(Image.open(p).convert("RGBA")))
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(19, 31, 4), dtype=np.uint8)
    for name, kw in [("raw.tiff", {}), ("lzw.tiff", {"compression": "tiff_lzw"})]:
        q = str(tmp_path / name)
        Image.fromarray(img, "RGBA").save(q, **kw)
        assert np.array_equal(ha.decode_image(q), img), name
    grey = rng.integers(0, 256, size=(8, 40), dtype=np.uint8)
    q = str(tmp_path / "grey.tiff")
    Image.fromarray(grey, "L").save(q, compression="tiff_lzw")
    got = ha.decode_image(q)
    assert np.array_equal(got[:, :, 0], grey) and np.array_equal(got[:, :, 2], grey) and (got[:, :, 3] == 255).all()


def test_png_write_round_trip(ha, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    path = str(tmp_path / "x.png")
    ha.write_png(path, img)
    assert np.array_equal(np.asarray(Image.open(path)), img)           # an independent decoder reads it back
    assert np.array_equal(ha.decode_image(path)[:, :, :3], img)       # and so does ours
    assert (ha.decode_image(path)[:, :, 3] == 255).all()


@pytest.mark.parametrize("face", ["posx", "negx", "posy", "negy", "posz", "negz"])
def test_jpeg_decode_close_to_libjpeg(ha, face):
    """Baseline 4:2:0 JPEG.  The reference's decoder crate (jpeg-decoder 0.1.15) is not available; the decoder
    is pinned against Pillow/libjpeg-turbo instead: different IDCT/upsampling roundings -> a few LSB."""
    from PIL import Image
    p = os.path.join(ASSETS, "textures/cube/Powerlines/%s.jpg" % face)
    mine = ha.decode_image(p)
    ref = np.asarray(Image.open(p).convert("RGB")).astype(int)
    assert mine.shape == (1024, 1024, 4) and (mine[:, :, 3] == 255).all()
    d = np.abs(mine[:, :, :3].astype(int) - ref)
    assert d.max() <= 4 and d.mean() < 0.1 and (d > 2).mean() < 1e-4


def test_decode_errors(ha, tmp_path):
    with pytest.raises(ha.HostError):
        ha.decode_image(str(tmp_path / "missing.png"))
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not an image at all")
    with pytest.raises(ha.HostError):
        ha.decode_image(str(bad))


def _parse_obj(path):
    v, f = [], []
    for line in open(path):
        t = line.rstrip("\r\n").split(" ")
        if t[0] == "v":
            v.append([float(t[1]), float(t[2]), float(t[3])])
        elif t[0] == "f":
            idx = [int(c.split("/")[0]) - 1 for c in t[1:]]
            f.append(idx[:3])
            if len(t) == 5:
                f.append([idx[0], idx[2], idx[3]])
    return np.array(v), np.array(f, dtype=np.uint64)


@pytest.mark.parametrize("rel,nv,nf", [("models/box.obj", 8, 12), ("models/picture_frame.obj", 56, 112), ("models/armadilo_1000.obj", 502, 1000),
                                       ("models/bunny/bunny_wired_300.obj", 1910, 6170), ("models/fractal_dodecahedron.obj", 3200, 7200)])
def test_obj_loader(ha, rel, nv, nf):
    path = os.path.join(ASSETS, rel)
    m = np.array([[2.0, 0, 0, 1.0], [0, 3.0, 0, -2.0], [0, 0, 0.5, 4.0], [0, 0, 0, 1]])
    verts, faces = ha.load_obj(path, m)
    rv, rf = _parse_obj(path)
    assert verts.shape == (nv, 3) and faces.shape == (nf, 3)      # SURVEY.md Appendix C.3 counts
    assert np.array_equal(faces, rf)
    exp = np.stack([rv[:, 0] * 2.0 + rv[:, 1] * 0.0 + rv[:, 2] * 0.0 + 1.0, rv[:, 0] * 0.0 + rv[:, 1] * 3.0 + rv[:, 2] * 0.0 - 2.0,
                    rv[:, 0] * 0.0 + rv[:, 1] * 0.0 + rv[:, 2] * 0.5 + 4.0], axis=1)
    assert np.array_equal(verts, exp)                               # matrix.rs:180-189 operation order, bit-exact


def test_obj_loader_errors(ha, tmp_path):
    with pytest.raises(ha.HostError):
        ha.load_obj(str(tmp_path / "nope.obj"))
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(ha.HostError):
        ha.load_obj(str(bad))
    empty = tmp_path / "empty.obj"
    empty.write_text("# nothing\n")
    v, f = ha.load_obj(str(empty))
    assert v.shape == (0, 3) and f.shape == (0, 3)


def test_rtcamp6_scene_structure(ha):
    sc = ha.Scene("rtcamp6_v3_1")
    d = sc.desc
    assert d.num_elements == 11 and d.num_images == 7
    kinds = [d.elements[i].kind for i in range(11)]
    assert kinds == [ha.SPHERE, ha.MESH, ha.MESH, ha.MESH, ha.CUBOID] + [ha.MESH] * 6          # main.rs:1043-1150
    assert [d.elements[i].num_faces for i in (1, 2, 3, 5)] == [6170, 12, 112, 1000]
    light = d.elements[0]
    assert light.center.tuple() == (-0.3, 0.7, 0.0) and light.radius == 0.2
    assert light.material.emission.color.tuple() == (30.0, 20.0, 4.0) and light.material.surface == ha.DIFFUSE
    surf = [d.elements[i].material.surface for i in range(11)]
    assert surf == [ha.DIFFUSE, ha.GGX, ha.SPECULAR, ha.GGX, ha.DIFFUSE, ha.REFRACTION, ha.GGX, ha.REFRACTION, ha.GGX, ha.REFRACTION, ha.GGX]
    assert d.elements[6].material.roughness.color.x == pytest.approx(0.05) and d.elements[10].material.roughness.color.x == pytest.approx(0.25)
    assert d.elements[4].material.albedo.image >= 0 and d.elements[4].aabb_min.tuple() == (-9.0, -1.0, -9.0)
    # camera.rs:45-64: tan of the FULL fov angle
    cam = d.camera
    theta = 2 * math.pi * 0.03
    assert cam.eye.tuple() == pytest.approx((6.5 * math.sin(theta), 2.0, 6.5 * math.cos(theta)))
    assert cam.lens_radius == 0.015 and cam.focus_distance == 5.0 and cam.lens_shape == 1
    phu = np.array(cam.plane_half_up.tuple())
    assert np.linalg.norm(phu) == pytest.approx(math.tan(math.radians(20.0)) * 5.0)
    # hsv_to_rgb(0.45, 0.2, 1.0) of armadillo 0 (color.rs:50-61)
    assert d.elements[5].material.albedo.color.tuple() == pytest.approx((0.8, 1.0, 0.94))
    assert sc.image(d.elements[4].material.albedo.image).shape == (3000, 3000, 4)


def test_spheres_scene(ha):
    sc = ha.Scene("spheres")
    d = sc.desc
    assert d.num_elements == 105 and all(d.elements[i].kind == ha.SPHERE for i in range(105))
    c = np.array([d.elements[i].center.tuple() for i in range(105)])
    # scene.rs:366-376: accepted spheres have non-overlapping AABBs (r = 0.1)
    for i in range(105):
        dd = np.abs(c - c[i]).max(axis=1)
        dd[i] = 1.0
        assert (dd >= 0.2).all()
    assert [d.elements[i].material.surface for i in range(4)] == [ha.DIFFUSE, ha.SPECULAR, ha.DIFFUSE, ha.SPECULAR]
    em = [i for i in range(105) if d.elements[i].material.emission.color.tuple() != (0.0, 0.0, 0.0)]
    assert em == list(range(100, 105))
    assert (c[:100, 0] >= -0.5).all() and (c[:100, 0] < 2.0).all() and (np.abs(c[100:, 1]) < 1.0).all()


def test_unknown_scene_and_missing_assets(ha, tmp_path):
    with pytest.raises(ha.HostError):
        ha.Scene("no_such_scene")
    with pytest.raises(ha.HostError):
        ha.Scene("rtcamp6_v3_1", str(tmp_path))


def test_rtcamp5_scene_matches_the_references_committed_render(ha, orc):
    """init_scene_rtcamp5 (main.rs:252-500) places 42 diamonds with ISAAC-64 gen_range draws and AABB-collision rejection over
    mesh boxes; the reference repository ships its render (rtcamp5.png).  A wrong draw order, matrix product or collision test
    moves every later object: compare a low-sample oracle render with the downscaled committed image (the floor texture of that
    older render differs, hence the loose numbers; an unrelated scene scores 54 / 0.64)."""
    from PIL import Image
    sc = ha.Scene("rtcamp5")
    d = sc.desc
    kinds = [d.elements[i].kind for i in range(d.num_elements)]
    assert d.num_elements == 53 and kinds.count(2) == 45 and kinds.count(0) == 7 and kinds.count(1) == 1   # 2 bunnies + 43 diamonds
    o = orc.OracleScene(sc.desc_ptr)
    assert o.num_emissions() == 1                                        # the earth-textured emissive sphere
    acc, _ = o.render(240, 135, 1, 5, threads=0)
    mine = np.asarray(Image.fromarray(orc.resolve(acc, 4)).resize((60, 34), Image.BOX)).astype(float)
    here = os.path.dirname(os.path.abspath(__file__))
    ref = np.asarray(Image.open(os.path.join(here, "golden", "reference_rtcamp5_480x270.png")).resize((60, 34), Image.BOX)).astype(float)
    assert np.abs(mine - ref).mean() < 20.0
    assert np.corrcoef(mine.ravel(), ref.ravel())[0, 1] > 0.9


def test_tbf3_scene_structure(ha, orc):
    sc = ha.Scene("tbf3")
    d = sc.desc
    kinds = [d.elements[i].kind for i in range(d.num_elements)]
    assert d.num_elements == 36 and kinds.count(2) == 23 and kinds.count(0) == 12 and kinds.count(1) == 1   # logo + 22 diamonds, 4 + 8 spheres
    assert orc.OracleScene(sc.desc_ptr).num_emissions() == 4
    assert tuple(round(v, 6) for v in d.skybox.intensity.tuple()) == (2.0, 2.0, 3.0)


def test_wave_budget_control_law_on_the_measured_plant(emu):
    """The governor's second control (device_scene.h gov_budget_next, the function governor_kernel calls) driven on the CPU against the plant that was
    measured on the headline workload (profiles/r05_ab_trace_grid_sweep.txt: trace / seed ms per launch by the number of trace workgroups kept): it
    walks down from "all" to 768 workgroups in three steps and stays; started too low it climbs back; a trace-bound pair keeps every workgroup; and
    when the plant changes under it (a heavier scene) it gives the workgroups back step by step."""
    import ctypes as C
    L = emu.lib()
    L.emu_gov_budget_next.argtypes = [C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]
    L.emu_gov_budget_next.restype = C.c_uint32
    lo, hi, step = 640, 896, 64
    plant = {0: (19.4, 25.3), 896: (20.6, 24.71), 832: (21.3, 24.54), 768: (22.0, 24.38), 704: (23.3, 24.33), 640: (24.5, 24.27)}

    def run(b, table, n=12):
        seen = [b]
        for _ in range(n):
            t, s = table[b]
            b = int(L.emu_gov_budget_next(b, t / s, lo, hi, step))
            seen.append(b)
        return seen
    down = run(0, plant)
    assert down[:4] == [0, 896, 832, 768] and set(down[3:]) == {768}
    assert run(640, plant)[:3] == [640, 704, 704]                      # 640 is trace-bound (ratio 1.01): one step up, 704 holds (0.958)
    heavy = {b: (t * 1.35, s) for b, (t, s) in plant.items()}          # a scene whose trace kernel needs a third more
    assert set(run(0, heavy)) == {0}                                   # never leaves "all"
    back = run(768, heavy)
    assert back[:4] == [768, 832, 896, 0] and set(back[3:]) == {0}     # the plant changed under a settled budget: given back step by step
    assert int(L.emu_gov_budget_next(704, 0.5, lo, hi, 0)) == 704      # a fixed level (step 0) pins the budget
