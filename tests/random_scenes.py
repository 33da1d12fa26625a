"""Random scenes for the fuzz tier of the parity tests: every element kind x every surface type x textured / constant albedo, emission and
roughness, overlapping and nested primitives — combinations none of the reference's eight scenes contains (a GGXRefraction cuboid with an
image roughness, a Specular mesh, a textured emitter inside a glass sphere ...).  Built through the same hr_scene_desc the host library
fills; camera (Camera::new, camera.rs:45-64) through hh_camera_new; images and skybox borrowed from the `cornell_mini` scene."""
import ctypes as C

import numpy as np


def build(ha, seed, spheres=12, cuboids=4, meshes=2):
    rng = np.random.default_rng(seed)
    base = ha.Scene("cornell_mini")
    n = 1 + spheres + cuboids + meshes
    el = (ha.Element * n)()
    keep = [base, el]

    def tex(t, lo, hi, image_chance):
        c = rng.uniform(lo, hi, 3)
        t.color = ha.Vec3(float(c[0]), float(c[1]), float(c[2]))
        t.image = int(rng.integers(0, 3)) if rng.random() < image_chance else -1       # images 0..2 of cornell_mini: 64^2, 32^2, 16^2

    def material(m, emissive=False):
        m.surface = int(rng.integers(0, 5))
        m.param = float(rng.uniform(0.2, 0.95) if m.surface == 3 else rng.uniform(1.1, 2.42))
        tex(m.albedo, 0.2, 1.0, 0.3)
        tex(m.roughness, 0.02, 0.7, 0.3)
        if emissive:
            tex(m.emission, 4.0, 25.0, 0.4)
        else:
            m.emission.color = ha.Vec3(0.0, 0.0, 0.0)
            m.emission.image = -1

    # floor
    el[0].kind = ha.CUBOID
    el[0].aabb_min, el[0].aabb_max = ha.Vec3(-5.0, -1.0, -5.0), ha.Vec3(5.0, 0.0, 5.0)
    material(el[0].material)
    el[0].material.surface = int(rng.choice([0, 3]))
    k = 1
    for i in range(spheres):
        e = el[k]; k += 1
        e.kind = ha.SPHERE
        r = float(rng.uniform(0.12, 0.6))
        e.center = ha.Vec3(float(rng.uniform(-2.2, 2.2)), float(rng.uniform(r * 0.5, 1.8)), float(rng.uniform(-2.2, 2.2)))
        e.radius = r
        material(e.material, emissive=i < 3)                      # three NEE emitters (spheres with a non-zero emission tint), any surface type
    for i in range(cuboids):
        e = el[k]; k += 1
        e.kind = ha.CUBOID
        c = np.array([rng.uniform(-2.0, 2.0), rng.uniform(0.1, 1.2), rng.uniform(-2.0, 2.0)])
        h = rng.uniform(0.1, 0.5, 3)
        e.aabb_min, e.aabb_max = ha.Vec3(*(c - h).tolist()), ha.Vec3(*(c + h).tolist())
        material(e.material, emissive=i == 0)                     # an emissive cuboid: lights the scene, is NOT an NEE emitter (scene.rs:89)
    for i in range(meshes):
        e = el[k]; k += 1
        e.kind = ha.MESH
        c = np.array([rng.uniform(-1.5, 1.5), rng.uniform(0.5, 1.4), rng.uniform(-1.5, 1.5)])
        s = rng.uniform(0.3, 0.7)
        rot = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        octa = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
        verts = np.ascontiguousarray(octa @ rot.T * s + c)
        faces = np.ascontiguousarray(np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], dtype=np.uint64))
        e.vertexes = verts.ctypes.data_as(C.POINTER(ha.Vec3)); e.num_vertexes = verts.shape[0]
        e.faces = faces.ctypes.data_as(C.POINTER(C.c_uint64)); e.num_faces = faces.shape[0]
        material(e.material, emissive=False)
        keep += [verts, faces]
    d = ha.SceneDesc()
    C.memmove(C.byref(d), base.desc_ptr, C.sizeof(d))
    d.elements = C.cast(el, C.POINTER(ha.Element))
    d.num_elements = n
    eye = ha.Vec3(float(rng.uniform(-1.5, 1.5)), float(rng.uniform(1.0, 2.5)), float(rng.uniform(5.0, 7.0)))
    ha.host_lib().hh_camera_new(eye, ha.Vec3(0.0, 0.7, 0.0), ha.Vec3(0.0, 1.0, 0.0), float(rng.uniform(18.0, 32.0)), int(rng.integers(0, 2)),
                                float(rng.uniform(0.0, 0.15)), float(rng.uniform(5.0, 7.0)), C.byref(d.camera))

    class Holder:
        pass
    h = Holder()
    h.desc_ptr = C.pointer(d)
    h.keep = keep + [d]
    h.num_elements = n
    return h
