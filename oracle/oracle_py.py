"""ctypes loader for oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The oracle consumes the same hr_scene_desc (plain data) the HIP library does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

COUNTER_FIELDS = ["paths", "rays_primary", "rays_bounce", "rays_shadow", "surface_hits", "draws", "tex_samples",
                  "sky_lookups", "top_node_tests", "mesh_roots", "mesh_node_tests", "tri_tests", "tri_accepted",
                  "sphere_tests", "cuboid_tests"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("oracle/liboracle.so not built — run `make -C oracle`")
        L = C.CDLL(LIB_PATH)
        L.orc_scene_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_counters_size.restype = C.c_size_t
        L.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_render_region.argtypes = [C.c_void_p] + [C.c_uint32] * 9 + [C.c_int, C.c_void_p]
        L.orc_render_debug.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_calc_pixel.argtypes = [C.c_void_p] + [C.c_uint32] * 7 + [C.c_void_p]
        L.orc_path_draws.argtypes = [C.c_uint32] * 7 + [C.c_void_p, C.c_int]
        L.orc_isaac64.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_int]
        L.orc_u64_to_f64.argtypes = [C.c_uint64]
        L.orc_u64_to_f64.restype = C.c_double
        L.orc_intersect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_intersect_material.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_material_sample.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_material_bsdf.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_material_bsdf.restype = C.c_double
        L.orc_skybox_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_image_sample_bilinear.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_void_p]
        L.orc_bvh_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_top_leaves.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_num_emissions.argtypes = [C.c_void_p]
        L.orc_path_log.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_resolve.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class OracleScene:
    def __init__(self, desc_ptr):
        """desc_ptr: ctypes pointer to an hr_scene_desc (e.g. hanamaru_amd.Scene.desc_ptr)."""
        L = lib()
        h = C.c_void_p()
        rc = L.orc_scene_create(C.cast(desc_ptr, C.c_void_p), C.byref(h))
        if rc != 0:
            raise RuntimeError("orc_scene_create failed: %d" % rc)
        self._h = h

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def path_log(self, w, h, sampling, threads=0):
        """Every path of one sampling: (radiance [h, w, 4, 3] float64, rays [h, w, 4] uint32, events [h, w, 4, 12] uint8 (nine event bytes, the count of sphere hits, the 16-bit texel-quad sum), element hash [h, w, 4]
        uint32) — the oracle's side of hr_debug_path_log (same event encoding, oracle.cpp PathLog)."""
        rad = np.zeros((h, w, 4, 3), dtype=np.float64)
        words = np.zeros((h, w, 4, 5), dtype=np.uint32)
        rc = lib().orc_path_log(self._h, w, h, sampling, threads, rad.ctypes.data, words.ctypes.data)
        if rc != 0:
            raise RuntimeError("orc_path_log failed: %d" % rc)
        ev = np.ascontiguousarray(words[..., 1:4]).view(np.uint8).reshape(h, w, 4, 12)[..., :12]
        return rad, words[..., 0].copy(), ev.copy(), words[..., 4].copy()

    def render(self, w, h, s_begin, s_end, stride=1, threads=0, acc=None, counters=False):
        """Accumulate samplings into a float64 (h, w, 3) array; returns (acc, counters dict or None)."""
        if acc is None:
            acc = np.zeros((h, w, 3), dtype=np.float64)
        cbuf = None
        if counters:
            cbuf = (C.c_uint64 * (lib().orc_counters_size() // 8))()
        rc = lib().orc_render(self._h, w, h, s_begin, s_end, stride, threads, acc.ctypes.data, cbuf)
        if rc != 0:
            raise RuntimeError("orc_render failed: %d" % rc)
        cd = None
        if counters:
            vals = list(cbuf)
            cd = dict(zip(COUNTER_FIELDS, vals[:len(COUNTER_FIELDS)]))
            cd["rays_per_path_hist"] = vals[len(COUNTER_FIELDS):len(COUNTER_FIELDS) + 24]
            cd["draws_per_path_hist"] = vals[len(COUNTER_FIELDS) + 24:len(COUNTER_FIELDS) + 64]
        return acc, cd

    def render_region(self, w, h, x0, y0, rw, rh, s_begin, s_end, stride=1, threads=0):
        acc = np.zeros((rh, rw, 3), dtype=np.float64)
        rc = lib().orc_render_region(self._h, w, h, x0, y0, rw, rh, s_begin, s_end, stride, threads, acc.ctypes.data)
        if rc != 0:
            raise RuntimeError("orc_render_region failed: %d" % rc)
        return acc

    def render_debug(self, w, h, mode):
        acc = np.zeros((h, w, 3), dtype=np.float64)
        rc = lib().orc_render_debug(self._h, w, h, mode, acc.ctypes.data)
        assert rc == 0
        return acc

    def calc_pixel(self, w, h, x, y, sx, sy, sampling):
        out = np.zeros(3, dtype=np.float64)
        lib().orc_calc_pixel(self._h, w, h, x, y, sx, sy, sampling, out.ctypes.data)
        return out

    def intersect(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.empty((r.shape[0], 8), dtype=np.float64)
        el = np.empty((r.shape[0],), dtype=np.int32)
        lib().orc_intersect(self._h, r.shape[0], r.ctypes.data, out.ctypes.data, el.ctypes.data)
        return out, el

    def intersect_material(self, origin, direction):
        """Closest hit with the material fetch (scene.rs:385-401): dict of hit, distance, position, normal, u, v, surface, param,
        albedo, emission, roughness, element."""
        r = np.ascontiguousarray(list(origin) + list(direction), dtype=np.float64)
        out = np.zeros(19, dtype=np.float64)
        el = C.c_int32(-1)
        lib().orc_intersect_material(self._h, r.ctypes.data, out.ctypes.data, C.byref(el))
        return {"hit": out[0] == 1.0, "distance": out[1], "position": out[2:5].copy(), "normal": out[5:8].copy(), "u": out[8], "v": out[9],
                "surface": int(out[10]), "param": out[11], "albedo": out[12:15].copy(), "emission": out[15:18].copy(), "roughness": out[18],
                "element": el.value}

    def skybox(self, d):
        d = np.ascontiguousarray(d, dtype=np.float64)
        out = np.zeros(3, dtype=np.float64)
        lib().orc_skybox_sample(self._h, d.ctypes.data, out.ctypes.data)
        return out

    def image_bilinear(self, image, u, v):
        out = np.zeros(3, dtype=np.float64)
        rc = lib().orc_image_sample_bilinear(self._h, image, u, v, out.ctypes.data)
        assert rc == 0
        return out

    def bvh_stats(self, element):
        st = (C.c_uint64 * 11)()
        bb = (C.c_double * 6)()
        rc = lib().orc_bvh_stats(self._h, element, st, bb)
        if rc != 0:
            return None
        v = list(st)
        return {"nodes": v[0], "leaves": v[1], "max_depth": v[2], "leaf_hist": v[3:11], "aabb": list(bb)}

    def top_leaves(self):
        buf = (C.c_int32 * 256)()
        n = lib().orc_top_leaves(self._h, buf, 256)
        leaves, cur = [], []
        for i in range(n):
            if buf[i] < 0:
                leaves.append(cur)
                cur = []
            else:
                cur.append(buf[i])
        return leaves

    def num_emissions(self):
        return lib().orc_num_emissions(self._h)


def material_sample(surface, param, roughness, r0, r1, position, view, normal):
    """PointMaterial::sample (material.rs:91-151): (some, origin, direction, reflectance)."""
    p, v, n = (np.ascontiguousarray(a, dtype=np.float64) for a in (position, view, normal))
    out = np.zeros(8, dtype=np.float64)
    lib().orc_material_sample(surface, param, roughness, r0, r1, p.ctypes.data, v.ctypes.data, n.ctypes.data, out.ctypes.data)
    return out[0] == 1.0, out[1:4].copy(), out[4:7].copy(), float(out[7])


def material_bsdf(surface, param, roughness, view, normal, light):
    v, n, l = (np.ascontiguousarray(a, dtype=np.float64) for a in (view, normal, light))
    return lib().orc_material_bsdf(surface, param, roughness, v.ctypes.data, n.ctypes.data, l.ctypes.data)


def isaac64(seed, count, skip=0):
    s = np.asarray(seed, dtype=np.uint64)
    out = np.empty(count, dtype=np.uint64)
    lib().orc_isaac64(s.ctypes.data, len(s), skip, out.ctypes.data, count)
    return out


def path_draws(w, h, x, y, sx, sy, sampling, count):
    out = np.empty(count, dtype=np.uint64)
    lib().orc_path_draws(w, h, x, y, sx, sy, sampling, out.ctypes.data, count)
    return out


def u64_to_f64(v):
    return lib().orc_u64_to_f64(int(v))


def resolve(acc, samplings, want_stage=False):
    a = np.ascontiguousarray(acc, dtype=np.float64)
    h, w, _ = a.shape
    out = np.empty((h, w, 3), dtype=np.uint8)
    stage = np.empty((h, w, 3), dtype=np.float64) if want_stage else None
    rc = lib().orc_resolve(a.ctypes.data, w, h, samplings, out.ctypes.data, stage.ctypes.data if want_stage else None)
    if rc != 0:
        raise RuntimeError("orc_resolve failed: %d" % rc)
    return (out, stage) if want_stage else out
