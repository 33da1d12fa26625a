"""ctypes loader for tests/emu/libhr_emu.so (host emulation of the kernels' per-lane code; tests only)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhr_emu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.emu_scene_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.emu_scene_destroy.argtypes = [C.c_void_p]
        L.emu_set_build_options.argtypes = [C.c_int, C.c_double]
        L.emu_set_builder.argtypes = [C.c_int]
        L.emu_set_walk_mode.argtypes = [C.c_int]
        L.emu_scene_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_path_draws.argtypes = [C.c_uint32] * 6 + [C.c_int, C.c_void_p]
        L.emu_raw_draws.argtypes = [C.c_uint32] * 6 + [C.c_int, C.c_void_p]
        L.emu_raw_draws_split.argtypes = [C.c_uint32] * 6 + [C.c_int, C.c_void_p]
        L.emu_raw_draws_pc.argtypes = [C.c_uint32] * 6 + [C.c_int, C.c_int, C.c_void_p]
        L.emu_raw_draws_seg.argtypes = [C.c_uint32] * 6 + [C.c_int, C.c_void_p]
        L.emu_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.emu_render_wf.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.emu_path_log_wf.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.emu_set_wf_precise.argtypes = [C.c_int]
        L.emu_path_log.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.emu_render_debug.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.emu_intersect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.emu_resolve.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


class EmuScene:
    def __init__(self, desc_ptr):
        h = C.c_void_p()
        rc = lib().emu_scene_create(C.cast(desc_ptr, C.c_void_p), C.byref(h))
        if rc != 0:
            raise RuntimeError("emu_scene_create failed: %d" % rc)
        self._h = h

    def __del__(self):
        try:
            if self._h:
                lib().emu_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def stats(self):
        out = (C.c_uint64 * 7)()
        lib().emu_scene_stats(self._h, out)
        return dict(zip(["nodes", "leaves", "max_depth", "tris", "spheres", "cuboids", "emitters"], list(out)))

    def render(self, w, h, s_begin, s_end, stride=1, threads=0, acc=None):
        if acc is None:
            acc = np.zeros((h, w, 3), dtype=np.float32)
        cn = (C.c_uint64 * 7)()
        lib().emu_render(self._h, w, h, s_begin, s_end, stride, threads, acc.ctypes.data, cn)
        return acc, dict(zip(["paths", "rays", "node_tests", "tri_tests", "sphere_tests", "cuboid_tests", "shadow_culled"], list(cn)))

    def render_wf(self, w, h, s_begin, s_end, stride=1, threads=0):
        """the split pipeline's per-lane functions (csrc/wf_core.h), path by path"""
        acc = np.zeros((h, w, 3), dtype=np.float32)
        lib().emu_render_wf(self._h, w, h, s_begin, s_end, stride, threads, acc.ctypes.data)
        return acc

    def path_log_wf(self, w, h, sampling, threads=0):
        """path_log's layout from the split pipeline with PRECISE shading (wf_surface_f64)"""
        raw = np.zeros((h, w, 4, 8), dtype=np.uint32)
        lib().emu_path_log_wf(self._h, w, h, sampling, threads, raw.ctypes.data)
        rad = raw[..., 0:3].copy().view(np.float32)
        ev = np.ascontiguousarray(raw[..., 4:7]).view(np.uint8).reshape(h, w, 4, 12)[..., :12]
        return rad, raw[..., 3].copy(), ev.copy(), raw[..., 7].copy()

    def path_log(self, w, h, sampling, threads=0):
        """(radiance [h, w, 4, 3] float32, rays, events [h, w, 4, 12] uint8 (nine event bytes, the count of sphere hits, the 16-bit texel-quad sum), element hash) — the layout of Renderer.debug_path_log"""
        raw = np.zeros((h, w, 4, 8), dtype=np.uint32)
        lib().emu_path_log(self._h, w, h, sampling, threads, raw.ctypes.data)
        rad = raw[..., 0:3].copy().view(np.float32)
        ev = np.ascontiguousarray(raw[..., 4:7]).view(np.uint8).reshape(h, w, 4, 12)[..., :12]
        return rad, raw[..., 3].copy(), ev.copy(), raw[..., 7].copy()

    def one_path(self, w, h, x, y, sub, sampling):
        """radiance of ONE path (fp32 or precise shading as set_precise says)"""
        out = np.zeros(3, dtype=np.float32)
        lib().emu_one_path.argtypes = [C.c_void_p] + [C.c_uint32] * 6 + [C.c_void_p]
        lib().emu_one_path(self._h, w, h, x, y, sub, sampling, out.ctypes.data)
        return out

    def render_debug(self, w, h, mode):
        acc = np.zeros((h, w, 3), dtype=np.float32)
        lib().emu_render_debug(self._h, w, h, mode, acc.ctypes.data)
        return acc

    def intersect(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        out = np.empty((r.shape[0], 8), dtype=np.float32)
        el = np.empty((r.shape[0],), dtype=np.int32)
        lib().emu_intersect(self._h, r.shape[0], r.ctypes.data, out.ctypes.data, el.ctypes.data)
        return out, el


def set_build_options(max_leaf=4, split_ratio=-1.0, builder=0):
    """builder: 0 = host SAH, 1 = LBVH, 2 = PLOC (the device builders' per-thread code, run sequentially)."""
    lib().emu_set_build_options(max_leaf, split_ratio)
    lib().emu_set_builder(builder)


def set_walk_mode(mode):
    """0 = node + leaf per visit, 1 = the trace kernel's postponed-leaf schedule, 2 = that schedule on the 16-byte quantised
    nodes (EmuScene.intersect)."""
    lib().emu_set_walk_mode(mode)


def set_draw_residuals(on):
    """precise shading with (default, as the library renders) or without the records' twin of draw residuals (isaac_core.h draw_lo_f32)"""
    lib().emu_set_draw_residuals.argtypes = [C.c_int]
    lib().emu_set_draw_residuals(1 if on else 0)


def set_precise(on):
    """option precise_shading for render / path_log (the megakernel's per-lane code, path_advance<.., PREC>)"""
    lib().emu_set_precise.argtypes = [C.c_int]
    lib().emu_set_precise(1 if on else 0)


def set_nee_cull(on):
    """nee_setup's shortcuts (pt_core.h): shadow rays known to add nothing are not traced (default on)."""
    lib().emu_set_nee_cull.argtypes = [C.c_int]
    lib().emu_set_nee_cull(7 if on is True else int(on))


def last_node_tests():
    lib().emu_last_node_tests.restype = C.c_uint64
    return int(lib().emu_last_node_tests())


def path_draws(w, h, x, y, sub, sampling, lens_shape=1):
    out = np.empty(20, dtype=np.float32)
    rc = lib().emu_path_draws(w, h, x, y, sub, sampling, lens_shape, out.ctypes.data)
    return out, rc == 0


def path_draw_residuals(w, h, x, y, sub, sampling, lens_shape=1):
    """the record's twin (precise shading): f64 draw k of the path = float(draw k) + residual k to 2^-49; [0], [1] belong to the raw lens draws"""
    out = np.empty(20, dtype=np.float32)
    lib().emu_path_draw_residuals.argtypes = [C.c_uint32] * 6 + [C.c_int, C.c_void_p]
    rc = lib().emu_path_draw_residuals(w, h, x, y, sub, sampling, lens_shape, out.ctypes.data)
    return out, rc == 0


def raw_draws(w, h, x, y, sub, sampling, window=64):
    out = np.empty(window, dtype=np.uint64)
    lib().emu_raw_draws(w, h, x, y, sub, sampling, window, out.ctypes.data)
    return out


def raw_draws_split(w, h, x, y, sub, sampling, window=64):
    out = np.empty(window, dtype=np.uint64)
    lib().emu_raw_draws_split(w, h, x, y, sub, sampling, window, out.ctypes.data)
    return out


def raw_draws_seg(w, h, x, y, sub, sampling, window=64):
    out = np.empty(window, dtype=np.uint64)
    rc = lib().emu_raw_draws_seg(w, h, x, y, sub, sampling, window, out.ctypes.data)
    assert rc == 0
    return out


def raw_draws_pc(w, h, x, y, sub, sampling, head, window=64):
    out = np.empty(window, dtype=np.uint64)
    rc = lib().emu_raw_draws_pc(w, h, x, y, sub, sampling, window, head, out.ctypes.data)
    assert rc == 0
    return out


def resolve(acc, samplings):
    a = np.ascontiguousarray(acc, dtype=np.float32)
    h, w, _ = a.shape
    out = np.empty((h, w, 3), dtype=np.uint8)
    lib().emu_resolve(a.ctypes.data, w, h, samplings, out.ctypes.data)
    return out
