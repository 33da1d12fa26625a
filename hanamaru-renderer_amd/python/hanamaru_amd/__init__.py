"""ctypes plumbing over the two C-ABI libraries of the MI355X hanamaru back end.

  libhanamaru_host.so  — scene authoring / asset IO (include/hanamaru_host.h)
  libhanamaru_hip.so   — the HIP render path      (include/hanamaru_hip.h)

This module holds no algorithm: it mirrors the C structs and forwards calls.  The HIP library is
required for anything that renders — there is no CPU fallback (`Renderer` raises if it cannot load).
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.normpath(os.path.join(_HERE, "..", ".."))
REPO_ROOT = os.path.normpath(os.path.join(PKG_ROOT, ".."))
ASSET_ROOT = os.path.join(REPO_ROOT, "assets")
HOST_LIB = os.path.join(PKG_ROOT, "libhanamaru_host.so")
HIP_LIB = os.path.join(PKG_ROOT, "libhanamaru_hip.so")

HR_OK = 0
HR_ERR_RNG_WINDOW = -5
DIFFUSE, SPECULAR, REFRACTION, GGX, GGX_REFRACTION = range(5)
SPHERE, CUBOID, MESH = range(3)


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def tuple(self):
        return (self.x, self.y, self.z)


class Texture(C.Structure):
    _fields_ = [("color", Vec3), ("image", C.c_int32), ("_pad", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("surface", C.c_int32), ("_pad", C.c_int32), ("param", C.c_double),
                ("albedo", Texture), ("emission", Texture), ("roughness", Texture)]


class Image(C.Structure):
    _fields_ = [("rgba", C.POINTER(C.c_uint8)), ("width", C.c_uint32), ("height", C.c_uint32)]


class Element(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("material", Material),
                ("center", Vec3), ("radius", C.c_double),
                ("aabb_min", Vec3), ("aabb_max", Vec3),
                ("vertexes", C.POINTER(Vec3)), ("num_vertexes", C.c_uint64),
                ("faces", C.POINTER(C.c_uint64)), ("num_faces", C.c_uint64)]


class Camera(C.Structure):
    _fields_ = [("eye", Vec3), ("right", Vec3), ("up", Vec3), ("forward", Vec3),
                ("plane_half_right", Vec3), ("plane_half_up", Vec3),
                ("lens_radius", C.c_double), ("focus_distance", C.c_double),
                ("lens_shape", C.c_int32), ("_pad", C.c_int32)]


class Skybox(C.Structure):
    _fields_ = [("face_image", C.c_int32 * 6), ("intensity", Vec3)]


class SceneDesc(C.Structure):
    _fields_ = [("elements", C.POINTER(Element)), ("num_elements", C.c_uint32),
                ("images", C.POINTER(Image)), ("num_images", C.c_uint32),
                ("skybox", Skybox), ("camera", Camera)]


class CommInfo(C.Structure):
    _fields_ = [("path", C.c_int32), ("nranks", C.c_int32), ("rank", C.c_int32), ("device", C.c_int32),
                ("rccl_version", C.c_int32), ("_pad", C.c_int32), ("allreduces", C.c_uint64)]


COMM_PATHS = {0: "none", 1: "rccl-rank", 2: "rccl-group", 3: "same-device-fallback"}


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("node_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("sphere_tests", C.c_uint64), ("cuboid_tests", C.c_uint64),
                ("rng_overflow", C.c_uint64),
                ("seed_kernel_ms", C.c_double), ("trace_kernel_ms", C.c_double), ("post_kernel_ms", C.c_double),
                ("seed_launches", C.c_uint64), ("trace_launches", C.c_uint64),
                ("bvh_nodes", C.c_uint64), ("triangles", C.c_uint64), ("spheres", C.c_uint64), ("cuboids", C.c_uint64),
                ("shade_calls", C.c_uint64), ("shade_lanes", C.c_uint64), ("box_passes", C.c_uint64), ("box_lanes", C.c_uint64),
                ("leaf_calls", C.c_uint64), ("leaf_lanes", C.c_uint64), ("outer_iters", C.c_uint64), ("phase_cycles", C.c_uint64 * 4),
                ("bvh_build_ms", C.c_double), ("seed_phase_cycles", C.c_uint64 * 8),
                ("debug_kernel_ms", C.c_double), ("debug_launches", C.c_uint64),
                ("governor_level", C.c_uint64), ("governor_decisions", C.c_uint64), ("governor_moves", C.c_uint64), ("shadow_culled", C.c_uint64), ("governor_budget", C.c_uint64), ("governor_budget_moves", C.c_uint64), ("bvh_builder_used", C.c_uint64), ("shading_in_force", C.c_uint64)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if hasattr(getattr(self, k), "__len__") else getattr(self, k)) for k, _ in self._fields_}


_host = None
_hip = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("libhanamaru_host.so not built — run __graft_entry__.build() / make -C hanamaru-renderer_amd")
        L = C.CDLL(HOST_LIB)
        L.hh_last_error.restype = C.c_char_p
        L.hh_scene_create.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.hh_scene_desc.argtypes = [C.c_void_p]
        L.hh_scene_desc.restype = C.POINTER(SceneDesc)
        L.hh_scene_destroy.argtypes = [C.c_void_p]
        L.hh_decode_image.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.hh_write_png_rgb8.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.hh_load_obj.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.POINTER(Vec3)), C.POINTER(C.c_uint64),
                                  C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
        L.hh_free.argtypes = [C.c_void_p]
        L.hh_debug_isaac64.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]
        L.hh_camera_new.argtypes = [Vec3, Vec3, Vec3, C.c_double, C.c_int32, C.c_double, C.c_double, C.POINTER(Camera)]
        _host = L
    return _host


def hip_lib():
    """Load the HIP back end.  Fails loudly: there is no fallback path."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError("libhanamaru_hip.so not built — run __graft_entry__.build() / make -C hanamaru-renderer_amd")
        # One HIP runtime per process: PyTorch ships its own libamdhip64.  Loaded after this library's (the system's, through DT_NEEDED)
        # it becomes a second runtime that finds no GPU ("No HIP GPUs are available"); loaded first, its SONAME satisfies this
        # library's dependency and both share it.  A process that will use torch next to this wrapper (bench.py binds torch tensors and
        # torch.distributed; some tests do) must therefore have torch loaded before the first CDLL below — do it here, once.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(HIP_LIB)
        L.hr_last_error.restype = C.c_char_p
        L.hr_abi_version.restype = C.c_int
        L.hr_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.hr_destroy.argtypes = [C.c_void_p]
        L.hr_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.hr_set_resolution.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.hr_bind_accumulator.argtypes = [C.c_void_p, C.c_void_p]
        L.hr_accumulator_device_ptr.argtypes = [C.c_void_p]
        L.hr_accumulator_device_ptr.restype = C.c_void_p
        L.hr_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.hr_clear.argtypes = [C.c_void_p]
        L.hr_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.hr_synchronize.argtypes = [C.c_void_p]
        L.hr_mark.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.hr_wait.argtypes = [C.c_void_p, C.c_uint64]
        L.hr_render_debug.argtypes = [C.c_void_p, C.c_int]
        L.hr_read_accumulator.argtypes = [C.c_void_p, C.c_void_p]
        L.hr_write_accumulator.argtypes = [C.c_void_p, C.c_void_p]
        L.hr_resolve.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.hr_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.hr_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.hr_set_debug_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.hr_debug_trace.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hr_debug_draws.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.hr_debug_intersect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hr_debug_path_log.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.hr_debug_path_draws.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.hr_debug_path_draw_residuals.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.hr_debug_wf_profile.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.hr_comm_get_unique_id.argtypes = [C.c_void_p]
        L.hr_comm_init_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.hr_comm_init_local.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.hr_comm_destroy.argtypes = [C.c_void_p]
        L.hr_allreduce_accumulator.argtypes = [C.c_void_p]
        L.hr_allreduce_accumulators.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.hr_total_device_ptr.argtypes = [C.c_void_p]
        L.hr_total_device_ptr.restype = C.c_void_p
        L.hr_comm_info.argtypes = [C.c_void_p, C.POINTER(CommInfo)]
        L.hr_accumulator_sum.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        _hip = L
    return _hip


COMM_ID_BYTES = 128


def comm_unique_id():
    """ncclGetUniqueId through the library: bytes to hand to every rank's Renderer.comm_init_rank."""
    L = hip_lib()
    buf = (C.c_char * COMM_ID_BYTES)()
    rc = L.hr_comm_get_unique_id(buf)
    if rc != 0:
        raise HipError(rc, L.hr_last_error().decode())
    return bytes(buf)


def comm_library():
    """hr_comm_library: (path of the RCCL shared object the library's collective runs on, True when it reused one already mapped into the process)."""
    L = hip_lib()
    L.hr_comm_library.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    buf = C.create_string_buffer(4096)
    reused = C.c_int(0)
    rc = L.hr_comm_library(buf, 4096, C.byref(reused))
    if rc != 0:
        raise HipError(rc, L.hr_last_error().decode())
    return buf.value.decode(), bool(reused.value)


def comm_init_local(renderers):
    """One process driving several GPUs: ncclCommInitAll over the renderers' devices."""
    L = hip_lib()
    arr = (C.c_void_p * len(renderers))(*[r._h for r in renderers])
    rc = L.hr_comm_init_local(arr, len(renderers))
    if rc != 0:
        raise HipError(rc, L.hr_last_error().decode())


def allreduce_accumulators(renderers):
    L = hip_lib()
    arr = (C.c_void_p * len(renderers))(*[r._h for r in renderers])
    rc = L.hr_allreduce_accumulators(arr, len(renderers))
    if rc != 0:
        raise HipError(rc, L.hr_last_error().decode())


class HostError(RuntimeError):
    pass


class HipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("hr error %d: %s" % (code, text))
        self.code = code


class Scene:
    """Owns an hh_scene (host memory) and exposes its hr_scene_desc."""

    def __init__(self, name, asset_root=ASSET_ROOT):
        L = host_lib()
        h = C.c_void_p()
        rc = L.hh_scene_create(name.encode(), asset_root.encode(), C.byref(h))
        if rc != 0:
            raise HostError("hh_scene_create(%s): %s" % (name, L.hh_last_error().decode()))
        self._h = h
        self.name = name
        self.desc_ptr = L.hh_scene_desc(h)
        self.desc = self.desc_ptr.contents

    def close(self):
        if self._h:
            host_lib().hh_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def image(self, i):
        im = self.desc.images[i]
        return np.ctypeslib.as_array(im.rgba, shape=(im.height, im.width, 4))


def decode_image(path):
    L = host_lib()
    p = C.POINTER(C.c_uint8)()
    w, h = C.c_uint32(), C.c_uint32()
    if L.hh_decode_image(path.encode(), C.byref(p), C.byref(w), C.byref(h)) != 0:
        raise HostError(L.hh_last_error().decode())
    arr = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    L.hh_free(p)
    return arr


def write_png(path, rgb8):
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    if host_lib().hh_write_png_rgb8(path.encode(), a.ctypes.data, a.shape[1], a.shape[0]) != 0:
        raise HostError(host_lib().hh_last_error().decode())


def load_obj(path, matrix=None):
    L = host_lib()
    v, f = C.POINTER(Vec3)(), C.POINTER(C.c_uint64)()
    nv, nf = C.c_uint64(), C.c_uint64()
    m = None
    if matrix is not None:
        m = (C.c_double * 16)(*np.asarray(matrix, dtype=np.float64).reshape(16))
    if L.hh_load_obj(path.encode(), m, C.byref(v), C.byref(nv), C.byref(f), C.byref(nf)) != 0:
        raise HostError(L.hh_last_error().decode())
    verts = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_double)), shape=(nv.value, 3)).copy()
    faces = np.ctypeslib.as_array(f, shape=(nf.value, 3)).copy()
    L.hh_free(v)
    L.hh_free(f)
    return verts, faces


class Renderer:
    """Thin wrapper over hr_ctx — mirrors the reference's Renderer trait usage (renderer.rs:20-99):
    render() accumulates samplings, resolve() is update_imgbuf."""

    def __init__(self, device=0):
        self.L = hip_lib()
        h = C.c_void_p()
        self._check(self.L.hr_create(device, C.byref(h)))
        self._h = h
        self.width = self.height = 0

    def _check(self, rc):
        if rc != 0:
            raise HipError(rc, self.L.hr_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self.L.hr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, scene):
        self._scene = scene  # keep host memory alive during the call
        self._check(self.L.hr_upload_scene(self._h, scene.desc_ptr))

    def set_resolution(self, w, h):
        self._check(self.L.hr_set_resolution(self._h, w, h))
        self.width, self.height = w, h

    def bind_accumulator(self, device_ptr):
        self._check(self.L.hr_bind_accumulator(self._h, device_ptr))

    def set_stream(self, stream_ptr):
        self._check(self.L.hr_set_stream(self._h, stream_ptr))

    def set_option(self, key, value):
        self._check(self.L.hr_set_option(self._h, key.encode(), float(value)))

    def set_debug_option(self, key, value):
        """Measurement knobs (include/hanamaru_hip.h: hr_set_debug_option) — not for product code."""
        self._check(self.L.hr_set_debug_option(self._h, key.encode(), float(value)))

    def clear(self):
        self._check(self.L.hr_clear(self._h))

    def render(self, begin, end, stride=1):
        self._check(self.L.hr_render(self._h, begin, end, stride))

    def render_debug(self, mode):
        self._check(self.L.hr_render_debug(self._h, mode))

    def mark(self):
        t = C.c_uint64()
        self._check(self.L.hr_mark(self._h, C.byref(t)))
        return t.value

    def wait(self, ticket):
        self._check(self.L.hr_wait(self._h, C.c_uint64(ticket)))

    def synchronize(self):
        self._check(self.L.hr_synchronize(self._h))

    def read_accumulator(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._check(self.L.hr_read_accumulator(self._h, out.ctypes.data))
        return out

    def write_accumulator(self, acc):
        a = np.ascontiguousarray(acc, dtype=np.float32)
        assert a.shape == (self.height, self.width, 3)
        self._check(self.L.hr_write_accumulator(self._h, a.ctypes.data))

    def resolve(self, samplings_done):
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._check(self.L.hr_resolve(self._h, samplings_done, out.ctypes.data))
        return out

    def stats(self):
        s = Stats()
        self._check(self.L.hr_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    # ---- multi-GPU: one all-reduce of the accumulators over RCCL (include/hanamaru_hip.h)
    def comm_init_rank(self, unique_id, world_size, rank):
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._check(self.L.hr_comm_init_rank(self._h, buf, world_size, rank))

    def comm_destroy(self):
        self._check(self.L.hr_comm_destroy(self._h))

    def allreduce_accumulator(self):
        self._check(self.L.hr_allreduce_accumulator(self._h))

    def total_device_ptr(self):
        return self.L.hr_total_device_ptr(self._h)

    def comm_info(self):
        """hr_comm_info: what the communicator reports about itself (ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclGetVersion)."""
        ci = CommInfo()
        self._check(self.L.hr_comm_info(self._h, C.byref(ci)))
        return {"path": COMM_PATHS.get(ci.path, "?"), "nranks": int(ci.nranks), "rank": int(ci.rank), "device": int(ci.device),
                "rccl_version": int(ci.rccl_version), "allreduces": int(ci.allreduces)}

    def accumulator_sum(self, total=False):
        """hr_accumulator_sum: per-channel f64 sums (device reduction) of this context's own accumulator, or of the all-reduced total."""
        out = (C.c_double * 3)()
        self._check(self.L.hr_accumulator_sum(self._h, 1 if total else 0, out))
        return [float(out[0]), float(out[1]), float(out[2])]

    def debug_draws(self, sampling, first_path, num_paths, window):
        out = np.empty((num_paths, window), dtype=np.uint64)
        self._check(self.L.hr_debug_draws(self._h, sampling, first_path, num_paths, window, out.ctypes.data))
        return out

    def debug_path_log(self, sampling):
        """hr_debug_path_log: (radiance [H, W, 4, 3] float32, rays [H, W, 4] uint32, events [H, W, 4, 12] uint8 — nine event bytes, the count of sphere hits, the 16-bit texel-quad sum —, hash [H, W, 4] uint32) of every
        path of one sampling, from the render kernel's logging instantiation."""
        raw = np.zeros((self.height, self.width, 4, 8), dtype=np.uint32)
        self._check(self.L.hr_debug_path_log(self._h, sampling, raw.ctypes.data))
        rad = raw[..., 0:3].copy().view(np.float32)
        ev = np.ascontiguousarray(raw[..., 4:7]).view(np.uint8).reshape(self.height, self.width, 4, 12)[..., :12]
        return rad, raw[..., 3].copy(), ev.copy(), raw[..., 7].copy()

    def debug_wf_profile(self, sampling=1, num_k=4):
        """hr_debug_wf_profile: (ms[21], counts[11, 2]) of one launch of the split pipeline, kernel by kernel (ms[0] camera rays, ms[2s-1] / ms[2s]
        traversal / shading of step s; counts[s] = rays, live paths of step s)."""
        ms = np.zeros(21, dtype=np.float64)
        cn = np.zeros(22, dtype=np.uint32)
        self._check(self.L.hr_debug_wf_profile(self._h, sampling, num_k, ms.ctypes.data, cn.ctypes.data))
        return ms, cn.reshape(11, 2)

    def debug_intersect(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        out = np.empty((r.shape[0], 8), dtype=np.float32)
        el = np.empty((r.shape[0],), dtype=np.int32)
        self._check(self.L.hr_debug_intersect(self._h, r.shape[0], r.ctypes.data, out.ctypes.data, el.ctypes.data))
        return out, el

    def debug_trace(self, rays, shadow_len=None):
        """hr_debug_trace: the same queries through the render kernel's traversal; shadow_len > 0 marks shadow rays."""
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        out = np.empty((r.shape[0], 8), dtype=np.float32)
        el = np.empty((r.shape[0],), dtype=np.int32)
        sl = None
        if shadow_len is not None:
            sl = np.ascontiguousarray(shadow_len, dtype=np.float32).reshape(-1)
            assert sl.shape[0] == r.shape[0]
        self._check(self.L.hr_debug_trace(self._h, r.shape[0], r.ctypes.data, sl.ctypes.data if sl is not None else None, out.ctypes.data, el.ctypes.data))
        return out, el
