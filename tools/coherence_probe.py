#!/usr/bin/env python3
"""tools/coherence_probe.py [scene] [bvh_builder] — what would re-queuing bounce rays by direction octant buy the traversal?  

Upper-bound experiment WITHOUT the queue machinery: the bounce rays of a real view (origins on the visible surfaces of the scene, directions
cosine-distributed about the surface normal — the second ray of a diffuse path) are sent through the production traversal
(hr_debug_trace = traverse_wave on the record format the renderer walks, 64 rays per wave) in several ORDERS:
  tile      the renderer's own: a wave holds the rays of one 8x8-pixel tile (origins coherent, directions not)
  octant    one global queue per direction octant, in arrival order (what 8 per-octant queues would hand a wave)
  octant_4k the same inside blocks of 4,096 rays (queues local to a few waves: origins stay close)
  octant_morton  per octant, origins in Morton order (the best a sort could do)
  random    no coherence at all
Per order: kernel time (HIP events), node and triangle tests per ray and lanes per box pass (counters build).  Coherent camera rays are the
reference point.  Run on the GPU box; prints a table and writes JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
import numpy as np  # noqa: E402
import hanamaru_amd as ha  # noqa: E402


def morton2(x, y):
    def spread(v):
        v = v.astype(np.uint64) & 0xffff
        v = (v | (v << 8)) & 0x00ff00ff
        v = (v | (v << 4)) & 0x0f0f0f0f
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return spread(x) | (spread(y) << 1)


def measure(r, rays, reps=3):
    r.set_option("counters", 0)
    r.debug_trace(rays[:64])                       # warm
    s0 = r.stats()
    for _ in range(reps):
        out = r.debug_trace(rays)
    s1 = r.stats()
    ms = (s1["debug_kernel_ms"] - s0["debug_kernel_ms"]) / reps
    r.set_option("counters", 1)
    r.clear()
    r.debug_trace(rays)
    c = r.stats()
    r.set_option("counters", 0)
    r.clear()
    n = max(1, c["rays"])
    return out, {"ms": round(ms, 4), "Mrays_per_s": round(rays.shape[0] / ms / 1e3, 1), "node_tests_per_ray": round(c["node_tests"] / n, 2),
                 "tri_tests_per_ray": round(c["tri_tests"] / n, 2), "lanes_per_box_pass": round(c["box_lanes"] / max(1, c["box_passes"]), 1),
                 "lanes_per_leaf_call": round(c["leaf_lanes"] / max(1, c["leaf_calls"]), 1), "box_passes_per_ray": round(c["box_passes"] / n, 3)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "rtcamp6_v3_1"
    builder = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    W, H = 1920, 1080
    sc = ha.Scene(name)
    r = ha.Renderer(0)
    r.set_option("bvh_builder", builder)
    r.upload_scene(sc)
    r.set_resolution(64, 36)
    cam = sc.desc.camera
    v = lambda a: np.array([a.x, a.y, a.z])
    # pinhole rays through the pixel centres (camera.rs:98-107), in 8x8 tiles: tile-major, row-major inside a tile
    ty, tx, py, px = np.meshgrid(np.arange(H // 8), np.arange(W // 8), np.arange(8), np.arange(8), indexing="ij")
    X = (tx * 8 + px).ravel().astype(np.float64)
    Y = (ty * 8 + py).ravel().astype(np.float64)
    m = float(min(W, H))
    ncx, ncy = ((X + 0.5) * 2.0 - W) / m, ((H - Y - 0.5) * 2.0 - H) / m
    d = ncx[:, None] * v(cam.plane_half_right) + ncy[:, None] * v(cam.plane_half_up) + cam.focus_distance * v(cam.forward)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    prim = np.concatenate([np.broadcast_to(v(cam.eye), d.shape), d], axis=1).astype(np.float32)
    res = {"scene": name, "bvh_builder": builder, "rays": int(prim.shape[0]), "orders": {}}
    (hit, _), res["orders"]["camera_rays_tile_order"] = measure(r, prim)
    ok = hit[:, 0] == 1
    pos, nrm = hit[ok, 2:5].astype(np.float64), hit[ok, 5:8].astype(np.float64)
    rng = np.random.default_rng(7)
    n = pos.shape[0]
    # the normal may face away from the viewer (two-sided triangles): bounce on the viewer's side
    flip = np.einsum("ij,ij->i", nrm, prim[ok, 3:6].astype(np.float64)) > 0
    nrm[flip] *= -1
    r0, r1 = rng.random(n), rng.random(n)
    up = np.where(np.abs(nrm[:, :1]) > 1e-4, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    t = np.cross(up, nrm)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(nrm, t)
    ph = 2 * np.pi * r0
    bd = (t * np.cos(ph)[:, None] + b * np.sin(ph)[:, None]) * np.sqrt(r1)[:, None] + nrm * np.sqrt(1 - r1)[:, None]
    bounce = np.concatenate([pos + nrm * 1e-4, bd], axis=1).astype(np.float32)
    octant = (bounce[:, 3] < 0).astype(np.int64) | ((bounce[:, 4] < 0).astype(np.int64) << 1) | ((bounce[:, 5] < 0).astype(np.int64) << 2)
    pix_x, pix_y = X[ok].astype(np.int64), Y[ok].astype(np.int64)
    orders = {
        "tile": np.arange(n),
        "octant": np.argsort(octant, kind="stable"),
        "octant_4k": np.argsort(octant + 8 * (np.arange(n) // 4096), kind="stable"),
        "octant_morton": np.lexsort((morton2(pix_x, pix_y), octant)),
        "random": rng.permutation(n),
    }
    ref = None
    for k, idx in orders.items():
        (h, _), res["orders"]["bounce_" + k] = measure(r, bounce[idx])
        inv = np.empty(n, dtype=np.int64)
        inv[idx] = np.arange(n)
        h = h[inv]
        if ref is None:
            ref = h
        assert np.array_equal(ref.view(np.uint32), h.view(np.uint32)), k          # the order of the rays changes nothing about their hits
    res["bounce_rays"] = int(n)
    for k, d_ in res["orders"].items():
        print("%-26s %s" % (k, json.dumps(d_)))
    out = sys.argv[3] if len(sys.argv) > 3 else None
    if out:
        json.dump(res, open(out, "w"), indent=1)
    r.close()


if __name__ == "__main__":
    main()
