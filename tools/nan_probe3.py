"""GPU box: the path behind a non-finite channel of a RANDOM scene (tools/finite_fuzz.py): python tools/nan_probe3.py <seed> <precise> [S]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("hanamaru-renderer_amd/python", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import hanamaru_amd as ha, random_scenes
seed = int(sys.argv[1]); prec = int(sys.argv[2]); S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
kw = {}
if seed % 4 == 1: kw = dict(spheres=40, cuboids=10, meshes=3)
if seed % 4 == 2: kw = dict(spheres=2, cuboids=1, meshes=1)
if seed % 4 == 3: kw = dict(spheres=0, cuboids=6, meshes=2)
sc = random_scenes.build(ha, seed, **kw)
r = ha.Renderer(0); r.set_option("bvh_builder", seed % 3); r.upload_scene(sc); r.set_resolution(640, 360); r.set_option("precise_shading", prec)
def bad_in(b, e):
    r.clear(); r.render(b, e); acc = r.read_accumulator()
    return np.argwhere(~np.isfinite(acc))
lo, hi = 1, S + 1
bad = bad_in(lo, hi)
print("seed", seed, "precise", prec, "non-finite:", bad[:9].tolist(), flush=True)
while len(bad) and hi - lo > 1:
    mid = (lo + hi) // 2
    b1 = bad_in(lo, mid)
    if len(b1): hi, bad = mid, b1
    else: lo = mid; bad = bad_in(lo, hi)
if len(bad):
    y, x = int(bad[0][0]), int(bad[0][1]); s = lo
    g = r.debug_path_log(s)
    print("sampling", s, "pixel x", x, "y", y)
    for sub in range(4):
        print(" sub", sub, "radiance", g[0][y, x, sub].tolist(), "rays", int(g[1][y, x, sub]), "events", [hex(v) for v in g[2][y, x, sub].tolist()[:10]])
