"""GPU box: the paths of one fuzz scene that differ most from the oracle's, with their classes (tools/fuzz_campaign.py's seeds).
python tools/fuzz_inspect.py <seed> [builder]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("hanamaru-renderer_amd/python", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
import hanamaru_amd as ha, oracle_py as orc, path_parity, random_scenes
seed = int(sys.argv[1])
builder = int(sys.argv[2]) if len(sys.argv) > 2 else seed % 3
kw = {}
if seed % 4 == 1: kw = dict(spheres=40, cuboids=10, meshes=3)
if seed % 4 == 2: kw = dict(spheres=2, cuboids=1, meshes=1)
if seed % 4 == 3: kw = dict(spheres=0, cuboids=6, meshes=2)
sc = random_scenes.build(ha, seed, **kw)
o = orc.OracleScene(sc.desc_ptr)
r = ha.Renderer(0)
r.set_option("bvh_builder", builder)
r.upload_scene(sc)
w, h = 320, 180
r.set_resolution(w, h)
g = r.debug_path_log(1)
ref = o.path_log(w, h, 1)
rg, rr = g[0].astype(np.float64), ref[0].astype(np.float64)
d = np.abs(rg - rr).sum(axis=-1)
print("seed", seed, "builder", builder, "mean gpu %.5f oracle %.5f" % (rg.mean(), rr.mean()), "sum of |diff| %.3f" % d.sum())
idx = np.dstack(np.unravel_index(np.argsort(-d, axis=None)[:8], d.shape))[0]
for (y, x, s) in idx:
    # as tests/path_parity.py account(): nine event bytes + the hash of every discrete decision make a branch; the count of sphere hits and the
    # texel-quad sum (bytes 9 - 11) are reported beside it
    eg, eo = g[2][y, x, s], ref[2][y, x, s]
    ev_same = (eg[:9] == eo[:9]).all()
    same = ev_same and g[3][y, x, s] == ref[3][y, x, s]
    if same:
        cls = "same branch" + ("" if (eg[10:12] == eo[10:12]).all() else ", another texel quad somewhere along the path")
    else:
        cls = "divergent: %s" % ("other_element_same_events" if ev_same else "%s at iteration %d" % path_parity.classify(eg[:9], eo[:9])[::-1])
    print("pixel (%d, %d) sub %d: gpu %s oracle %s rays %d / %d  events gpu %s oracle %s  %s" % (x, y, s, np.round(rg[y, x, s], 4), np.round(rr[y, x, s], 4), g[1][y, x, s], ref[1][y, x, s],
          [int(v) for v in eg], [int(v) for v in eo], cls))
