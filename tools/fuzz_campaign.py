"""Fuzz campaign (GPU box): 80 random scenes of tests/random_scenes.py in four size classes, host- and device-built trees in turn, every
path of one 320x180 sampling compared with the oracle's (tests/path_parity.py).  One line per scene; look for rays_equal False, means that
differ, or ppm figures far from their neighbours.  Round 5: on every scene the PRODUCTION kernel also renders two samplings with and without
nee_setup's shortcuts (debug option nee_cull 7 / 0: shadow rays known to add nothing are not traced / every shadow ray traced) — the two
accumulators must be the same bit for bit (`culls identical`), random emitters of random radii are where a marginal case would hide.
python tools/fuzz_campaign.py [first_seed [count [precise]]] > profiles/rNN_fuzz_campaign.txt      (precise 1: option precise_shading pinned on)"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("hanamaru-renderer_amd/python", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
import hanamaru_amd as ha, oracle_py as orc, path_parity, random_scenes
r = ha.Renderer(0)
FIRST = int(sys.argv[1]) if len(sys.argv) > 1 else 100
COUNT = int(sys.argv[2]) if len(sys.argv) > 2 else 80
PRECISE = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # 0: fp32 shading pinned (as rounds 4, 5 ran it); 1: precise shading
for seed in range(FIRST, FIRST + COUNT):
    kw = {}
    if seed % 4 == 1: kw = dict(spheres=40, cuboids=10, meshes=3)
    if seed % 4 == 2: kw = dict(spheres=2, cuboids=1, meshes=1)
    if seed % 4 == 3: kw = dict(spheres=0, cuboids=6, meshes=2)
    try:
        sc = random_scenes.build(ha, seed, **kw)
    except Exception as e:
        print("seed", seed, "build failed", e); continue
    o = orc.OracleScene(sc.desc_ptr)
    r.set_option("bvh_builder", seed % 3)
    r.upload_scene(sc)
    r.set_option("precise_shading", PRECISE)
    w, h = 320, 180
    r.set_resolution(w, h)
    a = path_parity.account(r.debug_path_log(1), o.path_log(w, h, 1))
    sb = a["same_branch"]
    m = a["mean_radiance"]
    acc = {}
    for cull in (7, 0):
        r.set_debug_option("nee_cull", cull)
        r.set_option("counters", 1)
        r.clear()
        r.render(1, 3)
        st = r.stats()
        r.set_option("counters", 0)
        acc[cull] = (r.read_accumulator().copy(), st)
    r.set_debug_option("nee_cull", 7)
    (ca, sa), (cb, sbb) = acc[7], acc[0]
    same = bool(ca.sum() > 0 and np.array_equal(ca, cb)) and sa["rays"] + sa["shadow_culled"] == sbb["rays"] and sbb["shadow_culled"] == 0
    print("seed %d builder %d %s: divergent %.0f ppm %s; same>1e-3 %.0f ppm (no sphere %.0f) max %.3g rays_equal %s mean %.5f/%.5f; culls identical %s (%.3f of %.3f rays per path not traced)" % (seed, seed % 3, kw, a["divergent_ppm"], a["divergent_by_class_ppm"], sb["over_1e-3_floor1_ppm"], sb["no_sphere_bounce"]["over_1e-3_floor1_ppm"], sb["max_rel_floor1"], sb["rays_equal"], m["gpu"], m["oracle"], same, sa["shadow_culled"] / max(1, sa["paths"]), sbb["rays"] / max(1, sbb["paths"])), flush=True)
