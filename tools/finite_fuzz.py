"""GPU box: random scenes (tests/random_scenes.py, the campaign's four size classes) rendered LONG — 640x360 x S samplings in both shading modes —
for non-finite accumulator channels and for the two modes' image means.   python tools/finite_fuzz.py [first_seed [count [S]]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("hanamaru-renderer_amd/python", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import hanamaru_amd as ha  # noqa: E402
import random_scenes  # noqa: E402

FIRST = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
COUNT = int(sys.argv[2]) if len(sys.argv) > 2 else 100
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
r = ha.Renderer(0)
worst, bad_total = 0.0, 0
for seed in range(FIRST, FIRST + COUNT):
    kw = {}
    if seed % 4 == 1: kw = dict(spheres=40, cuboids=10, meshes=3)
    if seed % 4 == 2: kw = dict(spheres=2, cuboids=1, meshes=1)
    if seed % 4 == 3: kw = dict(spheres=0, cuboids=6, meshes=2)
    sc = random_scenes.build(ha, seed, **kw)
    r.set_option("bvh_builder", seed % 3)
    r.upload_scene(sc)
    r.set_resolution(640, 360)
    means, bad = {}, {}
    for prec in (0, 1):
        r.set_option("precise_shading", prec)
        r.clear()
        r.render(1, S + 1)
        acc = r.read_accumulator()
        bad[prec] = int((~np.isfinite(acc)).sum())
        means[prec] = float(np.nan_to_num(acc.astype(np.float64)).mean()) / S
    r.set_option("precise_shading", -1)
    rel = abs(means[1] - means[0]) / max(means[0], 1e-12)
    worst = max(worst, rel)
    bad_total += bad[0] + bad[1]
    print("seed %d builder %d %s: non-finite channels fp32 %d precise %d, image means %.6f / %.6f (rel %.1e)" % (seed, seed % 3, kw, bad[0], bad[1], means[0], means[1], rel), flush=True)
print("%d scenes x %d samplings x 2 modes: %d non-finite channels, means differ by at most %.1e" % (COUNT, S, bad_total, worst))
