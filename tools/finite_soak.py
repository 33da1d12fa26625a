"""GPU box: every scene at 1920x1080 x N samplings with precise shading pinned on (the mode that is the default for scenes without meshes) and,
for comparison of the image means, with fp32 shading: no non-finite accumulator channel, means within 1e-4 of each other.
    python tools/finite_soak.py [samplings] > profiles/rNN_finite_soak_all_scenes_1024.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
import hanamaru_amd as ha  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
r = ha.Renderer(0)
for name in ("spheres", "simple", "material_examples", "cornell_mini", "rtcamp6_v3_1", "rtcamp6_v3", "rtcamp5", "tbf3", "rtcamp6_dodeca", "rtcamp6_v2", "rtcamp6_v1"):
    r.upload_scene(ha.Scene(name))
    r.set_resolution(1920, 1080)
    means = {}
    for prec in (1, 0):
        r.set_option("precise_shading", prec)
        r.clear()
        t0 = time.perf_counter()
        r.render(1, S + 1)
        r.synchronize()
        dt = time.perf_counter() - t0
        acc = r.read_accumulator()
        bad = int((~np.isfinite(acc)).sum())
        means[prec] = float(acc.astype(np.float64).mean()) / S
        print("%-18s precise %d: %d samplings, %.1f Mpaths/s, non-finite channels %d, image mean %.7f, max %.4g" % (name, prec, S, 1920 * 1080 * 4 * S / dt / 1e6, bad, means[prec], float(acc.max()) / S), flush=True)
    rel = abs(means[1] - means[0]) / means[0] if means[0] == means[0] and means[1] == means[1] else float('nan')
    print("%-18s means differ by %.2e" % (name, rel), flush=True)
    r.set_option("precise_shading", -1)
