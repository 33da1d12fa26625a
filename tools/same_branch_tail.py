#!/usr/bin/env python3
"""tools/same_branch_tail.py [scene w h samplings]...: the worst same-branch path (per channel, against max(1, |oracle|)) over several samplings — the
measurement behind PATH_LIMITS' same_max in tests/test_gpu_parity.py (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("hanamaru-renderer_amd/python", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import hanamaru_amd as ha  # noqa: E402
import oracle_py as orc  # noqa: E402
import path_parity  # noqa: E402

r = ha.Renderer(0)
cases = [("rtcamp6_v2", 192, 108, 12), ("rtcamp5", 192, 108, 12), ("tbf3", 192, 108, 8), ("spheres", 256, 144, 8), ("rtcamp6_v1", 192, 108, 8)]
for name, w, h, n in cases:
    sc = ha.Scene(name)
    o = orc.OracleScene(sc.desc_ptr)
    r.upload_scene(sc)
    r.set_resolution(w, h)
    worst, div, over, quad = 0.0, 0.0, 0.0, 0.0
    for s in range(1, n + 1):
        a = path_parity.account(r.debug_path_log(s), o.path_log(w, h, s))
        sb = a["same_branch"]
        worst = max(worst, sb["max_rel_floor1"]); div = max(div, a["divergent_ppm"]); over = max(over, sb["over_1e-3_floor1_ppm"]); quad = max(quad, sb["other_texel_quad"]["ppm"])
    print("%-12s %dx%d samplings 1..%d: worst same-branch path %.4f, most divergent %.1f ppm, most same-branch beyond 1e-3 %.1f ppm, most on another texel quad %.1f ppm" % (name, w, h, n, worst, div, over, quad))
