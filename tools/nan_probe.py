import sys, os
sys.path.insert(0, "hanamaru-renderer_amd/python")
import numpy as np, hanamaru_amd as ha
r = ha.Renderer(0); sc = ha.Scene("rtcamp5"); r.upload_scene(sc); r.set_resolution(1920, 1080)
for prec in (0, 1):
    r.set_option("precise_shading", prec)
    r.clear(); r.render(1, 193); acc = r.read_accumulator()
    bad = np.argwhere(~np.isfinite(acc))
    print("precise", prec, "non-finite channels", len(bad), bad[:6].tolist())
    if len(bad):
        y, x = int(bad[0][0]), int(bad[0][1])
        for s in range(1, 193):
            g = r.debug_path_log(s)
            rad = g[0][y, x]
            if not np.isfinite(rad).all():
                print("sampling", s, "pixel", x, y, "radiance", rad.tolist(), "rays", g[1][y, x].tolist(), "events", g[2][y, x].tolist(), "hash", g[3][y, x].tolist())
                break
