"""GPU box: find the path behind a non-finite accumulator channel (scene, samplings 1 .. S at 1920x1080, precise shading on): bisect over samplings,
then the per-path log of the sampling.   python tools/nan_probe2.py <scene> [S]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
import hanamaru_amd as ha
name = sys.argv[1]; S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
r = ha.Renderer(0); r.upload_scene(ha.Scene(name)); r.set_resolution(1920, 1080); r.set_option("precise_shading", 1)
lo, hi = 1, S + 1
def bad_in(b, e):
    r.clear(); r.render(b, e); acc = r.read_accumulator()
    return np.argwhere(~np.isfinite(acc))
bad = bad_in(lo, hi)
print("non-finite channels in samplings %d..%d: %s" % (lo, hi - 1, bad[:6].tolist()), flush=True)
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
        print(" sub", sub, "radiance", g[0][y, x, sub].tolist(), "rays", int(g[1][y, x, sub]), "events", g[2][y, x, sub].tolist(), "hash", int(g[3][y, x, sub]))
    for form in (0, 1):
        r.set_debug_option("trace_mode", form)
        print("trace_mode", form, "non-finite:", len(bad_in(s, s + 1)))
    r.set_debug_option("trace_mode", -1)
    r.set_debug_option("draw_residuals", 0)
    print("draw_residuals 0 non-finite:", len(bad_in(s, s + 1)))
