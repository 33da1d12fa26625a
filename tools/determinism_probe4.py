import sys
sys.path.insert(0, "hanamaru-renderer_amd/python")
import numpy as np, hanamaru_amd as ha
r = ha.Renderer(0); sc = ha.Scene("rtcamp6_v3_1"); r.upload_scene(sc); r.set_resolution(1920, 1080)
r.set_option("precise_shading", 0)
X, Y = 743, 339
def outcomes(b, e, reps=3):
    vals = []
    for i in range(reps):
        for mode in (0, 1):
            r.set_debug_option("trace_mode", mode)
            r.clear(); r.render(b, e); vals.append(tuple(r.read_accumulator()[Y, X].tolist()))
    return vals
v = outcomes(1, 257, 4)
ds = sorted(set(v))
print("1..256 distinct", len(ds), ["%.9g %.9g %.9g" % d for d in ds])
if len(ds) < 2:
    print("this box does not show it"); sys.exit(0)
b, e = 1, 257
while e - b > 4:
    m = b + ((e - b) // 2 + 3) // 4 * 4
    v1 = outcomes(b, m, 5)
    if len(set(v1)) > 1: e = m; continue
    v2 = outcomes(m, e, 5)
    if len(set(v2)) > 1: b = m; continue
    print("neither half of", b, e, "alone shows it"); break
print("range", b, e)
for s in range(b, e):
    vs = outcomes(s, s + 1, 4)
    print("sampling", s, "alone distinct", len(set(vs)), sorted(set(vs)))
for s in range(b, e):
    for mode in (0, 1):
        r.set_debug_option("trace_mode", mode)
        for rep in range(3):
            g = r.debug_path_log(s)
            print("log sampling", s, "mode", mode, g[0][Y, X].tolist(), g[1][Y, X].tolist(), g[2][Y, X][:, :5].tolist())
r.set_debug_option("trace_mode", -1)
