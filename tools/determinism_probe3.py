import sys
sys.path.insert(0, "hanamaru-renderer_amd/python")
import numpy as np, hanamaru_amd as ha
r = ha.Renderer(0); sc = ha.Scene("rtcamp6_v3_1"); r.upload_scene(sc); r.set_resolution(1920, 1080)
r.set_option("precise_shading", 0)
X, Y = 743, 339
def outcomes(b, e, reps=4):
    vals = []
    for i in range(reps):
        for mode in (0, 1):
            r.set_debug_option("trace_mode", mode)
            r.clear(); r.render(b, e); vals.append((mode, tuple(r.read_accumulator()[Y, X].tolist())))
    return vals
v = outcomes(1, 257, 3)
print("1..256:", v)
for b in range(1, 257, 32):
    v = outcomes(b, b + 32, 2)
    ds = sorted(set(x[1] for x in v))
    print(b, "distinct", len(ds), ds if len(ds) > 1 else "", [x[0] for x in v if x[1] != ds[0]] if len(ds) > 1 else "")
    if len(ds) > 1:
        for s in range(b, b + 32, 4):
            v2 = outcomes(s, s + 4, 3)
            d2 = sorted(set(x[1] for x in v2))
            print("   launch", s, "distinct", len(d2), d2 if len(d2) > 1 else "")
            if len(d2) > 1:
                for q in range(s, s + 4):
                    for mode in (0, 1):
                        r.set_debug_option("trace_mode", mode)
                        g = r.debug_path_log(q)
                        print("      sampling", q, "mode", mode, g[0][Y, X].tolist(), g[1][Y, X].tolist(), g[2][Y, X][:, :4].tolist())
r.set_debug_option("trace_mode", -1)
