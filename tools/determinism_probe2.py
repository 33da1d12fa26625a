"""GPU box: bisect the sampling range in which pixel (743, 339) of the headline render has two outcomes."""
import sys
sys.path.insert(0, "hanamaru-renderer_amd/python")
import numpy as np, hanamaru_amd as ha
r = ha.Renderer(0); sc = ha.Scene("rtcamp6_v3_1"); r.upload_scene(sc); r.set_resolution(1920, 1080)
r.set_option("precise_shading", 0)
X, Y = 743, 339
def outcomes(b, e, reps=8):
    vals = []
    for _ in range(reps):
        r.clear(); r.render(b, e); vals.append(tuple(r.read_accumulator()[Y, X].tolist()))
    return vals
b, e = 1, 257
v = outcomes(b, e)
print("range", b, e, "distinct", len(set(v)), sorted(set(v)))
while e - b > 4 and len(set(v)) > 1:
    m = (b + e) // 2
    m = b + ((m - b + 3) // 4) * 4          # keep launch boundaries (4 samplings per launch) where they were
    v1 = outcomes(b, m)
    v2 = outcomes(m, e)
    print("  ", (b, m), len(set(v1)), (m, e), len(set(v2)))
    if len(set(v1)) > 1: e, v = m, v1
    elif len(set(v2)) > 1: b, v = m, v2
    else:
        print("  neither half alone is non-deterministic"); break
print("final range", b, e, sorted(set(v)))
# in that range: single launches, each sampling alone, and the path logs
for s in range(b, e):
    vs = outcomes(s, s + 1, 6)
    print("sampling", s, "alone: distinct", len(set(vs)))
