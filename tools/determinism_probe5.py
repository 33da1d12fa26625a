import sys
sys.path.insert(0, "hanamaru-renderer_amd/python")
import numpy as np, hanamaru_amd as ha
r = ha.Renderer(0); sc = ha.Scene("rtcamp6_v3_1"); r.upload_scene(sc); r.set_resolution(1920, 1080)
r.set_option("precise_shading", 0)
X, Y, S = 743, 339, 39
logs = []
for rep in range(12):
    g = r.debug_path_log(S)
    logs.append((tuple(g[0][Y, X, 2].tolist()), int(g[1][Y, X, 2]), tuple(g[2][Y, X, 2][:6].tolist()), hex(int(g[3][Y, X, 2]))))
for k in sorted(set(logs)):
    print(logs.count(k), k)
if len(set(logs)) < 2:
    print("this box does not show it"); sys.exit(0)
dr = [r.debug_path_draws(S)[Y, X, 2].copy() for _ in range(12)] if hasattr(r, "debug_path_draws") else []
if dr:
    print("draws identical over 12 seedings:", all(np.array_equal(dr[0], d) for d in dr), dr[0].tolist())
# counters build and other instantiations
for opt in (("min_waves", 4), ("min_waves", 6), ("node_unroll", 1), ("adv_den", 1), ("leaf_den", 1), ("leaf_den", 64)):
    r.set_debug_option(opt[0], opt[1])
    vals = []
    for rep in range(10):
        r.clear(); r.render(S, S + 1); vals.append(tuple(r.read_accumulator()[Y, X].tolist()))
    print(opt, "distinct outcomes of the pixel over 10 renders:", len(set(vals)))
    r.set_debug_option(opt[0], {"min_waves": 5, "node_unroll": 2, "adv_den": 2, "leaf_den": 2}[opt[0]])
