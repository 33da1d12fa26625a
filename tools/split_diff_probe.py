"""GPU box: where does the fp32 split pipeline's accumulator differ from the megakernel's?  (they are the same arithmetic: any difference is a bug or a
contraction the compiler made differently)  python tools/split_diff_probe.py [scene] [samplings]"""
import sys
sys.path.insert(0, "hanamaru-renderer_amd/python")
import numpy as np, hanamaru_amd as ha
scene = sys.argv[1] if len(sys.argv) > 1 else "rtcamp6_v3_1"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
r = ha.Renderer(0); sc = ha.Scene(scene); r.upload_scene(sc); r.set_resolution(1920, 1080)
r.set_option("precise_shading", 0)
found = []
for b in range(1, S + 1, 16):
    accs = []
    for mode in (0, 1):
        r.set_debug_option("trace_mode", mode)
        r.clear(); r.render(b, min(b + 16, S + 1)); accs.append(r.read_accumulator().copy())
    d = np.argwhere(accs[0] != accs[1])
    if len(d):
        print("samplings", b, "..", b + 15, "differ at", d.tolist()[:6], accs[0][d[0][0], d[0][1]], accs[1][d[0][0], d[0][1]])
        y, x = int(d[0][0]), int(d[0][1])
        for s in range(b, min(b + 16, S + 1)):
            logs = []
            for mode in (0, 1):
                r.set_debug_option("trace_mode", mode)
                logs.append(r.debug_path_log(s))
            for k in range(4):
                same = all(np.array_equal(logs[0][i][y, x, k], logs[1][i][y, x, k]) for i in range(4))
                if not same:
                    print("  sampling", s, "sub", k, "mega", logs[0][0][y, x, k].tolist(), logs[0][1][y, x, k], logs[0][2][y, x, k].tolist(), hex(int(logs[0][3][y, x, k])))
                    print("  sampling", s, "sub", k, "split", logs[1][0][y, x, k].tolist(), logs[1][1][y, x, k], logs[1][2][y, x, k].tolist(), hex(int(logs[1][3][y, x, k])))
            # whole-image comparison of the logs of this sampling
            nd = int((logs[0][0] != logs[1][0]).any(axis=-1).sum())
            if nd:
                print("  sampling", s, ":", nd, "paths with different logged radiance")
        found.append((b, d.tolist()))
r.set_debug_option("trace_mode", -1)
print("done; chunks that differ:", len(found))
