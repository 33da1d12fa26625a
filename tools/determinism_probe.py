"""GPU box: is a long pipelined render bit-reproducible, megakernel and split pipeline?  python tools/determinism_probe.py [samplings] [repeats]"""
import sys
sys.path.insert(0, "hanamaru-renderer_amd/python")
import numpy as np, hanamaru_amd as ha
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
r = ha.Renderer(0); sc = ha.Scene("rtcamp6_v3_1"); r.upload_scene(sc); r.set_resolution(1920, 1080)
r.set_option("precise_shading", 0)
accs = {}
for rep in range(R):
    for mode in (0, 1):
        r.set_debug_option("trace_mode", mode)
        r.clear(); r.render(1, S + 1); accs[(mode, rep)] = r.read_accumulator().copy()
ref = accs[(0, 0)]
for k, a in accs.items():
    d = np.argwhere(a != ref)
    print("mode %d run %d: %d channels differ from mode 0 run 0 %s" % (k[0], k[1], len(d), d[:3].tolist() if len(d) else ""))
r.set_debug_option("trace_mode", -1)
