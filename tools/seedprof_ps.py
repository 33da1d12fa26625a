#!/usr/bin/env python3
"""Where the waves of the phase-shifted seed kernel (seed_mode 3) spend a group: cycle stamps around every barrier of consumer 0,
consumer 1 and producer 0 (option seed_prof = 1 | 2 | 3), seed kernel alone and next to the trace kernel.
usage: python tools/seedprof_ps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
import hanamaru_amd as ha

CONS = ["wait Ws", "window", "wait We", "round part 1", "wait (other Ws)", "round part 2", "wait (other We)", "part 3 + records"]
PROD = ["wait W0s", "window 0", "wait W0e", "slot A (ahead 1)", "wait W1s", "window 1", "wait W1e", "slot B (ahead 2)"]
sc = ha.Scene("rtcamp6_v3_1")
r = ha.Renderer(0)
r.upload_scene(sc)
W, H = 1920, 1080
r.set_resolution(W, H)
r.set_debug_option("seed_mode", 3)
for label, skip in (("seed kernel alone", 16), ("next to the trace kernel", 0)):
    for who, names in ((1, CONS), (2, CONS), (3, PROD)):
        r.set_debug_option("seed_prof", 0)
        r.set_debug_option("debug_skip", 0)
        r.render(1, 9)
        r.synchronize()
        r.clear()
        r.set_debug_option("seed_prof", who)
        r.set_debug_option("debug_skip", skip)
        r.render(1, 33)
        r.synchronize()
        st = r.stats()
        ph = st["seed_phase_cycles"]
        groups = ((W + 3) // 4) * ((H + 3) // 4) * 64 * 32 / 80.0
        ms = st["seed_kernel_ms"] / max(1, st["seed_launches"])
        print("%s, %s: seed kernel %.2f ms per launch" % (label, ["", "consumer 0", "consumer 1", "producer 0"][who], ms))
        tot = sum(ph)
        for n, v in zip(names, ph):
            print("   %-20s %9.0f cycles per group  %5.1f %%" % (n, v / groups, 100.0 * v / max(1, tot)))
        print("   %-20s %9.0f" % ("total", tot / groups))
