#!/usr/bin/env python3
"""Phase breakdown of the seed kernel's consumer waves (option seed_prof): alone and next to the trace kernel.
usage: python tools/seedprof.py [split] [seed_mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
import hanamaru_amd as ha

split = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
names = ["issue reg loads", "wait 16 regs", "init blocks", "barrier B", "round+head", "ovf note", "barrier A", "groups"]
sc = ha.Scene("rtcamp6_v3_1")
r = ha.Renderer(0)
r.upload_scene(sc)
r.set_resolution(1920, 1080)
r.set_debug_option("seed_split", split)
r.set_debug_option("seed_mode", mode)
for label, skip in (("seed kernel alone", 16), ("next to the trace kernel", 0)):
    r.set_debug_option("seed_prof", 0)
    r.set_debug_option("debug_skip", 0)
    r.render(1, 9)
    r.synchronize()
    r.clear()
    r.set_debug_option("seed_prof", 1)
    r.set_debug_option("debug_skip", skip)
    r.render(1, 33)
    r.synchronize()
    st = r.stats()
    ph = st["seed_phase_cycles"]
    groups = max(1, ph[7])
    ms = st["seed_kernel_ms"] / max(1, st["seed_launches"])
    print("seed_mode %d, split %d, %s: seed kernel %.2f ms per launch, %d consumer-wave groups" % (mode, split, label, ms, groups))
    tot = sum(ph[:7])
    for n, v in zip(names[:7], ph[:7]):
        print("   %-18s %9.0f cycles per group  %5.1f %%" % (n, v / groups, 100.0 * v / tot))
    print("   %-18s %9.0f cycles per group (100 MHz ticks if s_memtime is the constant clock)" % ("total", tot / groups))
