"""GPU box: megakernel (trace_mode 0) against the split pipeline (trace_mode 1) on one box, scene by scene.
    python tools/ab/split_ab.py [--scenes a,b] [--samplings 64] [--opt key=value ...]   (options apply to the split runs)
One line per (scene, mode): Mpaths/s of the pair, seed / trace milliseconds per launch (HIP events), the trace side alone (no seed kernel
beside it: debug_skip 2), and whether the accumulator equals the megakernel's bit for bit."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
import hanamaru_amd as ha  # noqa: E402


def run(r, W, H, S, skip=0):
    r.clear()
    r.render(1, 9)
    r.synchronize()
    r.clear()
    if skip:
        r.set_debug_option("debug_skip", skip)
    s0 = r.stats()
    t0 = time.perf_counter()
    r.render(1, S + 1)
    r.synchronize()
    dt = time.perf_counter() - t0
    s1 = r.stats()
    if skip:
        r.set_debug_option("debug_skip", 0)
    n = max(1, s1["trace_launches"] - s0["trace_launches"])
    return W * H * 4 * S / dt / 1e6, (s1["seed_kernel_ms"] - s0["seed_kernel_ms"]) / n, (s1["trace_kernel_ms"] - s0["trace_kernel_ms"]) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="rtcamp6_v3_1,rtcamp6_v2,rtcamp6_v1,rtcamp6_dodeca,tbf3,spheres")
    ap.add_argument("--samplings", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--modes", default="0,1")
    a = ap.parse_args()
    r = ha.Renderer(0)
    for name in a.scenes.split(","):
        sc = ha.Scene(name)
        r.upload_scene(sc)
        r.set_resolution(a.width, a.height)
        ref = None
        for mode in [int(m) for m in a.modes.split(",")]:
            r.set_debug_option("trace_mode", mode)
            if mode:
                for kv in a.opt:
                    k, v = kv.split("=")
                    r.set_debug_option(k, float(v))
            rate, seed, trace = run(r, a.width, a.height, a.samplings)
            acc = r.read_accumulator().copy()
            _, _, alone = run(r, a.width, a.height, min(a.samplings, 32), skip=2)
            same = "-" if ref is None else str(bool(np.array_equal(ref, acc)))
            if ref is None:
                ref = acc
            print("%-16s mode %d  %8.1f Mpaths/s   seed %6.2f ms   trace %6.2f ms   trace alone %6.2f ms   same bits %s   %s"
                  % (name, mode, rate, seed, trace, alone, same, " ".join(a.opt) if mode else ""), flush=True)
        r.set_debug_option("trace_mode", 0)


if __name__ == "__main__":
    main()
