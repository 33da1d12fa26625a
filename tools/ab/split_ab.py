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
    ap.add_argument("--profile-only", action="store_true")
    ap.add_argument("--counters", action="store_true", help="instrumented builds of both pipelines: node visits, lanes per box pass / leaf call")
    ap.add_argument("--profile", action="store_true", help="the split pipeline kernel by kernel (hr_debug_wf_profile), one launch alone on the chip")
    a = ap.parse_args()
    r = ha.Renderer(0)
    for name in a.scenes.split(","):
        sc = ha.Scene(name)
        r.upload_scene(sc)
        r.set_resolution(a.width, a.height)
        ref = None
        if a.counters:
            for mode in (0, 1):
                r.set_debug_option("trace_mode", mode)
                if mode:
                    for kv in a.opt:
                        k, v = kv.split("=")
                        r.set_debug_option(k, float(v))
                r.set_option("counters", 1)
                r.clear()
                s0 = r.stats()
                r.render(1, 5)
                r.synchronize()
                s1 = r.stats()
                r.set_option("counters", 0)
                d = {k: s1[k] - s0[k] for k in ("paths", "rays", "node_tests", "tri_tests", "box_passes", "box_lanes", "leaf_calls", "leaf_lanes")}
                pc = [s1["phase_cycles"][i] - s0["phase_cycles"][i] for i in range(4)]
                print("  %s mode %d: rays/path %.3f  node tests/ray %.2f  tri tests/ray %.2f  lanes per box pass %.1f  per leaf call %.1f  box passes/ray %.3f  leaf calls/ray %.3f  wave cycles: shade %.3g refill %.3g box %.3g leaf %.3g"
                      % (name, mode, d["rays"] / max(1, d["paths"]), d["node_tests"] / max(1, d["rays"]), d["tri_tests"] / max(1, d["rays"]), d["box_lanes"] / max(1, d["box_passes"]),
                         d["leaf_lanes"] / max(1, d["leaf_calls"]), d["box_passes"] / max(1, d["rays"]), d["leaf_calls"] / max(1, d["rays"]), pc[0], pc[1], pc[2], pc[3]), flush=True)
            r.set_debug_option("trace_mode", 0)
        for mode in ([] if a.profile_only else [int(m) for m in a.modes.split(",")]):
            r.set_debug_option("trace_mode", 1 if mode in (1, 2) else 0)
            r.set_option("precise_shading", 1 if mode in (2, 3) else 0)       # mode 2: the split pipeline with precise shading; mode 3: the megakernel with precise shading (other bits than modes 0 / 1 by design)
            if mode:
                for kv in a.opt:
                    k, v = kv.split("=")
                    r.set_debug_option(k, float(v))
            rate, seed, trace = run(r, a.width, a.height, a.samplings)
            acc = r.read_accumulator().copy()
            _, _, alone = run(r, a.width, a.height, min(a.samplings, 32), skip=2)
            same = "-" if ref is None else (str(bool(np.array_equal(ref, acc))) + ("" if np.array_equal(ref, acc) else " (%d channels differ, max %.3g)" % (int((ref != acc).sum()), float(np.abs(ref - acc).max()))))
            if ref is None:
                ref = acc
            st = r.stats()
            print("%-16s mode %d  %8.1f Mpaths/s   seed %6.2f ms   trace %6.2f ms   trace alone %6.2f ms   same bits %s   governor: level %d, %s workgroups   %s"
                  % (name, mode, rate, seed, trace, alone, same, st["governor_level"], st["governor_budget"] or "all", " ".join(a.opt) if mode else ""), flush=True)
        if a.profile or a.profile_only:
            r.set_option("precise_shading", 1 if "2" in a.modes.split(",") else 0)
            for kv in a.opt:
                k, v = kv.split("=")
                r.set_debug_option(k, float(v))
            nk = max(1, min(64, 33177600 // (a.width * a.height * 4)))
            r.debug_wf_profile(1, nk)
            ms, cn = r.debug_wf_profile(5, nk)
            print("  %s: one launch of %d samplings alone: camera rays %.2f ms, total %.2f ms (traversal %.2f, shading %.2f)" % (name, nk, ms[0], ms.sum(), ms[1::2].sum(), ms[2::2].sum()))
            for s_ in range(1, 11):
                print("    step %2d  rays %9d  paths %9d   traverse %7.3f ms (%6.1f Mrays/s)   shade %7.3f ms (%6.1f Mpaths/s)"
                      % (s_, cn[s_, 0], cn[s_, 1], ms[2 * s_ - 1], cn[s_, 0] / max(ms[2 * s_ - 1], 1e-6) / 1e3, ms[2 * s_], cn[s_, 1] / max(ms[2 * s_], 1e-6) / 1e3), flush=True)
        r.set_debug_option("trace_mode", 0)
        r.set_option("precise_shading", 0)


if __name__ == "__main__":
    main()
