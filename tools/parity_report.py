#!/usr/bin/env python3
"""GPU vs. CPU-oracle parity report on BASELINE's configurations (run on the GPU box; writes JSON).

    python tools/parity_report.py gpurun_out/parity_r01.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import hanamaru_amd as ha  # noqa: E402
import oracle_py as orc  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import path_parity  # noqa: E402  (the per-path accounting shared with tests/test_gpu_parity.py)


def main():
    out = {"tolerance": "per channel |gpu - oracle| <= 1e-2 * max(1, |oracle|)", "cases": []}
    r = ha.Renderer(0)
    for name, w, h, s in [("rtcamp6_v3_1", 480, 270, 1), ("rtcamp6_v3_1", 480, 270, 8), ("spheres", 480, 270, 4), ("rtcamp6_dodeca", 480, 270, 4),
                          ("rtcamp6_v3", 320, 180, 4), ("cornell_mini", 320, 200, 8), ("material_examples", 320, 180, 4), ("rtcamp6_v1", 320, 180, 4),
                          ("rtcamp6_v2", 320, 180, 2), ("rtcamp5", 320, 180, 2), ("tbf3", 320, 180, 2)]:
        sc = ha.Scene(name)
        o = orc.OracleScene(sc.desc_ptr)
        r.upload_scene(sc)
        r.set_resolution(w, h)
        r.set_option("counters", 1)
        r.clear()
        t0 = time.perf_counter()
        r.render(1, s + 1)
        acc = r.read_accumulator().astype(np.float64)
        t_gpu = time.perf_counter() - t0
        st = r.stats()
        r.set_option("counters", 0)
        t0 = time.perf_counter()
        ref, cn = o.render(w, h, 1, s + 1, threads=0, counters=True)
        t_cpu = time.perf_counter() - t0
        rel = np.abs(acc - ref) / np.maximum(1.0, np.abs(ref))
        img_g = r.resolve(s)
        img_o = orc.resolve(ref, s)
        d8 = np.abs(img_g.astype(int) - img_o.astype(int))
        ref_rays = cn["rays_primary"] + cn["rays_bounce"] + cn["rays_shadow"]
        out["cases"].append({
            "nonfinite_channels_gpu": int((~np.isfinite(acc)).sum()), "scene": name, "width": w, "height": h, "samplings": s, "paths": int(st["paths"]),
            "channels_within_tolerance": float((rel <= 1e-2).mean()), "channels_within_1e-3": float((rel <= 1e-3).mean()),
            "mean_radiance_gpu": float(acc.mean() / (4 * s)), "mean_radiance_oracle": float(ref.mean() / (4 * s)),
            "mean_rel_diff": float(abs(acc.mean() - ref.mean()) / ref.mean()),
            "rays_gpu": int(st["rays"] + st["shadow_culled"]), "rays_gpu_traced": int(st["rays"]), "rays_oracle": int(ref_rays), "rays_rel_diff": float(abs(st["rays"] + st["shadow_culled"] - ref_rays) / ref_rays),
            "png_channels_exact": float((d8 == 0).mean()), "png_channels_within_1": float((d8 <= 1).mean()), "png_max_diff": int(d8.max()),
            "node_tests_per_ray_gpu": st["node_tests"] / max(1, st["rays"]),
            "node_tests_per_ray_reference_order": (cn["mesh_node_tests"] + cn["top_node_tests"]) / ref_rays,
            "gpu_seconds_incl_readback": round(t_gpu, 4), "oracle_seconds_all_cores": round(t_cpu, 3)})
        print(json.dumps(out["cases"][-1]))
    # per-path accounting (round 4): every path of one sampling compared with the oracle's path — same branches or not, and how far apart
    out["per_path"] = {"what": "hr_debug_path_log (the render kernel's logging instantiation) vs orc_path_log, sampling 1: a path is 'same' when its event log "
                               "(per iteration: miss / surface type hit / sample None, reflected or transmitted, NEE visibility mask) and the hash of the elements "
                               "and mesh triangles it hit equal the oracle's; relative errors are per channel against max(1, |oracle|) ('floor1') or against the "
                               "path's own magnitude ('own')", "cases": []}
    for name, w, h in [("rtcamp6_v3_1", 480, 270), ("rtcamp6_dodeca", 480, 270), ("spheres", 480, 270), ("cornell_mini", 320, 200), ("rtcamp6_v2", 320, 180),
                       ("rtcamp5", 320, 180), ("tbf3", 320, 180), ("material_examples", 320, 180), ("rtcamp6_v1", 320, 180), ("rtcamp6_v3", 320, 180), ("simple", 320, 180)]:
        sc = ha.Scene(name)
        o = orc.OracleScene(sc.desc_ptr)
        r.upload_scene(sc)
        r.set_resolution(w, h)
        a = path_parity.account(r.debug_path_log(1), o.path_log(w, h, 1))
        a.update({"scene": name, "width": w, "height": h, "sampling": 1})
        out["per_path"]["cases"].append(a)
        print(json.dumps(a))
    # the same accounting at BASELINE's image size: every path of two samplings of configs 3, 5 (at 1080p) and 2 — 8.3 M paths per case,
    # so that a ppm figure rests on thousands of paths instead of a handful
    out["per_path_full_size"] = {"what": "as per_path, every path of whole samplings at 1920x1080 (8,294,400 paths; samplings 1 and 1000) and at 7680x4320 (132,710,400 paths; sampling 2)", "cases": []}
    for name, W, H, samplings in (("rtcamp6_v3_1", 1920, 1080, (1, 1000)), ("rtcamp6_dodeca", 1920, 1080, (1, 1000)), ("spheres", 1920, 1080, (1, 1000)),
                                 ("rtcamp6_v3_1", 7680, 4320, (2,))):       # 8K: 132.7 M paths of one sampling (the oracle takes ~2.5 minutes)
        sc = ha.Scene(name)
        o = orc.OracleScene(sc.desc_ptr)
        r.upload_scene(sc)
        r.set_resolution(W, H)
        for sampling in samplings:
            t0 = time.time()
            g = r.debug_path_log(sampling)
            t1 = time.time()
            ref = o.path_log(W, H, sampling)
            t2 = time.time()
            a = path_parity.account(g, ref)
            a.update({"scene": name, "width": W, "height": H, "sampling": sampling, "gpu_seconds_incl_readback": round(t1 - t0, 3), "oracle_seconds_all_cores": round(t2 - t1, 2)})
            out["per_path_full_size"]["cases"].append(a)
            print(json.dumps(a))
            del g, ref
    json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
