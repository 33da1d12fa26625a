"""GPU box: per-path parity of the default pipeline (megakernel, fp32 shading) and of option precise_shading (split pipeline, f64 bounce
geometry) against the oracle, sampling by sampling: divergent paths, same-branch paths beyond 1e-3 / 1e-4, worst same-branch path.
    python tools/ab/precise_check.py [--scenes a,b] [--width 480 --height 270] [--samplings 1,2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("hanamaru-renderer_amd/python", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import hanamaru_amd as ha  # noqa: E402
import oracle_py as orc  # noqa: E402
import path_parity  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="rtcamp6_v2,spheres,tbf3,rtcamp5,rtcamp6_v3_1,material_examples,rtcamp6_v1,cornell_mini,simple")
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=270)
    ap.add_argument("--samplings", default="1,2")
    a = ap.parse_args()
    r = ha.Renderer(0)
    for name in a.scenes.split(","):
        sc = ha.Scene(name)
        o = orc.OracleScene(sc.desc_ptr)
        r.upload_scene(sc)
        r.set_resolution(a.width, a.height)
        for smp in [int(x) for x in a.samplings.split(",")]:
            ref = o.path_log(a.width, a.height, smp)
            for label, prec in (("default", 0), ("precise", 1)):
                r.set_option("precise_shading", prec)
                g = r.debug_path_log(smp)
                acc = path_parity.account(g, ref)
                sb = acc["same_branch"]
                print("%-18s s%-3d %s  divergent %7.1f ppm   same-branch > 1e-3: %7.1f ppm   > 1e-4: %8.1f ppm   worst %.3g   rays equal %s   mean %.6f (oracle %.6f)"
                      % (name, smp, label, acc["divergent_ppm"], sb["over_1e-3_floor1_ppm"], sb["over_1e-4_floor1_ppm"], sb["max_rel_floor1"], sb["rays_equal"],
                         acc["mean_radiance"]["gpu"], acc["mean_radiance"]["oracle"]), flush=True)
            r.set_option("precise_shading", 0)


if __name__ == "__main__":
    main()
