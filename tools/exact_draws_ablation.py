"""Ablation on the host emulation (CPU only, test infrastructure): what the records' twin of draw residuals (isaac_core.h draw_lo_f32,
RenderParams::rec_lo_off) buys precise shading.  Per-path accounting against the oracle (tests/path_parity.py): fp32 shading, precise
shading on the fp32 draws alone, precise shading as the library renders it (fp32 draw + residual = the reference's f64 draw).

    python tools/exact_draws_ablation.py [--size 192 108] [--samplings 1 2] [--threads 1] > profiles/r06_exact_draws_ablation.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("hanamaru-renderer_amd/python", "oracle", "tests", "tests/emu"):
    sys.path.insert(0, os.path.join(ROOT, p))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=2, default=[192, 108])
    ap.add_argument("--samplings", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--scenes", nargs="+", default=["spheres", "rtcamp6_v2", "tbf3", "rtcamp5", "material_examples", "rtcamp6_v3_1"])
    a = ap.parse_args()
    import hanamaru_amd as ha
    import oracle_py as orc
    import emu_py as emu
    import path_parity
    w, h = a.size
    print("host emulation, %dx%d, per-path accounting against the oracle; columns: divergent ppm | same-branch beyond 1e-3 ppm | beyond 1e-4 ppm | worst same-branch path" % (w, h))
    for name in a.scenes:
        sc = ha.Scene(name)
        o = orc.OracleScene(sc.desc_ptr)
        e = emu.EmuScene(sc.desc_ptr)
        for s in a.samplings:
            ref = o.path_log(w, h, s)
            rows = []
            for label, lo in (("precise shading, fp32 draws alone", False), ("precise shading (draws with their residuals)", True)):
                emu.set_draw_residuals(lo)
                try:
                    acc = path_parity.account(e.path_log_wf(w, h, s, a.threads), ref)
                finally:
                    emu.set_draw_residuals(True)
                sb = acc["same_branch"]
                rows.append("%-44s %6.0f | %6.0f | %7.0f | %.3g" % (label, acc["divergent_ppm"], sb["over_1e-3_floor1_ppm"], sb["over_1e-4_floor1_ppm"], sb["max_rel_floor1"]))
            fp = path_parity.account(e.path_log(w, h, s, a.threads), ref)
            sb = fp["same_branch"]
            print("%s, sampling %d" % (name, s))
            print("  %-44s %6.0f | %6.0f | %7.0f | %.3g" % ("fp32 shading", fp["divergent_ppm"], sb["over_1e-3_floor1_ppm"], sb["over_1e-4_floor1_ppm"], sb["max_rel_floor1"]))
            for r in rows:
                print("  " + r)
            sys.stdout.flush()


if __name__ == "__main__":
    main()
