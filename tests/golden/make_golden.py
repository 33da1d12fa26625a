"""Regenerates the golden fixtures from the CPU oracle (oracle/oracle.cpp).

The reference (Rust) cannot be built or run in this environment, so these vectors are outputs of the
RESTATEMENT, pinned by the known answers it reproduces (ISAAC-64 KATs, BVH shapes, path statistics).
They guard the oracle and the kernels against regressions; they are not outputs of the Rust binary.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hanamaru_amd as ha  # noqa: E402
import oracle_py as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    for name, w, h, s in [("rtcamp6_v3_1", 64, 36, 2), ("cornell_mini", 48, 32, 2), ("spheres", 48, 27, 1)]:
        sc = ha.Scene(name)
        o = orc.OracleScene(sc.desc_ptr)
        acc, cn = o.render(w, h, 1, s + 1, threads=0, counters=True)
        img = orc.resolve(acc, s)
        short = {"rtcamp6_v3_1": "rtcamp6"}.get(name, name)
        np.savez_compressed(os.path.join(OUT, "%s_%dx%d_s%d.npz" % (short, w, h, s)), acc=acc.astype(np.float32), rgb8=img,
                            counters=np.array([cn[k] for k in orc.COUNTER_FIELDS], dtype=np.uint64))
    # per-path generator outputs: first 8 next_u64 for a few (w,h,x,y,sx,sy,sampling)
    cases = [(480, 270, 0, 0, 0, 0, 1), (480, 270, 479, 269, 1, 1, 1), (1920, 1080, 960, 540, 1, 0, 1024), (3840, 2160, 17, 2000, 0, 1, 4096)]
    draws = np.stack([orc.path_draws(*c, 8) for c in cases])
    np.savez_compressed(os.path.join(OUT, "path_draws.npz"), cases=np.array(cases, dtype=np.uint32), draws=draws)
    # the reference repository's committed render of init_scene_rtcamp5 (an OUTPUT of the reference binary), downscaled: pins the
    # scene builder's generator draws + collision rejection (tests/test_host_layer.py)
    src = "/root/reference/rtcamp5.png"
    if os.path.exists(src):
        from PIL import Image
        Image.open(src).convert("RGB").resize((480, 270), Image.LANCZOS).save(os.path.join(OUT, "reference_rtcamp5_480x270.png"))


if __name__ == "__main__":
    main()
