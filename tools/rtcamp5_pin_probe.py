#!/usr/bin/env python3
"""tools/rtcamp5_pin_probe.py — can the reference repository's rtcamp5.png pin the oracle the way rtcamp6_1000x4spp.png does (<= 1 LSB on crops)?

Run where /root/reference exists (the build container; CPU only).  Two questions:
  1. is there a sampling count S for which the oracle's resolve of samplings 1..S reproduces a crop of the image?  (the image's S is not
     recorded anywhere: S = 1 .. N is scanned on a 12 x 8 crop, the best S printed)
  2. region by region at a fixed S: are the differences noise (zero-mean, shrinking with S) or systematic?
Result (profiles/r05_rtcamp5_pin_probe.txt): no S stands out (best mean |diff| 7.4 LSB, flat from S = 77 to 150), and the differences are
systematic — the sky alone is 8 - 10 % brighter in today's program, the hue-generated spheres have other colours, the earth sphere another
blue: rtcamp5.png was written by an OLDER revision of the reference (other scene constants and / or post chain) than the source under
/root/reference.  It cannot pin today's arithmetic; it stays a picture-level check (correlation, tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("hanamaru-renderer_amd/python", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import hanamaru_amd as ha  # noqa: E402
import oracle_py as orc  # noqa: E402
from PIL import Image  # noqa: E402

ref = np.asarray(Image.open("/root/reference/rtcamp5.png").convert("RGB")).astype(int)
sc = ha.Scene("rtcamp5")
o = orc.OracleScene(sc.desc_ptr)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
x0, y0, rw, rh = 1640, 600, 12, 8
acc = np.zeros((rh, rw, 3))
best = []
for s in range(1, N + 1):
    acc += o.render_region(1920, 1080, x0, y0, rw, rh, s, s + 1, threads=0)
    d = np.abs(orc.resolve(acc, s).astype(int)[1:-1, 1:-1] - ref[y0 + 1:y0 + rh - 1, x0 + 1:x0 + rw - 1])
    best.append((round(float(d.mean()), 3), s, int(d.max())))
best.sort()
print("1. crop (%d, %d) on the magenta sphere, S = 1 .. %d: best (mean |diff| in LSB, S, max |diff|): %s" % (x0, y0, N, best[:6]))
S = 192
print("2. oracle at S = %d against rtcamp5.png, 16 x 10 crops (interior), mean 8-bit values per channel:" % S)
for name, (x0, y0) in {"magenta sphere": (1640, 600), "sky top": (1000, 40), "sky left": (300, 300), "red bunny": (640, 520), "green sphere": (820, 800),
                       "earth sphere": (960, 600), "floor far": (1000, 700), "blue sphere": (680, 860), "glass bunny": (1280, 560)}.items():
    rw, rh = 16, 10
    img = orc.resolve(o.render_region(1920, 1080, x0, y0, rw, rh, 1, S + 1, threads=0), S).astype(int)
    a, b = img[1:-1, 1:-1], ref[y0 + 1:y0 + rh - 1, x0 + 1:x0 + rw - 1]
    print("   %-15s oracle %s   rtcamp5.png %s   signed difference %s" % (name, a.mean(axis=(0, 1)).round(1), b.mean(axis=(0, 1)).round(1), (a - b).mean(axis=(0, 1)).round(1)))
