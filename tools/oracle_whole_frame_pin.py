"""Whole-frame pin of the ORACLE against the reference binary's committed render (build container only; hours of CPU).

tests/golden/reference_rtcamp6_1000x4spp.png is the Rust program's own output (default scene, 1920x1080, -s 1000, README.md:19).
tests/test_oracle.py pins the oracle to it on twelve 12x8 crops; the GPU path is compared with it over the whole frame.  This script
closes the gap: the oracle (oracle/oracle.cpp, f64) renders ALL 1920 x 1080 pixels x 1000 samplings, runs its own post chain
(Reinhard, gamma, bilateral, quantise) and is compared with the reference image pixel by pixel.

Output (data only, committed): tests/golden/oracle_whole_frame_pin.npz
    rows_exact[1080], rows_within1[1080]   channels of each row (of 5760) that are identical / within 1 LSB
    diff_yxc[n, 3] (uint16), diff_val[n] (int16)   every channel where oracle - reference != 0
    acc_rows[18, 1920, 3] float64 + acc_row_index[18]   the oracle's accumulator of six row triples spread over the frame: the CPU test
                                                        re-derives segments of them and resolves the middle rows (tests/test_oracle.py)
    meta: samplings, seconds, threads
The render runs in bands of rows with a checkpoint (gpurun_out-style scratch under oracle/_pin/, git-ignored) so that it can be resumed.

    nice -n 19 python tools/oracle_whole_frame_pin.py --threads 6
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

W, H, S = 1920, 1080, 1000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--band", type=int, default=8, help="rows per checkpointed band")
    ap.add_argument("--scratch", default=os.path.join(ROOT, "oracle", "_pin"))
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "oracle_whole_frame_pin.npz"))
    a = ap.parse_args()
    import hanamaru_amd as ha
    import oracle_py as orc
    from PIL import Image

    os.makedirs(a.scratch, exist_ok=True)
    ck = os.path.join(a.scratch, "acc.npy")
    done_f = os.path.join(a.scratch, "done.npy")
    acc = np.load(ck) if os.path.exists(ck) else np.zeros((H, W, 3), dtype=np.float64)
    done = np.load(done_f) if os.path.exists(done_f) else np.zeros(H, dtype=bool)
    secs_f = os.path.join(a.scratch, "seconds.txt")
    spent = float(open(secs_f).read()) if os.path.exists(secs_f) else 0.0
    sc = ha.Scene("rtcamp6_v3_1")
    o = orc.OracleScene(sc.desc_ptr)
    t_last = time.time()
    for y0 in range(0, H, a.band):
        rh = min(a.band, H - y0)
        if done[y0:y0 + rh].all():
            continue
        acc[y0:y0 + rh] = o.render_region(W, H, 0, y0, W, rh, 1, S + 1, threads=a.threads)
        done[y0:y0 + rh] = True
        now = time.time()
        spent += now - t_last
        t_last = now
        np.save(ck + ".tmp.npy", acc)
        os.replace(ck + ".tmp.npy", ck)
        np.save(done_f + ".tmp.npy", done)
        os.replace(done_f + ".tmp.npy", done_f)
        open(secs_f, "w").write("%.1f" % spent)
        print("rows %4d..%4d done, %.0f s so far" % (y0, y0 + rh - 1, spent), flush=True)
    img = orc.resolve(acc, S).astype(np.int16)
    ref = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "reference_rtcamp6_1000x4spp.png")).convert("RGB")).astype(np.int16)
    d = img - ref
    ad = np.abs(d)
    rows_exact = (ad == 0).reshape(H, -1).sum(axis=1).astype(np.uint16)
    rows_within1 = (ad <= 1).reshape(H, -1).sum(axis=1).astype(np.uint16)
    yxc = np.argwhere(d != 0).astype(np.uint16)
    val = d[d != 0].astype(np.int16)
    # six row TRIPLES spread over the frame (sky, the mirror, the glass armadillo, the frame, the floor's lettering, the bottom edge): the
    # middle row of a triple can be resolved on its own (the 3x3 bilateral sees its neighbours), so the CPU test ties accumulator -> 8-bit row
    mid = np.array([40, 300, 520, 700, 880, 1040])
    idx = np.stack([mid - 1, mid, mid + 1], axis=1).reshape(-1)
    np.savez_compressed(a.out, rows_exact=rows_exact, rows_within1=rows_within1, diff_yxc=yxc, diff_val=val, acc_rows=acc[idx], acc_row_index=idx.astype(np.uint16),
                        meta=np.array([S, spent, a.threads], dtype=np.float64))
    mse = float((d.astype(np.float64) ** 2).mean())
    print("whole frame: %.4f %% identical, %.4f %% within 1 LSB, worst %d LSB, PSNR %.2f dB, %d differing channels, %.0f s on %d threads"
          % (100.0 * (ad == 0).mean(), 100.0 * (ad <= 1).mean(), int(ad.max()), 10.0 * np.log10(255.0 ** 2 / mse) if mse > 0 else 999.0, len(val), spent, a.threads))


if __name__ == "__main__":
    main()
