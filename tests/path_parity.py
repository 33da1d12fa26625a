"""Per-path parity accounting: compares the event logs and radiances of the HIP path (hr_debug_path_log, or the host
emulation of the same per-lane code) with the oracle's (orc_path_log), path by path.

A path is SAME when its event log (one byte per iteration: miss / surface type hit / sample returned None, reflected or transmitted, NEE
visibility mask) and the hash of its discrete geometric decisions (the element and mesh triangle of every hit, the face of a cuboid hit, the
cube-map face of the sky lookup that ends it) equal the oracle's — it took the reference's branches; everything else is DIVERGENT and is
classified by the first iteration whose event byte differs (equal events, another hash: `other_element_same_events` — another element,
triangle, cuboid face or sky face).  The texel quads of the image lookups are logged too (a 16-bit sum) and reported, not gated.  Helper module: imported by tests and tools/parity_report.py."""
import numpy as np

SURFACES = ["diffuse", "specular", "refraction", "ggx", "ggx_refraction"]


def _kind_name(b):
    k = int(b) & 7
    return "not_reached" if k == 0 else "miss" if k == 1 else "sample_none" if k == 7 else SURFACES[k - 2]


def classify(ev_g, ev_o):
    """Why two event logs differ, from the first differing byte: (iteration 1..9, class)."""
    i = int(np.argmax(ev_g != ev_o))
    a, b = int(ev_g[i]), int(ev_o[i])
    ka, kb = a & 7, b & 7
    if ka != kb:
        if 1 in (ka, kb):
            return i + 1, "hit_vs_miss"                     # a silhouette: one side hit something, the other saw the sky
        if 7 in (ka, kb):
            return i + 1, "ggx_sample_below_horizon"        # material.rs:119-121 decided differently
        return i + 1, "other_surface_type"                  # the closest hit is an element of another material
    if (a ^ b) & 8:
        return i + 1, "reflect_vs_transmit"                 # Fresnel coin r0 <= fr, or total internal reflection (material.rs:163-199)
    return i + 1, "nee_visibility"                          # the shadow ray's proximity test of renderer.rs:280


def account(gpu, ref, floor=1.0):
    """gpu / ref: (radiance [h, w, 4, 3], rays [h, w, 4], events [h, w, 4, 10] (nine event bytes + sphere hits), hash [h, w, 4]).  Returns a dict of plain numbers."""
    rg, raysg, evg, hg = gpu
    ro, rayso, evo, ho = ref
    n = raysg.size
    rg = rg.reshape(n, 3).astype(np.float64); ro = ro.reshape(n, 3).astype(np.float64)
    evg = evg.reshape(n, 12); evo = evo.reshape(n, 12)    # nine event bytes, the count of sphere hits (equal wherever the events and the hash are), the texel-quad sum
    sph = evo[:, 9].astype(np.int64)
    quad_same = (evg[:, 10:12] == evo[:, 10:12]).all(axis=1)
    evg = evg[:, :9]; evo = evo[:, :9]
    hg = hg.reshape(n); ho = ho.reshape(n)
    ev_same = (evg == evo).all(axis=1)
    same = ev_same & (hg == ho)
    # relative error of a path's radiance: per channel, against max(floor, |reference|) — the accumulator tests' measure (floor 1.0)
    rel = np.abs(rg - ro) / np.maximum(floor, np.abs(ro))
    rel_path = rel.max(axis=1)
    # against the path's own magnitude (no floor beyond 1e-3): what fp32 does to one path
    mag = np.maximum(np.abs(ro).max(axis=1), 1e-3)
    rel_own = np.abs(rg - ro).max(axis=1) / mag
    out = {"paths": int(n), "same": int(same.sum()), "divergent": int((~same).sum()), "divergent_ppm": round(1e6 * float((~same).mean()), 2)}
    rs = rel_path[same]
    ro_ = rel_own[same]
    out["same_branch"] = {
        "max_rel_floor1": float(rs.max()) if rs.size else 0.0, "p999_rel_floor1": float(np.quantile(rs, 0.999)) if rs.size else 0.0,
        "median_rel_floor1": float(np.median(rs)) if rs.size else 0.0,
        "max_rel_own": float(ro_.max()) if ro_.size else 0.0, "p999_rel_own": float(np.quantile(ro_, 0.999)) if ro_.size else 0.0,
        "over_1e-3_floor1_ppm": round(1e6 * float((rs > 1e-3).sum()) / n, 2), "over_1e-4_floor1_ppm": round(1e6 * float((rs > 1e-4).sum()) / n, 2),
        "rays_equal": bool((raysg.reshape(n)[same] == rayso.reshape(n)[same]).all())}
    # same-branch paths off by more than 1e-3, by how many spheres the path bounced off: a sphere multiplies the position error of the ray
    # that hits it by 1 / radius (and by 1 / cos at grazing incidence) and hands it on as a direction error — the one amplifier fp32 rays meet
    over = same & (rel_path > 1e-3)
    out["same_branch"]["over_1e-3_by_sphere_bounces_ppm"] = {str(k): round(1e6 * float((over & (sph == k)).sum()) / n, 2) for k in sorted(set(sph[over].tolist()))}
    # same branch, but a texture value interpolated between OTHER texels (a lookup that fell on the other side of a quad border: bilinear
    # interpolation is continuous there, so this is not a branch — how many there are and what they cost is reported, not gated)
    oq = same & ~quad_same
    out["same_branch"]["other_texel_quad"] = {"ppm": round(1e6 * float(oq.sum()) / n, 2), "max_rel_floor1": float(rel_path[oq].max()) if oq.any() else 0.0,
                                              "over_1e-3_floor1_ppm": round(1e6 * float((oq & (rel_path > 1e-3)).sum()) / n, 2)}
    flat = same & (sph == 0)
    out["same_branch"]["no_sphere_bounce"] = {"paths": int(flat.sum()), "max_rel_floor1": float(rel_path[flat].max()) if flat.any() else 0.0,
                                              "over_1e-3_floor1_ppm": round(1e6 * float((flat & (rel_path > 1e-3)).sum()) / n, 2)}
    classes, by_iter = {}, {}
    idx = np.nonzero(~same)[0]
    for k in idx:
        if ev_same[k]:
            it, c = 0, "other_element_same_events"          # equal event bytes, another element index somewhere along the path
        else:
            it, c = classify(evg[k], evo[k])
        classes[c] = classes.get(c, 0) + 1
        by_iter[it] = by_iter.get(it, 0) + 1
    out["divergent_by_class_ppm"] = {c: round(1e6 * v / n, 2) for c, v in sorted(classes.items(), key=lambda kv: -kv[1])}
    out["divergent_by_first_iteration_ppm"] = {str(i): round(1e6 * v / n, 2) for i, v in sorted(by_iter.items())}
    rd = rel_path[~same]
    out["divergent_radiance"] = {"over_1e-3_floor1_ppm": round(1e6 * float((rd > 1e-3).sum()) / n, 2), "within_1e-3_floor1_ppm": round(1e6 * float((rd <= 1e-3).sum()) / n, 2)}
    # what this means for an S-sampling accumulator pixel: P(no divergent path among its 4 S paths)
    p = float((~same).mean())
    out["expected_clean_pixel_fraction"] = {str(s): round((1.0 - p) ** (4 * s), 6) for s in (1, 2, 4, 8, 64)}
    out["mean_radiance"] = {"gpu": float(rg.mean()), "oracle": float(ro.mean())}
    return out
