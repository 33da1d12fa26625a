// Precise shading (option precise_shading): the geometry of one bounce in the reference's own precision — shared by the megakernel's
// path_advance<.., PREC> (pt_core.h) and the split pipeline's shading kernel (wf_core.h wf_surface_f64).  Included by pt_core.h, after the fp32
// helpers it builds on.  __host__ __device__: the host emulation compiles the same code.
#pragma once

namespace hr {

// The walk stays fp32 — it only has to find the right primitive —, but the ray the path really follows is carried as fp32 + residual
// (o + o_lo, d + d_lo: what rounding the f64 ray to the fp32 ray the traversal walks took away), and everything between "this primitive was
// hit" and "the next ray" is f64: the hit distance again from the f64 ray and the f64 primitive (triangle plane from the f64 vertices:
// Scene::tri_exact; sphere centre and radius: sphere_lo; cuboid bounds: cuboid_lo), hit point, normal, mirror / Snell / Fresnel
// (material.rs:154-199, vector.rs:60-71), the directions sampled from the draws (diffuse lobe, GGX half vector; own f64 sincos) and the GGX
// reflectance scalar — from the reference's f64 DRAWS (the record's fp32 value + its residual from the record's twin, device_scene.h) and, where
// a roughness map drives the lobe, from the map read at f64 texture coordinates.  A faceted glass body is a billiard and a small sphere multiplies a
// position error by 2 t / r per bounce: fp32 ray state leaves 100 - 1,000 ppm of the paths that took the reference's branches off by more than
// 1e-3 in such scenes, this leaves 0 - 6.  What stays fp32: the walk, albedo / emission lookups, the sky lookup, NEE (sample
// point, weight), radiometry.
struct D3 { double x, y, z; };
HD D3 dv(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
HD D3 operator+(D3 a, D3 b) { return dv(a.x + b.x, a.y + b.y, a.z + b.z); }
HD D3 operator-(D3 a, D3 b) { return dv(a.x - b.x, a.y - b.y, a.z - b.z); }
HD D3 operator*(D3 a, double s) { return dv(a.x * s, a.y * s, a.z * s); }
HD D3 operator-(D3 a) { return dv(-a.x, -a.y, -a.z); }
HD double ddot(D3 a, D3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
HD D3 widen(V3f hi, V3f lo) { return dv((double)hi.x + (double)lo.x, (double)hi.y + (double)lo.y, (double)hi.z + (double)lo.z); }
HD V3f narrow(D3 a) { return v3((float)a.x, (float)a.y, (float)a.z); }
HD V3f residual(D3 a, V3f hi) { return v3((float)(a.x - (double)hi.x), (float)(a.y - (double)hi.y), (float)(a.z - (double)hi.z)); }
HD D3 dreflect(D3 v, D3 n) { return v - n * (2.0 * ddot(v, n)); }   // vector.rs:60-62

// material.rs:154-199 in f64.  `in` = direction of the arriving ray.  Returns the new ray and the reflectance scalar; transmitted as in pt_core.h.
HD void sample_refraction_f64(double r0, D3 pos, D3 in, D3 n, double ior, D3 &no, D3 &nd, float &refl, bool &transmitted) {
    transmitted = false;
    const bool incoming = signbit(ddot(in, n));
    const D3 on = incoming ? n : -n;
    const double nnt = incoming ? 1.0 / ior : ior;
    const D3 rdir = dreflect(in, on);
    const double vn = ddot(in, on);
    const double k = 1.0 - nnt * nnt * (1.0 - vn * vn);      // vector.rs:64-71
    if (k < 0.0) { no = pos + on * (double)OFFSET_F; nd = rdir; refl = 1.0f; return; }
    const D3 tdir = in * nnt - on * (nnt * vn + sqrt(k));
    if (tdir.x == 0.0 && tdir.y == 0.0 && tdir.z == 0.0) { no = pos + on * (double)OFFSET_F; nd = rdir; refl = 1.0f; return; }
    const double cos_i = -ddot(in, on), cos_t = -ddot(tdir, on);
    const double a = nnt * cos_i - cos_t, b = nnt * cos_i + cos_t, c = nnt * cos_t - cos_i, d = nnt * cos_t + cos_i;
    const double fr = 0.5 * (a * a / (b * b) + c * c / (d * d));
    if (r0 <= fr) { no = pos + on * (double)OFFSET_F; nd = rdir; refl = 1.0f; }
    else { no = pos - on * (double)OFFSET_F; nd = tdir; refl = (float)(nnt * nnt); transmitted = true; }
}

// sin / cos of 2 pi r for r in [0, 1), to f64 accuracy: quadrant + Taylor polynomials on [-pi/4, pi/4] (no f64 transcendental hardware;
// the fp32 v_sin_f32 / v_cos_f32 the megakernel uses are off by up to ~1e-6 — more than the rounding of the draw itself).  The same code
// on the host: the emulation and the kernels agree.
HD void sincos_2pi_f64(double r, double &sn, double &cs) {
    const double q = floor(r * 4.0 + 0.5);
    const double x = (r - q * 0.25) * 6.28318530717958647692, x2 = x * x;
    const double sx = x * fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, -1.0 / 1307674368000.0, 1.0 / 6227020800.0), -1.0 / 39916800.0), 1.0 / 362880.0), -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
    const double cx = fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, 1.0 / 20922789888000.0, -1.0 / 87178291200.0), 1.0 / 479001600.0), -1.0 / 3628800.0), 1.0 / 40320.0), -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
    const int k = (int)q & 3;
    sn = k == 0 ? sx : k == 1 ? cx : k == 2 ? -sx : -cx;
    cs = k == 0 ? cx : k == 1 ? -sx : k == 2 ? -cx : sx;
}
HD D3 dcross(D3 a, D3 b) { return dv(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HD void tangent_basis_f64(D3 n, D3 &t, D3 &b) {  // material.rs:202-211
    const D3 up = fabs(n.x) > 1e-4 ? dv(0, 1, 0) : dv(1, 0, 0);
    const D3 c = dcross(up, n);
    t = c * (1.0 / sqrt(ddot(c, c)));
    b = dcross(n, t);
}
HD D3 sample_diffuse_f64(double r0, double r1, D3 n) {  // material.rs:227-248
    D3 t, b;
    tangent_basis_f64(n, t, b);
    double sn, cs;
    sincos_2pi_f64(r0, sn, cs);
    return (t * cs + b * sn) * sqrt(r1) + n * sqrt(1.0 - r1);
}
HD D3 sample_ggx_half_f64(double r0, double r1, D3 n, double alpha2) {  // material.rs:260-269
    D3 t, b;
    tangent_basis_f64(n, t, b);
    double sn, cs;
    sincos_2pi_f64(r0, sn, cs);
    // (for a roughness of ~1e-8 — a dark texel of a roughness map — the quotient rounds to 1 + 2^-52 and 1 - cos^2 to -4e-16: the reference's own
    // expression, material.rs:264-265, is a NaN there; clamped, as the fp32 form clamps)
    const double cos_theta = sqrt(fmin((1.0 - r1) / (1.0 + (alpha2 - 1.0) * r1), 1.0));
    const double sin_theta = sqrt(fmax(1.0 - cos_theta * cos_theta, 0.0));
    return t * (sin_theta * cs) + b * (sin_theta * sn) + n * cos_theta;
}

// What shading a hit in f64 produces: the material at the hit, the fp32 normal (for NEE and the log), the next ray in f64, the sampled bounce's scalar.
struct PrecHit { PointMat m; Surf s; V3f nf; D3 no, nd; float cur_refl; bool sampled, transmitted; };
// A ROUGHNESS map is the one texture whose value becomes geometry (the GGX lobe): it is read at f64 texture coordinates — the texel quad and the
// bilinear WEIGHTS from the f64 coordinate; a 1024^2 map over a floor turns an fp32 coordinate's 6e-8 into 1e-4 texel, the lobe turns with it and
// the HDR sky multiplies that (NOTES G, P) —, the blend and the gamma curve stay fp32 (1e-7 of a smooth function).  Albedo and emission maps stay
// fp32 lookups: a 1e-4-texel shift there is a 1e-5 change of a colour.  The value is computed inside the primitive's branch and ONE float leaves it:
// with the coordinates kept in f64 across the material fetch the megakernel form rendered scenes without a single texture 25 % slower (register
// pressure), as device functions that are not inlined 40 % (the calls' spills) — NOTES P.
HD float roughness_map_f64(const Scene &sc, int32_t image, double u, double v) {   // texture.rs:29-114, channel x
    const ImageRef im = sc.images[image];
    const double x = u * (double)im.width, y = v * (double)im.height;
    const double x1 = floor(x), y1 = floor(y), x2 = x1 + 1.0, y2 = y1 + 1.0;
    const uint32_t ix1 = f32_as_u32_sat((float)x1), ix2 = f32_as_u32_sat((float)x2), iy1 = f32_as_u32_sat((float)y1), iy2 = f32_as_u32_sat((float)y2);
    const float w11 = (float)((x2 - x) * (y2 - y)), w21 = (float)((x - x1) * (y2 - y)), w12 = (float)((x2 - x) * (y - y1)), w22 = (float)((x - x1) * (y - y1));
    const float g = texel(sc, im, ix1, iy1).x * w11 + texel(sc, im, ix2, iy1).x * w21 + texel(sc, im, ix1, iy2).x * w12 + texel(sc, im, ix2, iy2).x * w22;
    return gamma_to_linear(g);
}
// a sphere's texture coordinates (scene.rs:67-71) from the f64 normal, and the roughness map there
HD float roughness_on_sphere_f64(const Scene &sc, int32_t image, double nx, double ny, double nz) {
    const double PI_D = 3.14159265358979323846;
    const double v = 1.0 - acos(fmin(fmax(ny, -1.0), 1.0)) * (1.0 / PI_D);
    const double u = 0.5 - (signbit(nz) ? -1.0 : 1.0) * acos(fmin(fmax(nx / sqrt(nx * nx + nz * nz), -1.0), 1.0)) * (0.5 / PI_D);
    return roughness_map_f64(sc, image, u, v);
}

// (ro, rd) + (fo, fd) = the ray the path follows (fp32 + residual); ts = what the fp32 walk found (a hit); r0, r1 = the iteration's draws
// (the record's fp32 values + their residuals when the launch carries them: the reference's f64 draws).
// material.rs:91-151 with the geometry of scene.rs:58-78,152-183 / bvh.rs:266-290 recomputed from the f64 ray and the f64 primitive.
HD void shade_hit_f64(const Scene &sc, V3f ro, V3f rd, V3f fo, V3f fd, const TraceState &ts, double r0, double r1, PrecHit &x) {
    const D3 o = widen(ro, fo), d = widen(rd, fd);
    D3 pos, n;
    Surf &s = x.s;
    s.u = ts.u; s.v = ts.v;
    float rough64 = -1.0f;      // the roughness map's value at the f64 coordinates (where the material has one)
    const bool want_uv = material_needs_uv(sc, hit_element(sc, ts));
    if (ts.type == 0) {            // bvh.rs:266-290: the plane of the f64 triangle
        s.elem = sc.tri_shade[ts.prim].element;
        const TriX tx = sc.tri_exact[sc.tri_face[ts.prim]];
        n = dv(tx.n[0], tx.n[1], tx.n[2]);
        const D3 v0 = dv(tx.v0[0], tx.v0[1], tx.v0[2]);
        const double t = -ddot(n, o - v0) / ddot(n, d);
        pos = o + d * t;
    } else if (ts.type == 1) {     // scene.rs:58-66, the root again from the f64 ray and the f64 sphere
        const f4 sp = sc.spheres[ts.prim], lo = sc.sphere_lo[ts.prim];
        s.elem = sc.sphere_elem[ts.prim];
        const D3 c = dv((double)sp.x + (double)lo.x, (double)sp.y + (double)lo.y, (double)sp.z + (double)lo.z);
        const double cr = (double)sp.w + (double)lo.w;
        const D3 a = o - c;
        const double idd = 1.0 / ddot(d, d);     // (a sampled direction is a unit vector to fp32 only)
        const double bq = ddot(a, d) * idd;
        const D3 q = a - d * bq;
        const double disc = (cr * cr - ddot(q, q)) * idd;
        const double t = -bq - sqrt(disc > 0.0 ? disc : 0.0);
        const D3 nn = a + d * t;
        n = nn * (1.0 / sqrt(ddot(nn, nn)));
        pos = c + nn;
        if (want_uv) {  // scene.rs:67-71
            const V3f nf = narrow(n);
            s.v = 1.0f - acosf(fminf(fmaxf(nf.y, -1.0f), 1.0f)) * (1.0f / PI_F);
            float sg = signbit(nf.z) ? -1.0f : 1.0f;
            s.u = 0.5f - sg * acosf(fminf(fmaxf(nf.x * HR_RSQ(nf.x * nf.x + nf.z * nf.z), -1.0f), 1.0f)) * (1.0f / PI2_F);
            const int32_t rimg = sc.materials[s.elem].roughness_img;
            if (rimg >= 0) rough64 = roughness_on_sphere_f64(sc, rimg, n.x, n.y, n.z);
        }
    } else {                       // bvh.rs:20-39 + scene.rs:152-182 on the f64 box
        const f4 mnf = sc.cuboids[2 * ts.prim], mxf = sc.cuboids[2 * ts.prim + 1];
        s.elem = float_as_int(mnf.w);
        const f4 mnl = sc.cuboid_lo[2 * s.elem], mxl = sc.cuboid_lo[2 * s.elem + 1];
        const D3 mn = dv((double)mnf.x + (double)mnl.x, (double)mnf.y + (double)mnl.y, (double)mnf.z + (double)mnl.z);
        const D3 mx = dv((double)mxf.x + (double)mxl.x, (double)mxf.y + (double)mxl.y, (double)mxf.z + (double)mxl.z);
        const double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z;
        const double t1 = (mn.x - o.x) * ix, t2 = (mx.x - o.x) * ix, t3 = (mn.y - o.y) * iy, t4 = (mx.y - o.y) * iy, t5 = (mn.z - o.z) * iz, t6 = (mx.z - o.z) * iz;
        const double tmin = fmax(fmax(fmin(t1, t2), fmin(t3, t4)), fmin(t5, t6)), tmax = fmin(fmin(fmax(t1, t2), fmax(t3, t4)), fmax(t5, t6));
        // the fp32 walk decided that the box is hit; should the f64 slabs disagree at a grazing edge, the walk's distance stands
        const double dist = (tmin <= tmax && !signbit(tmax)) ? (signbit(tmin) ? tmax : tmin) : (double)ts.t;
        pos = o + d * dist;
        const double E = 1e-4;   // config.rs:7
        int face;
        if (fabs(pos.y - mx.y) < E) face = 0; else if (fabs(pos.y - mn.y) < E) face = 1; else if (fabs(pos.x - mn.x) < E) face = 2;
        else if (fabs(pos.x - mx.x) < E) face = 3; else if (fabs(pos.z - mn.z) < E) face = 4; else if (fabs(pos.z - mx.z) < E) face = 5;
        else {   // (cannot happen for a hit found in f64; the nearest face, as hit_surface)
            const double dy1 = fabs(pos.y - mx.y), dy0 = fabs(pos.y - mn.y), dx0 = fabs(pos.x - mn.x), dx1 = fabs(pos.x - mx.x), dz0 = fabs(pos.z - mn.z), dz1 = fabs(pos.z - mx.z);
            const double best = fmin(fmin(fmin(dy1, dy0), fmin(dx0, dx1)), fmin(dz0, dz1));
            face = best == dy1 ? 0 : best == dy0 ? 1 : best == dx0 ? 2 : best == dx1 ? 3 : best == dz0 ? 4 : 5;
        }
        n = face == 0 ? dv(0, 1, 0) : face == 1 ? dv(0, -1, 0) : face == 2 ? dv(-1, 0, 0) : face == 3 ? dv(1, 0, 0) : face == 4 ? dv(0, 0, -1) : dv(0, 0, 1);
        if (want_uv) {   // scene.rs:166-181: the two coordinates the face uses (two divisions, not three)
            double un, uq, vn, vq;
            if (face < 2) { un = pos.x - mn.x; uq = mx.x - mn.x; vn = pos.z - mn.z; vq = mx.z - mn.z; }
            else if (face < 4) { un = pos.z - mn.z; uq = mx.z - mn.z; vn = pos.y - mn.y; vq = mx.y - mn.y; }
            else { un = pos.x - mn.x; uq = mx.x - mn.x; vn = pos.y - mn.y; vq = mx.y - mn.y; }
            double ud = un / uq, vd = vn / vq;
            if (face < 2) vd = 1.0 - vd;
            s.u = (float)ud; s.v = (float)vd;
            const int32_t rimg = sc.materials[s.elem].roughness_img;
            if (rimg >= 0) rough64 = roughness_map_f64(sc, rimg, ud, vd);
        }
    }
    const V3f nf = narrow(n);
    x.nf = nf;
    PointMat &m = x.m;
    {   // material_at (scene.rs:389-396), the roughness map's value from above
        const Material mt = sc.materials[s.elem];
        m.surface = mt.surface; m.param = mt.param;
        m.albedo = tex_sample(sc, mt.albedo_img, v3(mt.albedo), s.u, s.v);
        m.emission = tex_sample(sc, mt.emission_img, v3(mt.emission), s.u, s.v);
        m.roughness = mt.roughness_img < 0 ? mt.roughness : (rough64 >= 0.0f ? rough64 : sample_bilinear(sc, mt.roughness_img, s.u, s.v).x) * mt.roughness;
    }
    bool &transmitted = x.transmitted, &sampled = x.sampled;
    transmitted = false; sampled = true;
    D3 &no = x.no, &nd = x.nd;
    no = dv(0, 0, 0); nd = dv(0, 0, 0);
    switch (m.surface) {       // material.rs:91-151
        case 0: no = pos + n * (double)OFFSET_F; nd = sample_diffuse_f64(r0, r1, n); x.cur_refl = 1.0f; break;
        case 1: no = pos + n * (double)OFFSET_F; nd = dreflect(d, n); x.cur_refl = 1.0f; break;
        case 2: sample_refraction_f64(r0, pos, d, n, (double)m.param, no, nd, x.cur_refl, transmitted); break;
        case 3: {
            const float alpha2 = m.roughness * m.roughness;
            const D3 hh = sample_ggx_half_f64(r0, r1, n, (double)m.roughness * (double)m.roughness);
            nd = dreflect(d, hh);
            const double lnd = ddot(nd, n);
            if (signbit(lnd)) { sampled = false; break; }
            // the reflectance scalar in f64 too (material.rs:123-141): a bounce that leaves at grazing incidence has l.n ~ 1e-20 here, (l.n)^2
            // underflows in fp32, and 1 / 0 x alpha2 = 0 (a texel of roughness 0) is a NaN the f64 reference does not have
            const double vn = -ddot(d, n), vh = -ddot(d, hh), hn = ddot(hh, n), a2 = (double)alpha2;
            const double lam_l = 0.5 * sqrt(1.0 + a2 * (1.0 / (lnd * lnd) - 1.0)) - 0.5, lam_v = 0.5 * sqrt(1.0 + a2 * (1.0 / (vn * vn) - 1.0)) - 0.5;
            const double g = 1.0 / (1.0 + lam_l + lam_v), om = 1.0 - vh, f = (double)m.param + (1.0 - (double)m.param) * (om * om * om * om * om);
            const double q = g * vh / (hn * vn);
            x.cur_refl = (float)(f * (q > 0.0 ? (q < 1.0 ? q : 1.0) : 0.0));      // saturate: a NaN counts as 0 (f64::max / min)
            no = pos + n * (double)OFFSET_F;
            break;
        }
        default: {
            const D3 hh = sample_ggx_half_f64(r0, r1, n, (double)m.roughness * (double)m.roughness);
            sample_refraction_f64(r0, pos, d, hh, (double)m.param, no, nd, x.cur_refl, transmitted);
            break;
        }
    }
}

}  // namespace hr
