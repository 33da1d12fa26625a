// Per-path ISAAC-64 seeding for the device (rand 0.4.3 `StdRng::from_seed(&[8700304, sampling, s, t])`,
// renderer.rs:165-168; algorithm: Bob Jenkins' public-domain ISAAC-64 as arranged by `rand`,
// SURVEY.md Appendix B).  One lane = one generator.  The 2 KiB `mem` state of a lane is reached through
// a `Mem` accessor: on the GPU that is a bank-column of LDS (mem[i][lane], conflict-free gathers), in the
// host emulation a plain array.
//
// Only what a path can consume is produced: the generator hands out rsl[255], rsl[254], ... so the
// k-th draw is the output of round step 255-k.  The last ISAAC_TAIL steps go to a `Tail` sink.
#pragma once
#include "device_scene.h"

namespace hr {

typedef unsigned long long u64;

#if defined(__HIPCC__)
#define HR_NOUNROLL _Pragma("unroll 1")
#else
#define HR_NOUNROLL
#endif

#define HR_ISAAC_MIX(a, b, c, d, e, f, g, h) \
    a -= e; f ^= h >> 9;  h += a;            \
    b -= f; g ^= a << 9;  a += b;            \
    c -= g; h ^= b >> 23; b += c;            \
    d -= h; a ^= c << 15; c += d;            \
    e -= a; b ^= d >> 14; d += e;            \
    f -= b; c ^= e << 20; e += f;            \
    g -= c; d ^= f >> 17; f += g;            \
    h -= d; e ^= g << 14; g += h;

// state of the eight registers after the four warm-up mixes of 0x9e3779b97f4a7c13 (seed independent)
struct IsaacWarm { u64 r[8]; };
HD IsaacWarm isaac_warm() {
    u64 a, b, c, d, e, f, g, h;
    a = b = c = d = e = f = g = h = 0x9e3779b97f4a7c13ULL;
    for (int i = 0; i < 4; i++) { HR_ISAAC_MIX(a, b, c, d, e, f, g, h) }
    IsaacWarm w;
    w.r[0] = a; w.r[1] = b; w.r[2] = c; w.r[3] = d; w.r[4] = e; w.r[5] = f; w.r[6] = g; w.r[7] = h;
    return w;
}

// u64 -> f64 of rand 0.4.3 `Rng::next_f64`: 52 mantissa bits in [1,2) minus 1
HD double isaac_to_f64(u64 v) {
    union { u64 u; double d; } cv;
    cv.u = 0x3FF0000000000000ULL | (v & 0x000FFFFFFFFFFFFFULL);
    return cv.d - 1.0;
}

// Mem: u64 ld(int i) / void st(int i, u64 v).  Tail: void put(int step, u64 value) for step >= 256 - ISAAC_TAIL.
template <class Mem, class Tail>
HD void isaac_seed_round(Mem &mem, const IsaacWarm &w, u64 s0, u64 s1, u64 s2, u64 s3, Tail &tail) {
    u64 a = w.r[0], b = w.r[1], c = w.r[2], d = w.r[3], e = w.r[4], f = w.r[5], g = w.r[6], h = w.r[7];
    // pass 1 over rsl = seed ++ zeros
    a += s0; b += s1; c += s2; d += s3;
    HR_NOUNROLL
    for (int i = 0; i < 256; i += 8) {
        HR_ISAAC_MIX(a, b, c, d, e, f, g, h)
        mem.st(i, a); mem.st(i + 1, b); mem.st(i + 2, c); mem.st(i + 3, d);
        mem.st(i + 4, e); mem.st(i + 5, f); mem.st(i + 6, g); mem.st(i + 7, h);
    }
    // pass 2 over mem
    HR_NOUNROLL
    for (int i = 0; i < 256; i += 8) {
        a += mem.ld(i); b += mem.ld(i + 1); c += mem.ld(i + 2); d += mem.ld(i + 3);
        e += mem.ld(i + 4); f += mem.ld(i + 5); g += mem.ld(i + 6); h += mem.ld(i + 7);
        HR_ISAAC_MIX(a, b, c, d, e, f, g, h)
        mem.st(i, a); mem.st(i + 1, b); mem.st(i + 2, c); mem.st(i + 3, d);
        mem.st(i + 4, e); mem.st(i + 5, f); mem.st(i + 6, g); mem.st(i + 7, h);
    }
    // one isaac64() round: a = b = 0, c = 1  ->  aa = 0, bb = 1
    u64 aa = 0, bb = 1;
    // software-pipelined: the first gather of step i+1 is issued before the second gather of step i is consumed
    u64 x = mem.ld(0);
    u64 g1 = mem.ld((int)((x >> 3) & 255));
    for (int half = 0; half < 2; half++) {
        const int mr = half ? 128 : 0, m2 = half ? 0 : 128;
        HR_NOUNROLL
        for (int base = 0; base < 128; base += 4) {
#define HR_ISAAC_STEP(J, MIXEXPR)                                         \
    {                                                                     \
        const int i = base + J;                                           \
        u64 mixv = MIXEXPR;                                               \
        aa = mixv + mem.ld(i + m2);                                       \
        u64 y = g1 + aa + bb;                                             \
        mem.st(i + mr, y);                                                \
        u64 g2 = mem.ld((int)((y >> 11) & 255));                          \
        int nxt = (i + mr + 1) & 255; /* wraps to 0 after the last step (unused) */ \
        u64 xn = mem.ld(nxt);                                             \
        g1 = mem.ld((int)((xn >> 3) & 255));                              \
        bb = g2 + x;                                                      \
        if (i + mr >= 256 - ISAAC_TAIL) tail.put(i + mr, bb);             \
        x = xn;                                                           \
    }
            HR_ISAAC_STEP(0, ~(aa ^ (aa << 21)))
            HR_ISAAC_STEP(1, aa ^ (aa >> 5))
            HR_ISAAC_STEP(2, aa ^ (aa << 12))
            HR_ISAAC_STEP(3, aa ^ (aa >> 33))
#undef HR_ISAAC_STEP
        }
    }
}

// renderer.rs:34-36,53-54 + 165-167: per-path seed words s, t from the pixel / sub-sample (f64, exact)
HD void path_seed_words(uint32_t W, uint32_t H, uint32_t px, uint32_t py, uint32_t sub, u64 &s, u64 &t) {
    double fx = (double)px, fy = (double)(H - py);
    double ox = (double)(sub & 1) / 2.0 - 0.5, oy = (double)(sub >> 1) / 2.0 - 0.5;
    double rx = (double)W, ry = (double)H;
    double m = rx < ry ? rx : ry;
    double ncx = ((fx + ox) * 2.0 - rx) / m;
    double ncy = ((fy + oy) * 2.0 - ry) / m;
    s = (u64)((4.0 + ncx) * 100870.0);
    t = (u64)((4.0 + ncy) * 100304.0);
}

// Tail sink used by the production seed kernel: keeps fp32 draws + resolves the lens rejection loop
// (camera.rs:66-81) in f64 on the fly.  Draw k = step 255-k; lens attempt j uses draws (2j, 2j+1).
// Steps arrive in increasing order, i.e. v (odd k) before u (even k), attempts in decreasing j, so the
// last accepted attempt seen is the first one the reference's loop would accept.
template <class TailMem>  // void st(int k, float v); float ld(int k)
struct LensTail {
    TailMem &tm;
    int lens_shape;
    int accepted;      // attempt index, -1 = none yet
    float sqx, sqy;    // (2u-1, 2v-1) of the accepted attempt, rounded once from f64
    double pend_v;
    HD LensTail(TailMem &t, int shape) : tm(t), lens_shape(shape), accepted(-1), sqx(0.f), sqy(0.f), pend_v(0.0) {}
    HD void put(int step, u64 value) {
        int k = 255 - step;
        double dv = isaac_to_f64(value);
        tm.st(k, (float)dv);
        if (k & 1) {
            pend_v = dv;
        } else {
            double x = 2.0 * dv - 1.0, y = 2.0 * pend_v - 1.0;
            bool ok = (lens_shape == 0) || (x * x + y * y < 1.0);
            if (ok) { accepted = k >> 1; sqx = (float)x; sqy = (float)y; }
        }
    }
};

}  // namespace hr
