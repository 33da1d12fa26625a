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

// The two init passes of `Isaac64Rng::init(true)` WITHOUT a 2 KiB scratch array: pass 2 needs the end state of
// pass 1 before it can start, so pass 1 is run once "dry" (registers only) and then regenerated block by block
// next to pass 2.  96 mixes instead of 64, but no LDS: this is what lets the init run at full occupancy in its own
// kernel while the LDS-bound round kernel only has to load the result.  Out: void st(int i, u64 v), i = 0..255.
template <class Out>
HD void isaac_init_final(Out &out, const IsaacWarm &w, u64 s0, u64 s1, u64 s2, u64 s3) {
    u64 a = w.r[0] + s0, b = w.r[1] + s1, c = w.r[2] + s2, d = w.r[3] + s3, e = w.r[4], f = w.r[5], g = w.r[6], h = w.r[7];
    HR_NOUNROLL
    for (int i = 0; i < 32; i++) { HR_ISAAC_MIX(a, b, c, d, e, f, g, h) }
    u64 A = a, B = b, C = c, D = d, E = e, F = f, G = g, H = h;   // pass 2 continues from here
    a = w.r[0] + s0; b = w.r[1] + s1; c = w.r[2] + s2; d = w.r[3] + s3; e = w.r[4]; f = w.r[5]; g = w.r[6]; h = w.r[7];
    HR_NOUNROLL
    for (int i = 0; i < 256; i += 8) {
        HR_ISAAC_MIX(a, b, c, d, e, f, g, h)                       // pass-1 block: what rsl-pass stored in mem[i..i+8)
        A += a; B += b; C += c; D += d; E += e; F += f; G += g; H += h;
        HR_ISAAC_MIX(A, B, C, D, E, F, G, H)
        out.st(i, A); out.st(i + 1, B); out.st(i + 2, C); out.st(i + 3, D);
        out.st(i + 4, E); out.st(i + 5, F); out.st(i + 6, G); out.st(i + 7, H);
    }
}

// Mem: u64 ld(int i) / void st(int i, u64 v).  Tail: void put(int step, u64 value) for step >= 256 - ISAAC_TAIL.
template <class Mem, class Tail>
HD void isaac_round(Mem &mem, Tail &tail);

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
    isaac_round(mem, tail);
}

template <class Mem, class Tail>
HD void isaac_round(Mem &mem, Tail &tail) {
    // one isaac64() round: a = b = 0, c = 1  ->  aa = 0, bb = 1.
    // Step n (0..255) reads x = mem[n] and mem[(n+128)&255], gathers mem[(x>>3)&255], stores y to mem[n],
    // gathers mem[(y>>11)&255].  The only serial chain is bb -> y -> second gather -> bb; everything else
    // (x two steps ahead, the +128 operand and the first gather of the NEXT step) is issued in the same
    // batch as the second gather, so a step costs one LDS round trip.  The two halves of the reference
    // loop are kept as two loops so that every static index is affine in the loop counter (no "& 255").
    u64 aa = 0, bb = 1;
    u64 x = mem.ld(0), xn = mem.ld(1), m2v = mem.ld(128);
    u64 g1 = mem.ld((int)((x >> 3) & 255));
#define HR_ISAAC_STEP(N, MIXEXPR, XNN_IDX, M2N_IDX, TAIL)                 \
    {                                                                     \
        u64 mixv = MIXEXPR;                                               \
        aa = mixv + m2v;                                                  \
        u64 y = g1 + aa + bb;                                             \
        mem.st(N, y);                                                     \
        u64 g2 = mem.ld((int)((y >> 11) & 255));                          \
        g1 = mem.ld((int)((xn >> 3) & 255));                              \
        u64 xnn = mem.ld(XNN_IDX);                                        \
        m2v = mem.ld(M2N_IDX);                                            \
        bb = g2 + x;                                                      \
        if (TAIL) tail.put(N, bb);                                        \
        x = xn; xn = xnn;                                                 \
    }
    // first half: n in [0,128): x from mem[n..], +128 operand from mem[n+128..]; the last group is peeled
    // because its "next" +128 operand is mem[0]
    HR_NOUNROLL
    for (int n = 0; n < 124; n += 4) {
        HR_ISAAC_STEP(n, ~(aa ^ (aa << 21)), n + 2, n + 129, false)
        HR_ISAAC_STEP(n + 1, aa ^ (aa >> 5), n + 3, n + 130, false)
        HR_ISAAC_STEP(n + 2, aa ^ (aa << 12), n + 4, n + 131, false)
        HR_ISAAC_STEP(n + 3, aa ^ (aa >> 33), n + 5, n + 132, false)
    }
    HR_ISAAC_STEP(124, ~(aa ^ (aa << 21)), 126, 253, false)
    HR_ISAAC_STEP(125, aa ^ (aa >> 5), 127, 254, false)
    HR_ISAAC_STEP(126, aa ^ (aa << 12), 128, 255, false)
    HR_ISAAC_STEP(127, aa ^ (aa >> 33), 129, 0, false)
    // second half: n in [128,256): +128 operand from mem[n-128..]
    HR_NOUNROLL
    for (int n = 128; n < 256 - ISAAC_TAIL; n += 4) {
        HR_ISAAC_STEP(n, ~(aa ^ (aa << 21)), n + 2, n - 127, false)
        HR_ISAAC_STEP(n + 1, aa ^ (aa >> 5), n + 3, n - 126, false)
        HR_ISAAC_STEP(n + 2, aa ^ (aa << 12), n + 4, n - 125, false)
        HR_ISAAC_STEP(n + 3, aa ^ (aa >> 33), n + 5, n - 124, false)
    }
    HR_NOUNROLL
    for (int n = 256 - ISAAC_TAIL; n < 252; n += 4) {
        HR_ISAAC_STEP(n, ~(aa ^ (aa << 21)), n + 2, n - 127, true)
        HR_ISAAC_STEP(n + 1, aa ^ (aa >> 5), n + 3, n - 126, true)
        HR_ISAAC_STEP(n + 2, aa ^ (aa << 12), n + 4, n - 125, true)
        HR_ISAAC_STEP(n + 3, aa ^ (aa >> 33), n + 5, n - 124, true)
    }
    // last group: the look-ahead loads past the end are never used; point them at valid slots
    HR_ISAAC_STEP(252, ~(aa ^ (aa << 21)), 254, 125, true)
    HR_ISAAC_STEP(253, aa ^ (aa >> 5), 255, 126, true)
    HR_ISAAC_STEP(254, aa ^ (aa << 12), 255, 127, true)
    HR_ISAAC_STEP(255, aa ^ (aa >> 33), 255, 127, true)
#undef HR_ISAAC_STEP
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

// camera.rs:66-81: does lens attempt (u, v) pass the rejection test?  f64, exactly as the reference.
HD bool lens_accept(u64 raw_u, u64 raw_v, int lens_shape) {
    double x = 2.0 * isaac_to_f64(raw_u) - 1.0, y = 2.0 * isaac_to_f64(raw_v) - 1.0;
    return lens_shape == 0 || (x * x + y * y < 1.0);
}

// Tail sink of the production seed kernel.  Draw k (= the k-th next_u64 of the path) is the output of
// step 255-k; it is stored RAW (the trace kernel converts the few it consumes).  Lens attempt j uses
// draws (2j, 2j+1); the first LENS_FAST attempts are judged on the fly (steps arrive v before u, attempts
// in decreasing j, so the last accepted one seen is the first the reference's loop would accept); the
// rare path that rejects all of them is resolved afterwards by lens_slow() from the stored outputs.
static const int LENS_FAST = 8;
template <class Store>  // void st(int k, u64 v); u64 ld(int k)
struct RawLensTail {
    Store &store;
    int lens_shape;
    int accepted;  // attempt index, -1 = none among the first LENS_FAST
    u64 pend_v;
    HD RawLensTail(Store &s, int shape) : store(s), lens_shape(shape), accepted(-1), pend_v(0) {}
    HD void put(int step, u64 value) {
        int k = 255 - step;
        store.st(k, value);
        if (k < 2 * LENS_FAST) {
            if (k & 1) pend_v = value;
            else if (lens_accept(value, pend_v, lens_shape)) accepted = k >> 1;
        }
    }
    HD void lens_slow() {
        if (accepted >= 0) return;
        for (int j = LENS_FAST; j < ISAAC_TAIL / 2; j++)
            if (lens_accept(store.ld(2 * j), store.ld(2 * j + 1), lens_shape)) { accepted = j; return; }
    }
    // a path consumes draws up to index 2*(accepted + 9) + 1
    HD bool in_window() const { return accepted >= 0 && 2 * (accepted + 9) + 1 < ISAAC_TAIL; }
};

// raw draw -> the fp32 value the trace kernel computes with (one rounding from the reference's f64)
HD float draw_f32(u64 raw) { return (float)isaac_to_f64(raw); }
HD float draw_lens_f32(u64 raw) { return (float)(2.0 * isaac_to_f64(raw) - 1.0); }

}  // namespace hr
