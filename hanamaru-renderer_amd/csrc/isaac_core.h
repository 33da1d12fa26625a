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
// kernel / wave while the LDS-bound round only has to load the result.  Out: void st2(int i, u64 v0, u64 v1) stores
// words i (even) and i + 1.
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
        out.st2(i, A, B); out.st2(i + 2, C, D); out.st2(i + 4, E, F); out.st2(i + 6, G, H);
    }
}

// isaac_init_final cut in two for the producer / consumer seed kernel: the PRODUCER (isaac_init_front) runs the dry pass 1
// and blocks < SPLIT of the pass-2 sweep, stores those blocks (out.st2) and the 16 registers the sweep continues from
// (out.end2: a..h of the regenerated pass 1, then A..H of pass 2); the CONSUMER (isaac_init_back) continues with blocks
// >= SPLIT straight into its generator memory while the stored part is still in flight.  Together they produce exactly
// the state of isaac_init_final in 32 + 64 mixes; what travels is SPLIT / 32 of the state plus 128 bytes.
template <int SPLIT, class Out>
HD void isaac_init_front(Out &out, const IsaacWarm &w, u64 s0, u64 s1, u64 s2, u64 s3) {
    u64 a = w.r[0] + s0, b = w.r[1] + s1, c = w.r[2] + s2, d = w.r[3] + s3, e = w.r[4], f = w.r[5], g = w.r[6], h = w.r[7];
    HR_NOUNROLL
    for (int i = 0; i < 32; i++) { HR_ISAAC_MIX(a, b, c, d, e, f, g, h) }
    u64 A = a, B = b, C = c, D = d, E = e, F = f, G = g, H = h;
    a = w.r[0] + s0; b = w.r[1] + s1; c = w.r[2] + s2; d = w.r[3] + s3; e = w.r[4]; f = w.r[5]; g = w.r[6]; h = w.r[7];
    HR_NOUNROLL
    for (int i = 0; i < 8 * SPLIT; i += 8) {
        HR_ISAAC_MIX(a, b, c, d, e, f, g, h)
        A += a; B += b; C += c; D += d; E += e; F += f; G += g; H += h;
        HR_ISAAC_MIX(A, B, C, D, E, F, G, H)
        out.st2(i, A, B); out.st2(i + 2, C, D); out.st2(i + 4, E, F); out.st2(i + 6, G, H);
    }
    out.end2(0, a, b); out.end2(2, c, d); out.end2(4, e, f); out.end2(6, g, h);
    out.end2(8, A, B); out.end2(10, C, D); out.end2(12, E, F); out.end2(14, G, H);
}
template <int SPLIT, class Mem>
HD void isaac_init_back(Mem &mem, const u64 *st16) {
    u64 a = st16[0], b = st16[1], c = st16[2], d = st16[3], e = st16[4], f = st16[5], g = st16[6], h = st16[7];
    u64 A = st16[8], B = st16[9], C = st16[10], D = st16[11], E = st16[12], F = st16[13], G = st16[14], H = st16[15];
    HR_NOUNROLL
    for (int i = 8 * SPLIT; i < 256; i += 8) {
        HR_ISAAC_MIX(a, b, c, d, e, f, g, h)
        A += a; B += b; C += c; D += d; E += e; F += f; G += g; H += h;
        HR_ISAAC_MIX(A, B, C, D, E, F, G, H)
        mem.st(i, A); mem.st(i + 1, B); mem.st(i + 2, C); mem.st(i + 3, D);
        mem.st(i + 4, E); mem.st(i + 5, F); mem.st(i + 6, G); mem.st(i + 7, H);
    }
}

// Mem: u64 ld(int i) / void st(int i, u64 v) / uint32_t off(int i) + u64 ldo(uint32_t) (offset of word i, load from it).  Tail: void put(int step, u64 value) for step >= 256 - ISAAC_TAIL.
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

#if defined(__HIP_DEVICE_COMPILE__)
#define HR_OPAQUE64(v) asm volatile("" : "+v"(v))   // keeps an off-chain partial sum from being re-associated onto the serial chain
#define HR_OPAQUE32(p) asm volatile("" : "+v"(p))
#define HR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)   // nothing is scheduled across: keeps off-chain work out of the chain
#else
#define HR_OPAQUE64(v)
#define HR_OPAQUE32(p)
#define HR_SCHED_FENCE()
#endif

template <class Mem, class Tail>
HD void isaac_round(Mem &mem, Tail &tail) {
    // one isaac64() round: a = b = 0, c = 1  ->  aa = 0, bb = 1.  Reference step n:
    //     x = mem[n]; aa = mix(aa) + mem[(n+128)&255]; y = mem[(x>>3)&255] + aa + bb; mem[n] = y; bb = mem[(y>>11)&255] + x; out[n] = bb
    // With g1_n = mem[(x_n>>3)&255] (read after the store of y_{n-1}) and g2_n = mem[(y_n>>11)&255] (read after the store of
    // y_n):  y_{n+1} = (g1_{n+1} + aa_{n+1} + x_n) + g2_n.  The bracket T_{n+1} does not depend on g2_n, so the only serial
    // chain is  g2 arrives -> ONE 64-bit add -> store -> index of the next g2 -> LDS round trip.  g1_{n+1} is issued between
    // the store of y_n and the g2_n gather (its address is known early), so it is back first; x three steps ahead and the
    // +128 operand ride in the same LDS batch.  The two halves of the reference loop are kept as two loops so that every
    // static index is affine in the loop counter (no "& 255").  out[n] = g2_n + x_n is formed one step later, off the chain.
    u64 aa = 0;
    u64 x = mem.ld(0), xn = mem.ld(1), xnn = mem.ld(2), m2v = mem.ld(128);
    u64 g1 = mem.ld((int)((x >> 3) & 255));
    // step 0's T: y_0 = g1_0 + aa_0 + bb with bb = 1 and aa_0 = mix(0) + mem[128]
    aa = ~(aa ^ (aa << 21)) + m2v;
    m2v = mem.ld(129);
    u64 T = g1 + aa + 1, g2 = 0, xprev = 0;
    uint32_t p1 = mem.off((int)((xn >> 3) & 255));   // where g1 of the next step lives (known early)
#define HR_ISAAC_STEP(N, MIXEXPR_NEXT, XN3_IDX, M2N_IDX, TAIL, TAILPREV)                                               \
    {                                                                                                                  \
        HR_OPAQUE32(p1);                                                                                               \
        HR_SCHED_FENCE();                                                                                              \
        u64 y = T + g2;                            /* the chain: one add after g2_{N-1} is back */                      \
        mem.st(N, y);                                                                                                  \
        g1 = mem.ldo(p1);                          /* g1_{N+1}: after the store of y_N, before the g2_N gather */        \
        u64 g2n = mem.ld((int)((y >> 11) & 255));                                                                      \
        HR_SCHED_FENCE();                                                                                              \
        if (TAILPREV) tail.put((N) - 1, g2 + xprev);                                                                   \
        g2 = g2n;                                                                                                      \
        p1 = mem.off((int)((xnn >> 3) & 255));                                                                         \
        u64 xn3 = mem.ld(XN3_IDX);                                                                                     \
        u64 m2n = mem.ld(M2N_IDX);                                                                                     \
        aa = (MIXEXPR_NEXT) + m2v;                 /* aa_{N+1} */                                                       \
        T = g1 + (aa + x);                         /* T_{N+1} = g1_{N+1} + aa_{N+1} + x_N */                             \
        HR_OPAQUE64(T);                                                                                                \
        xprev = x; x = xn; xn = xnn; xnn = xn3; m2v = m2n;                                                             \
    }
    // mix schedule: aa_{n+1} uses the mix of step n+1: n+1 = 0 mod 4: ~(a ^ a<<21), 1: a ^ a>>5, 2: a ^ a<<12, 3: a ^ a>>33
    // first half: n in [0,128): x from mem[n..], +128 operand from mem[n+128..]; the last groups are peeled because their
    // look-ahead operands wrap
    HR_ISAAC_STEP(0, aa ^ (aa >> 5), 3, 130, false, false)
    HR_ISAAC_STEP(1, aa ^ (aa << 12), 4, 131, false, false)
    HR_ISAAC_STEP(2, aa ^ (aa >> 33), 5, 132, false, false)
    HR_ISAAC_STEP(3, ~(aa ^ (aa << 21)), 6, 133, false, false)
    HR_NOUNROLL
    for (int n = 4; n < 124; n += 4) {
        HR_ISAAC_STEP(n, aa ^ (aa >> 5), n + 3, n + 130, false, false)
        HR_ISAAC_STEP(n + 1, aa ^ (aa << 12), n + 4, n + 131, false, false)
        HR_ISAAC_STEP(n + 2, aa ^ (aa >> 33), n + 5, n + 132, false, false)
        HR_ISAAC_STEP(n + 3, ~(aa ^ (aa << 21)), n + 6, n + 133, false, false)
    }
    HR_ISAAC_STEP(124, aa ^ (aa >> 5), 127, 254, false, false)
    HR_ISAAC_STEP(125, aa ^ (aa << 12), 128, 255, false, false)
    HR_ISAAC_STEP(126, aa ^ (aa >> 33), 129, 0, false, false)       // m2v for step 128 is mem[0]
    HR_ISAAC_STEP(127, ~(aa ^ (aa << 21)), 130, 1, false, false)
    // second half: n in [128,256): +128 operand from mem[n-128..]
    HR_NOUNROLL
    for (int n = 128; n < 256 - ISAAC_TAIL; n += 4) {
        HR_ISAAC_STEP(n, aa ^ (aa >> 5), n + 3, n - 126, false, false)
        HR_ISAAC_STEP(n + 1, aa ^ (aa << 12), n + 4, n - 125, false, false)
        HR_ISAAC_STEP(n + 2, aa ^ (aa >> 33), n + 5, n - 124, false, false)
        HR_ISAAC_STEP(n + 3, ~(aa ^ (aa << 21)), n + 6, n - 123, false, false)
    }
    HR_ISAAC_STEP(256 - ISAAC_TAIL, aa ^ (aa >> 5), 256 - ISAAC_TAIL + 3, 256 - ISAAC_TAIL - 126, true, false)
    HR_ISAAC_STEP(256 - ISAAC_TAIL + 1, aa ^ (aa << 12), 256 - ISAAC_TAIL + 4, 256 - ISAAC_TAIL - 125, true, true)
    HR_ISAAC_STEP(256 - ISAAC_TAIL + 2, aa ^ (aa >> 33), 256 - ISAAC_TAIL + 5, 256 - ISAAC_TAIL - 124, true, true)
    HR_ISAAC_STEP(256 - ISAAC_TAIL + 3, ~(aa ^ (aa << 21)), 256 - ISAAC_TAIL + 6, 256 - ISAAC_TAIL - 123, true, true)
    HR_NOUNROLL
    for (int n = 256 - ISAAC_TAIL + 4; n < 252; n += 4) {
        HR_ISAAC_STEP(n, aa ^ (aa >> 5), n + 3, n - 126, true, true)
        HR_ISAAC_STEP(n + 1, aa ^ (aa << 12), n + 4, n - 125, true, true)
        HR_ISAAC_STEP(n + 2, aa ^ (aa >> 33), n + 5, n - 124, true, true)
        HR_ISAAC_STEP(n + 3, ~(aa ^ (aa << 21)), n + 6, n - 123, true, true)
    }
    // last group: the look-ahead loads past the end are never used; point them at valid slots
    HR_ISAAC_STEP(252, aa ^ (aa >> 5), 255, 126, true, true)
    HR_ISAAC_STEP(253, aa ^ (aa << 12), 255, 127, true, true)
    HR_ISAAC_STEP(254, aa ^ (aa >> 33), 255, 127, true, true)
    HR_ISAAC_STEP(255, ~(aa ^ (aa << 21)), 255, 127, true, true)
    tail.put(255, g2 + xprev);
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
