// Per-path ISAAC-64 seeding for the device (rand 0.4.3 `StdRng::from_seed(&[8700304, sampling, s, t])`,
// renderer.rs:165-168; algorithm: Bob Jenkins' public-domain ISAAC-64 as arranged by `rand`,
// SURVEY.md Appendix B).  One lane = one generator.  The 2 KiB `mem` state of a lane is reached through
// a `Mem` accessor: on the GPU that is a bank-column of LDS (mem[i][lane], conflict-free gathers), in the
// host emulation a plain array.
//
// Only what a path can consume is produced: the generator hands out rsl[255], rsl[254], ... so the
// k-th draw is the output of round step 255-k.  The last TAILN steps (template parameter of isaac_round) go to a `Tail` sink.
#pragma once
#include "device_scene.h"

namespace hr {

typedef unsigned long long u64;

#if defined(__HIPCC__)
#define HR_NOUNROLL _Pragma("unroll 1")
#define HR_UNROLL _Pragma("unroll")
#else
#define HR_NOUNROLL
#define HR_UNROLL
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
    // (a software-pipelined form — pass-1 mix of block i + 1 next to the pass-2 mix of block i, two independent chains — was
    // measured and is no faster: a lone wave issues one instruction per ~4.5 cycles whether or not it depends on the last one)
    HR_NOUNROLL
    for (int i = 8 * SPLIT; i < 256; i += 8) {
        HR_ISAAC_MIX(a, b, c, d, e, f, g, h)
        A += a; B += b; C += c; D += d; E += e; F += f; G += g; H += h;
        HR_ISAAC_MIX(A, B, C, D, E, F, G, H)
        mem.st(i, A); mem.st(i + 1, B); mem.st(i + 2, C); mem.st(i + 3, D);
        mem.st(i + 4, E); mem.st(i + 5, F); mem.st(i + 6, G); mem.st(i + 7, H);
    }
}

// Mem: u64 ld(int i) / void st(int i, u64 v) / uint32_t off(int i) + u64 ldo(uint32_t) (offset of word i, load from it).  Tail: void put(int step, u64 value) for step >= 256 - TAILN (TAILN a multiple of 4, 8 .. 124).
template <int TAILN, class Mem, class Tail>
HD void isaac_round(Mem &mem, Tail &tail);

template <int TAILN, class Mem, class Tail>
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
    isaac_round<TAILN>(mem, tail);
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

// NG generators per lane, stepped in lockstep (mem[g], tail[g]): the wave's instruction stream then holds NG independent
// serial chains, and while one generator's gather is in flight the others' instructions issue — a single wave is otherwise
// idle for most of a step (one dependent LDS round trip per step, ~20 instructions to issue).  NG = 1 is the plain round.
template <int TAILN, int NG, class Mem, class Tail>
HD void isaac_round_n(Mem *mem, Tail *tail) {
    static_assert(TAILN % 4 == 0 && TAILN >= 8 && TAILN <= 124, "tail steps");
    // one isaac64() round: a = b = 0, c = 1  ->  aa = 0, bb = 1.  Reference step n:
    //     x = mem[n]; aa = mix(aa) + mem[(n+128)&255]; y = mem[(x>>3)&255] + aa + bb; mem[n] = y; bb = mem[(y>>11)&255] + x; out[n] = bb
    // With g1_n = mem[(x_n>>3)&255] (read after the store of y_{n-1}) and g2_n = mem[(y_n>>11)&255] (read after the store of
    // y_n):  y_{n+1} = (g1_{n+1} + aa_{n+1} + x_n) + g2_n.  The bracket T_{n+1} does not depend on g2_n, so the only serial
    // chain is  g2 arrives -> ONE 64-bit add -> store -> index of the next g2 -> LDS round trip.  g1_{n+1} is issued between
    // the store of y_n and the g2_n gather (its address is known early), so it is back first; x three steps ahead and the
    // +128 operand ride in the same LDS batch.  The two halves of the reference loop are kept as two loops so that every
    // static index is affine in the loop counter (no "& 255").  out[n] = g2_n + x_n is formed one step later, off the chain.
    u64 aa[NG], x[NG], xn[NG], xnn[NG], m2v[NG], g1[NG], T[NG], g2[NG], xprev[NG];
    uint32_t p1[NG];
    HR_UNROLL for (int g = 0; g < NG; g++) {
        x[g] = mem[g].ld(0); xn[g] = mem[g].ld(1); xnn[g] = mem[g].ld(2); m2v[g] = mem[g].ld(128);
        g1[g] = mem[g].ld((int)((x[g] >> 3) & 255));
        // step 0's T: y_0 = g1_0 + aa_0 + bb with bb = 1 and aa_0 = mix(0) + mem[128]
        aa[g] = ~(u64)0 + m2v[g];
        m2v[g] = mem[g].ld(129);
        T[g] = g1[g] + aa[g] + 1; g2[g] = 0; xprev[g] = 0;
        p1[g] = mem[g].off((int)((xn[g] >> 3) & 255));   // where g1 of the next step lives (known early)
    }
#define HR_ISAAC_STEP(N, MIXEXPR_NEXT, XN3_IDX, M2N_IDX, TAIL, TAILPREV)                                               \
    {                                                                                                                  \
        HR_UNROLL for (int g = 0; g < NG; g++) HR_OPAQUE32(p1[g]);                                                            \
        HR_SCHED_FENCE();                                                                                              \
        u64 g2n[NG];                                                                                                   \
        HR_UNROLL for (int g = 0; g < NG; g++) {                                                                       \
            u64 y = T[g] + g2[g];                      /* the chain: one add after g2_{N-1} is back */                  \
            mem[g].st(N, y);                                                                                           \
            g1[g] = mem[g].ldo(p1[g]);                 /* g1_{N+1}: after the store of y_N, before the g2_N gather */    \
            g2n[g] = mem[g].ld((int)((y >> 11) & 255));                                                                \
        }                                                                                                              \
        HR_SCHED_FENCE();                                                                                              \
        HR_UNROLL for (int g = 0; g < NG; g++) {                                                                       \
            if (TAILPREV) tail[g].put((N) - 1, g2[g] + xprev[g]);                                                      \
            g2[g] = g2n[g];                                                                                            \
            p1[g] = mem[g].off((int)((xnn[g] >> 3) & 255));                                                            \
            u64 xn3 = mem[g].ld(XN3_IDX);                                                                              \
            u64 m2n = mem[g].ld(M2N_IDX);                                                                              \
            const u64 A_ = aa[g];                                                                                      \
            aa[g] = (MIXEXPR_NEXT) + m2v[g];           /* aa_{N+1} */                                                   \
            T[g] = g1[g] + (aa[g] + x[g]);             /* T_{N+1} = g1_{N+1} + aa_{N+1} + x_N */                         \
            HR_OPAQUE64(T[g]);                                                                                         \
            xprev[g] = x[g]; x[g] = xn[g]; xn[g] = xnn[g]; xnn[g] = xn3; m2v[g] = m2n;                                 \
        }                                                                                                              \
    }
    // mix schedule: aa_{n+1} uses the mix of step n+1: n+1 = 0 mod 4: ~(a ^ a<<21), 1: a ^ a>>5, 2: a ^ a<<12, 3: a ^ a>>33
    // first half: n in [0,128): x from mem[n..], +128 operand from mem[n+128..]; the last groups are peeled because their
    // look-ahead operands wrap
    HR_ISAAC_STEP(0, A_ ^ (A_ >> 5), 3, 130, false, false)
    HR_ISAAC_STEP(1, A_ ^ (A_ << 12), 4, 131, false, false)
    HR_ISAAC_STEP(2, A_ ^ (A_ >> 33), 5, 132, false, false)
    HR_ISAAC_STEP(3, ~(A_ ^ (A_ << 21)), 6, 133, false, false)
    HR_NOUNROLL
    for (int n = 4; n < 124; n += 4) {
        HR_ISAAC_STEP(n, A_ ^ (A_ >> 5), n + 3, n + 130, false, false)
        HR_ISAAC_STEP(n + 1, A_ ^ (A_ << 12), n + 4, n + 131, false, false)
        HR_ISAAC_STEP(n + 2, A_ ^ (A_ >> 33), n + 5, n + 132, false, false)
        HR_ISAAC_STEP(n + 3, ~(A_ ^ (A_ << 21)), n + 6, n + 133, false, false)
    }
    HR_ISAAC_STEP(124, A_ ^ (A_ >> 5), 127, 254, false, false)
    HR_ISAAC_STEP(125, A_ ^ (A_ << 12), 128, 255, false, false)
    HR_ISAAC_STEP(126, A_ ^ (A_ >> 33), 129, 0, false, false)       // m2v for step 128 is mem[0]
    HR_ISAAC_STEP(127, ~(A_ ^ (A_ << 21)), 130, 1, false, false)
    // second half: n in [128,256): +128 operand from mem[n-128..]
    HR_NOUNROLL
    for (int n = 128; n < 256 - TAILN; n += 4) {
        HR_ISAAC_STEP(n, A_ ^ (A_ >> 5), n + 3, n - 126, false, false)
        HR_ISAAC_STEP(n + 1, A_ ^ (A_ << 12), n + 4, n - 125, false, false)
        HR_ISAAC_STEP(n + 2, A_ ^ (A_ >> 33), n + 5, n - 124, false, false)
        HR_ISAAC_STEP(n + 3, ~(A_ ^ (A_ << 21)), n + 6, n - 123, false, false)
    }
    HR_ISAAC_STEP(256 - TAILN, A_ ^ (A_ >> 5), 256 - TAILN + 3, 256 - TAILN - 126, true, false)
    HR_ISAAC_STEP(256 - TAILN + 1, A_ ^ (A_ << 12), 256 - TAILN + 4, 256 - TAILN - 125, true, true)
    HR_ISAAC_STEP(256 - TAILN + 2, A_ ^ (A_ >> 33), 256 - TAILN + 5, 256 - TAILN - 124, true, true)
    HR_ISAAC_STEP(256 - TAILN + 3, ~(A_ ^ (A_ << 21)), 256 - TAILN + 6, 256 - TAILN - 123, true, true)
    HR_NOUNROLL
    for (int n = 256 - TAILN + 4; n < 252; n += 4) {
        HR_ISAAC_STEP(n, A_ ^ (A_ >> 5), n + 3, n - 126, true, true)
        HR_ISAAC_STEP(n + 1, A_ ^ (A_ << 12), n + 4, n - 125, true, true)
        HR_ISAAC_STEP(n + 2, A_ ^ (A_ >> 33), n + 5, n - 124, true, true)
        HR_ISAAC_STEP(n + 3, ~(A_ ^ (A_ << 21)), n + 6, n - 123, true, true)
    }
    // last group: the look-ahead loads past the end are never used; point them at valid slots
    HR_ISAAC_STEP(252, A_ ^ (A_ >> 5), 255, 126, true, true)
    HR_ISAAC_STEP(253, A_ ^ (A_ << 12), 255, 127, true, true)
    HR_ISAAC_STEP(254, A_ ^ (A_ >> 33), 255, 127, true, true)
    HR_ISAAC_STEP(255, ~(A_ ^ (A_ << 21)), 255, 127, true, true)
    HR_UNROLL for (int g = 0; g < NG; g++) tail[g].put(255, g2[g] + xprev[g]);
#undef HR_ISAAC_STEP
}
template <int TAILN, class Mem, class Tail>
HD void isaac_round(Mem &mem, Tail &tail) { isaac_round_n<TAILN, 1>(&mem, &tail); }

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

// raw draw -> the fp32 value the trace kernel computes with (one rounding from the reference's f64)
HD float draw_f32(u64 raw) { return (float)isaac_to_f64(raw); }
HD float draw_lens_f32(u64 raw) { return (float)(2.0 * isaac_to_f64(raw) - 1.0); }
HD float uint_as_float(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
HD uint32_t float_as_uint(float f) { union { uint32_t u; float f; } c; c.f = f; return c.u; }

// Tail sink of the production seed kernels: fills the path's 128-byte hand-off record (device_scene.h).  Draw k (= the
// k-th next_u64 of the path) is the output of step 255-k; draws arrive in DECREASING k and are stored four at a time as
// the fp32 values the trace kernel computes with.  Lens attempt j uses draws (2j, 2j+1) (camera.rs:66-81); the first
// LENS_FAST attempts are judged on the fly, in f64 exactly as the reference (steps arrive v before u, attempts in
// decreasing j, so the last accepted one seen is the first the reference's loop would accept).  A path that rejects all of
// them reports overflow(): the caller queues it for the fix-up kernel.
template <class Store>  // void st4(int slot, float, float, float, float): slots slot .. slot + 3 of the record (slot % 4 == 0)
struct RecordTail {
    Store &store;
    int lens_shape;
    int accepted;  // attempt index, -1 = none among the first LENS_FAST
    u64 pend_v;
    float lx, ly, q1, q2, q3;
    HD RecordTail(Store &s, int shape) : store(s), lens_shape(shape), accepted(-1), pend_v(0), lx(0), ly(0), q1(0), q2(0), q3(0) {}
    HD void put(int step, u64 value) {
        const int k = 255 - step;
        if (k >= REC_DRAWS) return;
        const float f = draw_f32(value);
        switch (k & 3) {
            case 3: q3 = f; break;
            case 2: q2 = f; break;
            case 1: q1 = f; break;
            default: store.st4(k, f, q1, q2, q3); break;
        }
        if (k < 2 * LENS_FAST) {
            if (k & 1) pend_v = value;
            else if (lens_accept(value, pend_v, lens_shape)) { accepted = k >> 1; lx = draw_lens_f32(value); ly = draw_lens_f32(pend_v); }
        }
    }
    HD bool overflow() const { return accepted < 0; }
    HD void finish() { store.st4(REC_HEAD, uint_as_float(accepted < 0 ? 0u : (uint32_t)accepted), lx, ly, 0.0f); }
};

// The fix-up path (and the host emulation): given a window of raw outputs w[0 .. n), find the accepted lens attempt as the
// reference's loop does and write the record REBASED to a = 0 (slot i holds draw 2a + i).  Returns false when the path
// would need outputs beyond the window.
template <class Win, class Store>  // u64 Win::ld(int k)
HD bool record_from_window(const Win &w, int n, int lens_shape, Store &store) {
    int a = -1;
    for (int j = 0; 2 * j + 1 < n; j++)
        if (lens_accept(w.ld(2 * j), w.ld(2 * j + 1), lens_shape)) { a = j; break; }
    const bool ok = a >= 0 && 2 * (a + 9) + 1 < n;
    if (!ok) a = 0;
    for (int i = 0; i < REC_DRAWS; i += 4) {
        float f[4];
        for (int q = 0; q < 4; q++) f[q] = (i + q < DRAWS_PER_PATH) ? draw_f32(w.ld(2 * a + i + q)) : 0.0f;
        store.st4(i, f[0], f[1], f[2], f[3]);
    }
    store.st4(REC_HEAD, uint_as_float(0u), draw_lens_f32(w.ld(2 * a)), draw_lens_f32(w.ld(2 * a + 1)), 0.0f);
    return ok;
}

}  // namespace hr
