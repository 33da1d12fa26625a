// ISAAC-64 as arranged by crate `rand 0.4.3` (`StdRng` on 64-bit targets), host-side copy used by the
// scene-authoring code (main.rs:805-806 seeds a StdRng for the sphere generator).  The crate is a
// Cargo.lock dependency of the reference and is not vendored under /root/reference; this follows Bob
// Jenkins' public-domain ISAAC-64.  Pinned by the crate's known-answer vectors
// in tests/test_isaac64.py (through hh_debug_isaac64).
#pragma once
#include <cstdint>
#include <cstring>

namespace hh {

struct Isaac64 {
    uint64_t rsl[256], mem[256];
    uint64_t a, b, c;
    uint32_t cnt;

    void from_seed(const uint64_t *seed, int n) {
        for (int i = 0; i < 256; i++) rsl[i] = i < n ? seed[i] : 0;
        a = b = c = 0;
        init();
    }
    static inline void mix(uint64_t *r) {
        uint64_t &a = r[0], &b = r[1], &c = r[2], &d = r[3], &e = r[4], &f = r[5], &g = r[6], &h = r[7];
        a -= e; f ^= h >> 9;  h += a;
        b -= f; g ^= a << 9;  a += b;
        c -= g; h ^= b >> 23; b += c;
        d -= h; a ^= c << 15; c += d;
        e -= a; b ^= d >> 14; d += e;
        f -= b; c ^= e << 20; e += f;
        g -= c; d ^= f >> 17; f += g;
        h -= d; e ^= g << 14; g += h;
    }
    void init() {
        uint64_t r[8];
        for (auto &v : r) v = 0x9e3779b97f4a7c13ULL;
        for (int i = 0; i < 4; i++) mix(r);
        for (int i = 0; i < 256; i += 8) {
            for (int k = 0; k < 8; k++) r[k] += rsl[i + k];
            mix(r);
            for (int k = 0; k < 8; k++) mem[i + k] = r[k];
        }
        for (int i = 0; i < 256; i += 8) {
            for (int k = 0; k < 8; k++) r[k] += mem[i + k];
            mix(r);
            for (int k = 0; k < 8; k++) mem[i + k] = r[k];
        }
        round();
    }
    void round() {
        c += 1;
        uint64_t aa = a, bb = b + c;
        for (int half = 0; half < 2; half++) {
            int mr = half ? 128 : 0, m2 = half ? 0 : 128;
            for (int i = 0; i < 128; i++) {
                uint64_t m;
                switch (i & 3) {
                    case 0: m = ~(aa ^ (aa << 21)); break;
                    case 1: m = aa ^ (aa >> 5); break;
                    case 2: m = aa ^ (aa << 12); break;
                    default: m = aa ^ (aa >> 33); break;
                }
                uint64_t x = mem[i + mr];
                aa = m + mem[i + m2];
                uint64_t y = mem[(x >> 3) & 255] + aa + bb;
                mem[i + mr] = y;
                bb = mem[(y >> 11) & 255] + x;
                rsl[i + mr] = bb;
            }
        }
        a = aa; b = bb; cnt = 256;
    }
    uint64_t next_u64() {
        if (cnt == 0) round();
        cnt--;
        return rsl[cnt & 255];
    }
    // rand 0.4.3 Rng::next_f64 default: 52 mantissa bits in [1,2) minus 1 (unpinned)
    double next_f64() {
        uint64_t bits = 0x3FF0000000000000ULL | (next_u64() & 0x000FFFFFFFFFFFFFULL);
        double d;
        memcpy(&d, &bits, 8);
        return d - 1.0;
    }
    double gen_range(double lo, double hi) { return lo + (hi - lo) * next_f64(); }
};

}  // namespace hh
