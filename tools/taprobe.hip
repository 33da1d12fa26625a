// What a node visit costs the L1's tag lookup (TA / TCP) under three fetch shapes, on an L2-resident table the size of the renderer's tree
// ("a traversal that shares one 64-byte node among the four lanes of a quad").  Every lane (or quad) chases pointers through
// a table of random successors, as a stackless walk does, with the box test's VALU work between two loads:
//   lane16 : one ray per lane, one 16-byte record per visit (the production box phase: one box per visit)
//   lane64 : one ray per lane, one 64-byte record per visit fetched by four 16-byte loads (a 4-wide node walked by ONE lane)
//   quad64 : one ray per QUAD, one 64-byte record per visit, each lane of the quad loads its own 16 bytes (one box per lane),
//            successor = a quad-level reduction (two DPP steps) of the four lanes' results
// `active` masks lanes off the way a box pass does (39 of 64 in production); quads are masked as a whole.  Second part: the cost table of ONE load
// instruction by width, distinct lines, slot spread, neighbouring-lane sharing and active lanes (profiles/r05_taprobe.txt, profiles/NOTES.md L).
// Output: ns per wave-iteration, visits per second per CU, and box tests per second (lane16: 1 per visit, lane64 / quad64: 4 per visit).
// hipcc --offload-arch=gfx950 -O3 tools/taprobe.hip -o tools/bin/taprobe
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32;
struct alignas(16) Rec { u32 next; float a, b, c; };

__device__ __forceinline__ float quad_min(float x) {
    // min over the four lanes of a quad: two DPP quad_perm exchanges
    float y = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x = fminf(x, y);
    y = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));         // quad_perm [2,3,0,1]
    return fminf(x, y);
}
__device__ __forceinline__ u32 quad_bcast0(u32 x) { return (u32)__builtin_amdgcn_mov_dpp((int)x, 0x00, 0xF, 0xF, true); }   // lane 0 of the quad

// the box test's arithmetic, roughly: 3 packed-ish FMAs per axis, max3 / min3, a compare (~19 VALU in production)
template <int WORK>
__device__ __forceinline__ float box_work(const Rec &r, float ox, float oy, float oz, float t) {
    float lo = fmaf(r.a, ox, -oy), hi = fmaf(r.b, oy, -oz), mid = fmaf(r.c, oz, -ox);
#pragma unroll
    for (int k = 0; k < WORK; k++) { lo = fmaf(lo, 0.999f, hi); hi = fmaf(hi, 1.001f, mid); mid = fmaf(mid, 0.998f, lo); }
    return fminf(fmaxf(fmaxf(lo, hi), mid), t);
}

template <int MODE, int WORK>
__global__ __launch_bounds__(256) void chase(const Rec *__restrict__ tab, u32 mask, int iters, unsigned long long active, float *out) {
    const u32 lane = threadIdx.x & 63u;
    const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = (active >> lane) & 1ull;
    u32 cur = ((MODE == 2 ? gid >> 2 : gid) * 2654435761u) & mask;
    float ox = 0.1f * (float)(lane & 7u), oy = 0.3f, oz = 0.7f, t = 1e30f, acc = 0.0f;
    if (act) {
        for (int i = 0; i < iters; i++) {
            if (MODE == 0) {
                const Rec r = tab[cur];
                const float e = box_work<WORK>(r, ox, oy, oz, t);
                acc += e;
                cur = (r.next + (e > 1e37f ? 1u : 0u)) & mask;
            } else if (MODE == 1) {
                const Rec *p = tab + (size_t)(cur & ~3u);
                const Rec r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
                const float e0 = box_work<WORK>(r0, ox, oy, oz, t), e1 = box_work<WORK>(r1, ox, oy, oz, t);
                const float e2 = box_work<WORK>(r2, ox, oy, oz, t), e3 = box_work<WORK>(r3, ox, oy, oz, t);
                const float e = fminf(fminf(e0, e1), fminf(e2, e3));
                acc += e;
                cur = (r0.next + (e > 1e37f ? 1u : 0u)) & mask;
            } else {
                const Rec r = tab[(size_t)(cur & ~3u) + (lane & 3u)];
                const float e = quad_min(box_work<WORK>(r, ox, oy, oz, t));
                acc += e;
                cur = (quad_bcast0(r.next) + (e > 1e37f ? 1u : 0u)) & mask;
            }
        }
    }
    if (acc == 12345.678f) out[gid] = acc + (float)cur;
    if (cur == 0xFFFFFFFFu) out[gid] = 1.0f;
}


// The cost table of ONE vector load: WIDTH bytes per lane (4 / 8 / 16), L distinct 128-byte lines per wave instruction (lanes l with the same
// l % L share a line and read different 16-byte slots of it), `active` lanes.  All records of a line carry the same successor, so a group stays
// together while it chases through the table.
template <int WIDTH>
__global__ __launch_bounds__(256) void width_chase(const Rec *__restrict__ tab, u32 line_mask, int iters, u32 L, unsigned long long active, float *out, u32 adjacent = 0, u32 spread = 1) {
    const u32 lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const bool act = (active >> lane) & 1ull;
    // adjacent = 0: lanes l, l + L, l + 2L ... share a line (neighbouring lanes never do); adjacent = G: G NEIGHBOURING lanes share a line
    u32 line = ((wave * 64u + (adjacent ? lane / adjacent : lane % L)) * 2654435761u >> 7) & line_mask;
    // spread = 1: the lanes' 16-byte slots are spread over the line (as a tree's records are); 0: every line is read at the same offset(s)
    const u32 slot = (adjacent ? lane % adjacent : lane / L + (spread ? lane % L : 0u)) & 7u;
    float acc = 0.0f;
    if (act) {
        for (int i = 0; i < iters; i++) {
            const Rec *p = tab + (size_t)line * 8u + slot;
            u32 nx;
            if (WIDTH == 4) { nx = p->next; acc += __builtin_bit_cast(float, nx); }
            else if (WIDTH == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(p); nx = v.x; acc += __builtin_bit_cast(float, v.y); }
            else { const Rec r = *p; nx = r.next; acc += r.a + r.b + r.c; }
            line = (nx >> 3) & line_mask;
        }
    }
    if (acc == 12345.678f || line == 0xFFFFFFFFu) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int WIDTH>
static void run_width(const Rec *tab, u32 line_mask, u32 L, int nactive, float *out, u32 adjacent = 0, u32 spread = 1) {
    unsigned long long active = 0;
    for (int l = 0; l < 64; l++) if ((l * nactive) / 64 != ((l + 1) * nactive) / 64) active |= 1ull << l;
    const int iters = 4000, wps = 5, blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    width_chase<WIDTH><<<blocks, 256>>>(tab, line_mask, 200, L, active, out, adjacent, spread);
    hipEventRecord(e0);
    width_chase<WIDTH><<<blocks, 256>>>(tab, line_mask, iters, L, active, out, adjacent, spread);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters;
    printf("width %2d B  lines/instr %2u (%s)  active lanes %2d : %7.3f ms  %5.1f CU-cycles per wave load instruction  %6.2f G wave loads/s\n",
           WIDTH, adjacent ? 64u / adjacent : L, adjacent ? "neighbouring lanes share" : spread ? "strided lanes share    " : "strided, ONE offset     ", __builtin_popcountll(active), ms, ms * 1e-3 * 2.4e9 * 256.0 / instr, instr / ms * 1e-6);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int MODE, int WORK>
static void run(const char *name, const Rec *tab, u32 mask, int waves_per_simd, int nactive, float *out) {
    // lanes (MODE 0, 1) or whole quads (MODE 2) switched off, spread over the wave
    unsigned long long active = 0;
    if (MODE == 2) {
        const int nq = (nactive + 3) / 4;
        for (int q = 0; q < 16; q++) if ((q * nq) / 16 != ((q + 1) * nq) / 16) active |= 0xFull << (4 * q);
    } else {
        for (int l = 0; l < 64; l++) if ((l * nactive) / 64 != ((l + 1) * nactive) / 64) active |= 1ull << l;
    }
    const int nact = __builtin_popcountll(active);
    const int iters = 4000, blocks = 256 * waves_per_simd;   // 4 waves per workgroup = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    chase<MODE, WORK><<<blocks, 256>>>(tab, mask, 200, active, out);
    hipEventRecord(e0);
    chase<MODE, WORK><<<blocks, 256>>>(tab, mask, iters, active, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double rays = (double)blocks * 4 * (MODE == 2 ? nact / 4 : nact);
    const double visits = rays * iters, boxes = visits * (MODE == 0 ? 1 : 4);
    const double wave_iters = (double)blocks * 4 * iters;
    printf("%-8s work %2d  waves/SIMD %d  active lanes %2d : %7.3f ms  %6.1f cycles per wave-iteration per SIMD-slot  %7.2f G visits/s  %7.2f G box tests/s\n",
           name, WORK, waves_per_simd, nact, ms, ms * 1e-3 * 2.4e9 / (wave_iters / (256.0 * 4)) , visits / ms * 1e-6, boxes / ms * 1e-6);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char **argv) {
    const u32 n = 1u << 17;   // 131,072 records x 16 B = 2 MiB: L2-resident, far beyond a CU's L1
    std::vector<Rec> h(n);
    u32 s = 12345u;
    for (u32 i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        h[i].next = (s >> 8) & (n - 1);
        if (i & 7u) h[i].next = h[i & ~7u].next;   // (width sweep: the records of a 128-byte line share their successor; the chase kernels above read it from slot 0 or from a random slot: still a random walk)
        h[i].a = 1.0f + (float)(s & 255u) * 1e-3f; h[i].b = 0.5f; h[i].c = 0.25f;
    }
    Rec *tab; float *out;
    hipMalloc(&tab, n * sizeof(Rec)); hipMalloc(&out, 1 << 24);
    hipMemcpy(tab, h.data(), n * sizeof(Rec), hipMemcpyHostToDevice);
    const u32 mask = n - 1;
    for (int wps : {4, 5, 8}) {
        for (int act : {64, 40}) {
            run<0, 4>("lane16", tab, mask, wps, act, out);
            run<1, 4>("lane64", tab, mask, wps, act, out);
            run<2, 4>("quad64", tab, mask, wps, act, out);
        }
    }
    // no arithmetic between the loads: the memory pipe alone
    run<0, 0>("lane16", tab, mask, 5, 64, out);
    run<1, 0>("lane64", tab, mask, 5, 64, out);
    run<2, 0>("quad64", tab, mask, 5, 64, out);
    run<0, 0>("lane16", tab, mask, 5, 40, out);
    run<2, 0>("quad64", tab, mask, 5, 40, out);
    // the cost table of one load instruction
    const u32 line_mask = n / 8 - 1;
    for (u32 L : {1u, 4u, 8u, 16u, 32u, 64u}) {
        run_width<4>(tab, line_mask, L, 64, out);
        run_width<8>(tab, line_mask, L, 64, out);
        run_width<16>(tab, line_mask, L, 64, out);
    }
    for (u32 L : {16u, 64u}) {             // every lane at the SAME offset of its line (one bank of the L1's data array)
        run_width<16>(tab, line_mask, L, 64, out, 0, 0);
        run_width<8>(tab, line_mask, L, 64, out, 0, 0);
    }
    for (u32 G : {2u, 4u, 8u, 16u}) {      // G neighbouring lanes read consecutive 16-byte (8-byte, 4-byte) slots of one line
        run_width<16>(tab, line_mask, 64 / G, 64, out, G);
        run_width<8>(tab, line_mask, 64 / G, 64, out, G);
        run_width<4>(tab, line_mask, 64 / G, 64, out, G);
    }
    for (int act : {16, 32, 48}) {
        run_width<16>(tab, line_mask, 64, act, out);
        run_width<16>(tab, line_mask, 8, act, out);
        run_width<4>(tab, line_mask, 64, act, out);
    }
    return 0;
}
