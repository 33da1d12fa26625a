// Image decode / encode for the host layer (the reference uses the `image` crate: texture.rs:18,
// renderer.rs:97, main.rs:1215-1217).  No libpng / libjpeg headers exist in this image, so both codecs
// are written here over zlib.
//
//  * PNG reader: 8-bit, colour types 0/2/3/4/6, non-interlaced (magic-circle3.png is 8-bit RGBA).
//  * PNG writer: RGB8, filter 0.
//  * JPEG reader: baseline / extended-sequential Huffman, 8-bit, up to 4:2:0.  The crate the reference
//    pins (jpeg-decoder 0.1.15) is not vendored under /root/reference; its decode pipeline is, as far as
//    recalled, the public-domain stb_image one: integer "islow" IDCT with 12-bit constants, triangle-filter
//    (3:1) chroma upsampling with +8 rounding, float YCbCr->RGB with +0.5 rounding.  That published
//    algorithm is what is restated below.  No reference test covers the crate directly; it is pinned through
//    the renders: the oracle's 1920x1080 x 1,000-sampling image of the default scene — its sky is six of
//    these JPEGs — is byte-identical to the reference binary's committed PNG over the whole frame
//    (tests/test_oracle.py test_oracle_whole_frame_pin); unit tests compare with Pillow/libjpeg besides.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>

#include "hanamaru_host.h"
#include "hh_math.h"

namespace hh {

static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

static bool read_file(const char *path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return false; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if (got != out.size()) { set_error("short read on %s", path); return false; }
    return true;
}

// ------------------------------------------------------------------------------------------ PNG

static uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

static int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

static bool decode_png(const std::vector<uint8_t> &d, std::vector<uint8_t> &rgba, uint32_t &w, uint32_t &h) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 8 || memcmp(d.data(), sig, 8)) { set_error("not a PNG"); return false; }
    size_t pos = 8;
    std::vector<uint8_t> idat, plte;
    int depth = 0, ctype = 0, interlace = 0;
    w = h = 0;
    while (pos + 12 <= d.size()) {
        uint32_t len = be32(&d[pos]);
        const uint8_t *type = &d[pos + 4];
        const uint8_t *body = &d[pos + 8];
        if (pos + 12 + (size_t)len > d.size()) { set_error("truncated PNG chunk"); return false; }
        if (!memcmp(type, "IHDR", 4)) {
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!w || !h) { set_error("PNG without IHDR"); return false; }
    if (depth != 8 || interlace != 0) { set_error("PNG: only 8-bit non-interlaced supported"); return false; }
    int ch;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 3: ch = 1; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: set_error("PNG: bad colour type %d", ctype); return false;
    }
    size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawlen = raw.size();
    int zr = uncompress(raw.data(), &rawlen, idat.data(), idat.size());
    if (zr != Z_OK || rawlen != raw.size()) { set_error("PNG inflate failed (%d)", zr); return false; }
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *in = &raw[(stride + 1) * y];
        uint8_t ft = *in++;
        uint8_t *cur = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            int a = i >= (size_t)ch ? cur[i - ch] : 0;
            int b = up ? up[i] : 0;
            int c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int v = in[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: set_error("PNG: bad filter %d", ft); return false;
            }
            cur[i] = (uint8_t)v;
        }
    }
    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t *p = &img[i * ch];
        uint8_t r, g, b, a = 255;
        switch (ctype) {
            case 0: r = g = b = p[0]; break;
            case 2: r = p[0]; g = p[1]; b = p[2]; break;
            case 3: {
                size_t k = (size_t)p[0] * 3;
                if (k + 2 < plte.size()) { r = plte[k]; g = plte[k + 1]; b = plte[k + 2]; }
                else { r = g = b = 0; }
                break;
            }
            case 4: r = g = b = p[0]; a = p[1]; break;
            default: r = p[0]; g = p[1]; b = p[2]; a = p[3]; break;
        }
        rgba[i * 4] = r; rgba[i * 4 + 1] = g; rgba[i * 4 + 2] = b; rgba[i * 4 + 3] = a;
    }
    return true;
}

static void put_be32(std::vector<uint8_t> &o, uint32_t v) {
    o.push_back(v >> 24); o.push_back(v >> 16); o.push_back(v >> 8); o.push_back(v);
}
static void put_chunk(std::vector<uint8_t> &o, const char *type, const uint8_t *body, size_t len) {
    put_be32(o, (uint32_t)len);
    size_t start = o.size();
    o.insert(o.end(), type, type + 4);
    if (len) o.insert(o.end(), body, body + len);
    put_be32(o, (uint32_t)crc32(0, &o[start], (uInt)(len + 4)));
}

static bool write_png_rgb8(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)w * 3 + 1) * h);
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    uLongf clen = compressBound(raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), raw.size(), 6) != Z_OK) { set_error("deflate failed"); return false; }
    std::vector<uint8_t> o = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    uint8_t ihdr[13];
    ihdr[0] = w >> 24; ihdr[1] = w >> 16; ihdr[2] = w >> 8; ihdr[3] = w;
    ihdr[4] = h >> 24; ihdr[5] = h >> 16; ihdr[6] = h >> 8; ihdr[7] = h;
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    put_chunk(o, "IHDR", ihdr, 13);
    put_chunk(o, "IDAT", comp.data(), clen);
    put_chunk(o, "IEND", nullptr, 0);
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("cannot write %s", path); return false; }
    bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    fclose(f);
    if (!ok) set_error("short write on %s", path);
    return ok;
}

// ----------------------------------------------------------------------------------------- JPEG

namespace jpg {

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    void build(const uint8_t *counts, const uint8_t *symbols) {
        present = true;
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k;
            mincode[len] = code;
            for (int i = 0; i < counts[len - 1]; i++) vals[k++] = *symbols++;
            code += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;  // blocks per row / column (padded to MCU)
    int pred = 0;
    std::vector<uint8_t> plane;  // (bw*8) x (bh*8)
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
    void fill() {
        while (nbits <= 24) {
            int byte = 0;
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xff) {
                    if (p + 1 < end && p[1] == 0x00) { p += 2; }
                    else { hit_marker = true; byte = 0; }
                } else {
                    p++;
                }
            }
            acc |= (uint32_t)byte << (24 - nbits);
            nbits += 8;
        }
    }
    int get(int n) {
        if (!n) return 0;
        if (nbits < n) fill();
        int v = (int)(acc >> (32 - n));
        acc <<= n;
        nbits -= n;
        return v;
    }
    int decode(const Huff &h) {
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | get(1);
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len])
                return h.vals[h.valptr[len] + code - h.mincode[len]];
        }
        return -1;
    }
};

static inline int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }
static inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

#define F2F(x) ((int)(((x) * 4096 + 0.5)))
#define FSH(x) ((x) * 4096)
#define IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)       \
    int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3; \
    p2 = s2; p3 = s6;                                 \
    p1 = (p2 + p3) * F2F(0.5411961f);                 \
    t2 = p1 + p3 * F2F(-1.847759065f);                \
    t3 = p1 + p2 * F2F(0.765366865f);                 \
    p2 = s0; p3 = s4;                                 \
    t0 = FSH(p2 + p3); t1 = FSH(p2 - p3);             \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2; \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;               \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2; \
    p5 = (p3 + p4) * F2F(1.175875602f);               \
    t0 = t0 * F2F(0.298631336f);                      \
    t1 = t1 * F2F(2.053119869f);                      \
    t2 = t2 * F2F(3.072711026f);                      \
    t3 = t3 * F2F(1.501321110f);                      \
    p1 = p5 + p1 * F2F(-0.899976223f);                \
    p2 = p5 + p2 * F2F(-2.562915447f);                \
    p3 = p3 * F2F(-1.961570560f);                     \
    p4 = p4 * F2F(-0.390180644f);                     \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

static void idct_block(const int *d, uint8_t *out, int stride) {
    int val[64];
    for (int i = 0; i < 8; i++) {
        const int *c = d + i;
        int *v = val + i;
        if (!c[8] && !c[16] && !c[24] && !c[32] && !c[40] && !c[48] && !c[56]) {
            int dc = c[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
        } else {
            IDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; i++) {
        const int *v = val + i * 8;
        uint8_t *o = out + i * stride;
        IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17);
        x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
        o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
    }
}

// chroma upsampling to full resolution; triangle filter with +8 rounding for the 2x cases
static void upsample(const Comp &c, int hmax, int vmax, int W, int H, std::vector<uint8_t> &out) {
    int sw = c.bw * 8, sh = c.bh * 8;
    int iw = (W * c.h + hmax - 1) / hmax, ih = (H * c.v + vmax - 1) / vmax;  // valid input extent
    out.assign((size_t)W * H, 0);
    int hs = hmax / c.h, vs = vmax / c.v;
    (void)sh;
    if (hs == 1 && vs == 1) {
        for (int y = 0; y < H; y++) memcpy(&out[(size_t)y * W], &c.plane[(size_t)y * sw], W);
        return;
    }
    std::vector<int> t(iw + 2);
    std::vector<uint8_t> row(2 * (size_t)iw + 2);
    for (int y = 0; y < H; y++) {
        const uint8_t *near_, *far_;
        if (vs == 2) {
            int rn = y >> 1;
            int rf = (y & 1) ? rn + 1 : rn - 1;
            if (rf < 0) rf = 0;
            if (rf > ih - 1) rf = ih - 1;
            near_ = &c.plane[(size_t)rn * sw]; far_ = &c.plane[(size_t)rf * sw];
        } else if (vs == 1) {
            near_ = far_ = &c.plane[(size_t)y * sw];
        } else {
            near_ = far_ = &c.plane[(size_t)(y / vs) * sw];
        }
        if (hs == 2) {
            for (int i = 0; i < iw; i++) t[i] = (vs == 2) ? 3 * near_[i] + far_[i] : 4 * near_[i];
            if (iw == 1) {
                row[0] = row[1] = (uint8_t)((t[0] + 2) >> 2);
            } else {
                row[0] = (uint8_t)((4 * t[0] + 8) >> 4);
                row[1] = (uint8_t)((3 * t[0] + t[1] + 8) >> 4);
                for (int i = 2; i < iw; i++) {
                    row[i * 2 - 2] = (uint8_t)((3 * t[i - 1] + t[i - 2] + 8) >> 4);
                    row[i * 2 - 1] = (uint8_t)((3 * t[i - 1] + t[i] + 8) >> 4);
                }
                row[iw * 2 - 2] = (uint8_t)((3 * t[iw - 1] + t[iw - 2] + 8) >> 4);
                row[iw * 2 - 1] = (uint8_t)((4 * t[iw - 1] + 8) >> 4);
            }
            memcpy(&out[(size_t)y * W], row.data(), W);
        } else {
            for (int x = 0; x < W; x++) {
                int i = x / hs;
                out[(size_t)y * W + x] = (vs == 2) ? (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2) : near_[i];
            }
        }
    }
}

static bool decode(const std::vector<uint8_t> &d, std::vector<uint8_t> &rgba, uint32_t &W, uint32_t &H) {
    if (d.size() < 4 || d[0] != 0xff || d[1] != 0xd8) { set_error("not a JPEG"); return false; }
    uint16_t qt[4][64] = {};
    Huff hdc[4], hac[4];
    std::vector<Comp> comps;
    int restart_interval = 0;
    W = H = 0;
    size_t pos = 2;
    bool done = false;
    while (!done && pos + 4 <= d.size()) {
        if (d[pos] != 0xff) { pos++; continue; }
        uint8_t m = d[pos + 1];
        if (m == 0xff) { pos++; continue; }
        pos += 2;
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
        if (m == 0xd9) break;
        size_t len = (size_t)d[pos] << 8 | d[pos + 1];
        if (len < 2 || pos + len > d.size()) { set_error("JPEG: bad segment length"); return false; }
        const uint8_t *s = &d[pos + 2];
        size_t n = len - 2;
        switch (m) {
            case 0xdb: {  // DQT
                size_t i = 0;
                while (i < n) {
                    int pq = s[i] >> 4, tq = s[i] & 15;
                    i++;
                    if (tq > 3) { set_error("JPEG: bad DQT"); return false; }
                    for (int k = 0; k < 64; k++) {
                        qt[tq][ZIGZAG[k]] = pq ? (uint16_t)(s[i] << 8 | s[i + 1]) : s[i];
                        i += pq ? 2 : 1;
                    }
                }
                break;
            }
            case 0xc0: case 0xc1: {  // SOF0 / SOF1
                if (s[0] != 8) { set_error("JPEG: only 8-bit precision"); return false; }
                H = (uint32_t)s[1] << 8 | s[2];
                W = (uint32_t)s[3] << 8 | s[4];
                int nc = s[5];
                if (nc != 1 && nc != 3) { set_error("JPEG: %d components unsupported", nc); return false; }
                comps.resize(nc);
                for (int i = 0; i < nc; i++) {
                    comps[i].id = s[6 + i * 3];
                    comps[i].h = s[7 + i * 3] >> 4;
                    comps[i].v = s[7 + i * 3] & 15;
                    comps[i].tq = s[8 + i * 3];
                }
                break;
            }
            case 0xc2: case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb:
            case 0xcd: case 0xce: case 0xcf:
                set_error("JPEG: unsupported SOF marker 0x%02x (only baseline)", m);
                return false;
            case 0xc4: {  // DHT
                size_t i = 0;
                while (i + 17 <= n) {
                    int tc = s[i] >> 4, th = s[i] & 15;
                    const uint8_t *counts = &s[i + 1];
                    int total = 0;
                    for (int k = 0; k < 16; k++) total += counts[k];
                    if (th > 3 || i + 17 + total > n) { set_error("JPEG: bad DHT"); return false; }
                    (tc ? hac[th] : hdc[th]).build(counts, &s[i + 17]);
                    i += 17 + total;
                }
                break;
            }
            case 0xdd: restart_interval = s[0] << 8 | s[1]; break;
            case 0xda: {  // SOS — baseline: one interleaved scan with all components
                if (comps.empty()) { set_error("JPEG: SOS before SOF"); return false; }
                int ns = s[0];
                if (ns != (int)comps.size()) { set_error("JPEG: non-interleaved scans unsupported"); return false; }
                for (int i = 0; i < ns; i++) {
                    int cid = s[1 + i * 2];
                    for (auto &c : comps)
                        if (c.id == cid) { c.td = s[2 + i * 2] >> 4; c.ta = s[2 + i * 2] & 15; }
                }
                int hmax = 1, vmax = 1;
                for (auto &c : comps) { if (c.h > hmax) hmax = c.h; if (c.v > vmax) vmax = c.v; }
                int mcux = ((int)W + 8 * hmax - 1) / (8 * hmax), mcuy = ((int)H + 8 * vmax - 1) / (8 * vmax);
                for (auto &c : comps) {
                    c.bw = mcux * c.h; c.bh = mcuy * c.v; c.pred = 0;
                    c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
                    if (!hdc[c.td].present || !hac[c.ta].present) { set_error("JPEG: missing Huffman table"); return false; }
                }
                BitReader br;
                br.p = &d[pos + len];
                br.end = d.data() + d.size();
                int coef[64];
                int mcu_count = 0, total = mcux * mcuy;
                for (int my = 0; my < mcuy; my++) {
                    for (int mx = 0; mx < mcux; mx++) {
                        if (restart_interval && mcu_count && mcu_count % restart_interval == 0) {
                            // byte-align, expect RSTn
                            const uint8_t *q = br.p;
                            while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) q++;
                            br.p = q + 2;
                            br.reset();
                            for (auto &c : comps) c.pred = 0;
                        }
                        for (auto &c : comps) {
                            for (int by = 0; by < c.v; by++) {
                                for (int bx = 0; bx < c.h; bx++) {
                                    memset(coef, 0, sizeof coef);
                                    int t = br.decode(hdc[c.td]);
                                    if (t < 0) { set_error("JPEG: bad DC code"); return false; }
                                    int diff = t ? extend(br.get(t), t) : 0;
                                    c.pred += diff;
                                    coef[0] = c.pred * qt[c.tq][0];
                                    for (int k = 1; k < 64;) {
                                        int rs = br.decode(hac[c.ta]);
                                        if (rs < 0) { set_error("JPEG: bad AC code"); return false; }
                                        int r = rs >> 4, sz = rs & 15;
                                        if (!sz) {
                                            if (r != 15) break;
                                            k += 16;
                                            continue;
                                        }
                                        k += r;
                                        if (k > 63) break;
                                        int z = ZIGZAG[k];
                                        coef[z] = extend(br.get(sz), sz) * qt[c.tq][z];
                                        k++;
                                    }
                                    int ox = (mx * c.h + bx) * 8, oy = (my * c.v + by) * 8;
                                    idct_block(coef, &c.plane[(size_t)oy * c.bw * 8 + ox], c.bw * 8);
                                }
                            }
                        }
                        mcu_count++;
                    }
                }
                (void)total;
                // colour conversion
                rgba.resize((size_t)W * H * 4);
                if (comps.size() == 1) {
                    int sw = comps[0].bw * 8;
                    for (uint32_t y = 0; y < H; y++)
                        for (uint32_t x = 0; x < W; x++) {
                            uint8_t v = comps[0].plane[(size_t)y * sw + x];
                            uint8_t *o = &rgba[((size_t)y * W + x) * 4];
                            o[0] = o[1] = o[2] = v; o[3] = 255;
                        }
                } else {
                    std::vector<uint8_t> pl[3];
                    for (int i = 0; i < 3; i++) upsample(comps[i], hmax, vmax, (int)W, (int)H, pl[i]);
                    for (size_t i = 0; i < (size_t)W * H; i++) {
                        float y = (float)pl[0][i], cb = (float)pl[1][i] - 128.0f, cr = (float)pl[2][i] - 128.0f;
                        float r = y + 1.40200f * cr;
                        float g = y - 0.34414f * cb - 0.71414f * cr;
                        float b = y + 1.77200f * cb;
                        uint8_t *o = &rgba[i * 4];
                        o[0] = clamp8((int)(r + 0.5f)); o[1] = clamp8((int)(g + 0.5f)); o[2] = clamp8((int)(b + 0.5f));
                        o[3] = 255;
                    }
                }
                done = true;
                break;
            }
            default: break;  // APPn, COM, ...
        }
        pos += len;
    }
    if (!done) { set_error("JPEG: no scan decoded"); return false; }
    return true;
}

}  // namespace jpg

// ---------------------------------------------------------------------------------------------- TIFF
// Baseline TIFF 6.0, what the reference's floor textures need (textures/2d/MarbleFloorTiles2/*.tiff): strips, 8-bit
// grey / RGB / RGBA chunky, uncompressed or LZW (compression 5, MSB-first codes with the "early change" of the TIFF
// variant), optional horizontal differencing (predictor 2).  Lossless, so the texels equal what the `image` crate's
// `tiff` decoder hands to Texture::from_path (texture.rs:80-92).
namespace tif {
template <class... A>
static bool fail(const char *fmt, A... a) { set_error(fmt, a...); return false; }
struct Reader {
    const std::vector<uint8_t> &d;
    bool le;
    uint16_t u16(size_t o) const { return o + 2 <= d.size() ? (le ? (uint16_t)(d[o] | d[o + 1] << 8) : (uint16_t)(d[o] << 8 | d[o + 1])) : 0; }
    uint32_t u32(size_t o) const {
        if (o + 4 > d.size()) return 0;
        return le ? (uint32_t)d[o] | (uint32_t)d[o + 1] << 8 | (uint32_t)d[o + 2] << 16 | (uint32_t)d[o + 3] << 24
                  : (uint32_t)d[o] << 24 | (uint32_t)d[o + 1] << 16 | (uint32_t)d[o + 2] << 8 | (uint32_t)d[o + 3];
    }
};
// value k of an IFD entry (type 3 = SHORT, 4 = LONG; inline when it fits in 4 bytes)
static uint32_t entry_value(const Reader &r, size_t entry, uint32_t k) {
    const uint16_t type = r.u16(entry + 2);
    const uint32_t count = r.u32(entry + 4);
    const size_t size = type == 3 ? 2 : 4;
    const size_t base = (size * count <= 4) ? entry + 8 : r.u32(entry + 8);
    return type == 3 ? r.u16(base + 2 * k) : r.u32(base + 4 * k);
}
static bool lzw(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t expect) {
    struct Entry { int prefix; uint8_t byte; uint16_t len; };
    std::vector<Entry> table(4096);
    for (int i = 0; i < 256; i++) table[i] = Entry{-1, (uint8_t)i, 1};
    int next = 258, bits = 9, prev = -1;
    uint32_t acc = 0;
    int nacc = 0;
    size_t pos = 0;
    std::vector<uint8_t> tmp;
    auto emit = [&](int code) {
        tmp.resize(table[code].len);
        for (int c = code, k = (int)table[code].len - 1; c >= 0; c = table[c].prefix, k--) tmp[k] = table[c].byte;
        out.insert(out.end(), tmp.begin(), tmp.end());
    };
    while (out.size() < expect) {
        while (nacc < bits) {
            if (pos >= n) return out.size() >= expect;
            acc = (acc << 8) | src[pos++];
            nacc += 8;
        }
        int code = (int)((acc >> (nacc - bits)) & ((1u << bits) - 1u));
        nacc -= bits;
        if (code == 257) break;                         // EndOfInformation
        if (code == 256) { next = 258; bits = 9; prev = -1; continue; }   // ClearCode
        if (prev < 0) {
            if (code >= 256) return false;
            emit(code);
        } else {
            if (code < next) {
                emit(code);
                int c = code;
                while (table[c].prefix >= 0) c = table[c].prefix;
                if (next < 4096) table[next++] = Entry{prev, table[c].byte, (uint16_t)(table[prev].len + 1)};
            } else if (code == next && next < 4096) {
                int c = prev;
                while (table[c].prefix >= 0) c = table[c].prefix;
                table[next++] = Entry{prev, table[c].byte, (uint16_t)(table[prev].len + 1)};
                emit(code);
            } else {
                return false;
            }
            if (next == 511 || next == 1023 || next == 2047) bits++;   // early change: one code before the table is full
        }
        prev = code;
    }
    return out.size() >= expect;
}
static bool decode(const std::vector<uint8_t> &d, std::vector<uint8_t> &rgba, uint32_t &W, uint32_t &H) {
    if (d.size() < 8) return fail("tiff: truncated");
    Reader r{d, d[0] == 'I'};
    if (r.u16(2) != 42) return fail("tiff: bad magic");
    size_t ifd = r.u32(4);
    const uint16_t n = r.u16(ifd);
    uint32_t compression = 1, photometric = 2, spp = 1, rows_per_strip = 0xffffffffu, predictor = 1, planar = 1, bits = 8;
    size_t e_offsets = 0, e_counts = 0;
    uint32_t n_strips = 0;
    W = H = 0;
    for (uint16_t i = 0; i < n; i++) {
        const size_t e = ifd + 2 + 12 * (size_t)i;
        const uint16_t tag = r.u16(e);
        switch (tag) {
            case 256: W = entry_value(r, e, 0); break;
            case 257: H = entry_value(r, e, 0); break;
            case 258: bits = entry_value(r, e, 0); break;
            case 259: compression = entry_value(r, e, 0); break;
            case 262: photometric = entry_value(r, e, 0); break;
            case 273: e_offsets = e; n_strips = r.u32(e + 4); break;
            case 277: spp = entry_value(r, e, 0); break;
            case 278: rows_per_strip = entry_value(r, e, 0); break;
            case 279: e_counts = e; break;
            case 284: planar = entry_value(r, e, 0); break;
            case 317: predictor = entry_value(r, e, 0); break;
            default: break;
        }
    }
    if (!W || !H || !e_offsets || !e_counts) return fail("tiff: missing dimension or strip tags");
    if (bits != 8 || planar != 1 || spp < 1 || spp > 4 || (compression != 1 && compression != 5) || predictor > 2 || photometric > 2)
        return fail("tiff: unsupported layout (bits %u, planar %u, samples %u, compression %u, predictor %u)", bits, planar, spp, compression, predictor);
    if (rows_per_strip > H) rows_per_strip = H;
    rgba.assign((size_t)W * H * 4, 255);
    std::vector<uint8_t> strip;
    for (uint32_t si = 0; si < n_strips; si++) {
        const uint32_t y0 = si * rows_per_strip;
        if (y0 >= H) break;
        const uint32_t rows = std::min(rows_per_strip, H - y0);
        const size_t off = entry_value(r, e_offsets, si), cnt = entry_value(r, e_counts, si);
        if (off + cnt > d.size()) return fail("tiff: strip %u out of range", si);
        const size_t expect = (size_t)rows * W * spp;
        strip.clear();
        if (compression == 1) {
            if (cnt < expect) return fail("tiff: short strip %u", si);
            strip.assign(d.begin() + off, d.begin() + off + expect);
        } else if (!lzw(d.data() + off, cnt, strip, expect)) {
            return fail("tiff: LZW error in strip %u", si);
        }
        for (uint32_t y = 0; y < rows; y++) {
            uint8_t *row = &strip[(size_t)y * W * spp];
            if (predictor == 2)
                for (size_t x = spp; x < (size_t)W * spp; x++) row[x] = (uint8_t)(row[x] + row[x - spp]);
            uint8_t *dst = &rgba[((size_t)(y0 + y) * W) * 4];
            for (uint32_t x = 0; x < W; x++) {
                const uint8_t *px = row + (size_t)x * spp;
                if (spp >= 3) { dst[4 * x] = px[0]; dst[4 * x + 1] = px[1]; dst[4 * x + 2] = px[2]; if (spp == 4) dst[4 * x + 3] = px[3]; }
                else {
                    uint8_t g = photometric == 0 ? (uint8_t)(255 - px[0]) : px[0];
                    dst[4 * x] = dst[4 * x + 1] = dst[4 * x + 2] = g;
                    if (spp == 2) dst[4 * x + 3] = px[1];
                }
            }
        }
    }
    return true;
}
}  // namespace tif

}  // namespace hh

extern "C" {

const char *hh_last_error(void) { return hh::g_err.c_str(); }

int hh_decode_image(const char *path, uint8_t **rgba, uint32_t *width, uint32_t *height) {
    if (!path || !rgba || !width || !height) { hh::set_error("hh_decode_image: null argument"); return HR_ERR_INVALID; }
    std::vector<uint8_t> d, px;
    if (!hh::read_file(path, d)) return HR_ERR_INVALID;
    bool ok;
    if (d.size() >= 2 && d[0] == 0xff && d[1] == 0xd8) ok = hh::jpg::decode(d, px, *width, *height);
    else if (d.size() >= 4 && ((d[0] == 'I' && d[1] == 'I') || (d[0] == 'M' && d[1] == 'M'))) ok = hh::tif::decode(d, px, *width, *height);
    else ok = hh::decode_png(d, px, *width, *height);
    if (!ok) return HR_ERR_INVALID;
    *rgba = (uint8_t *)malloc(px.size());
    memcpy(*rgba, px.data(), px.size());
    return HR_OK;
}

int hh_write_png_rgb8(const char *path, const uint8_t *rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || !width || !height) { hh::set_error("hh_write_png_rgb8: bad argument"); return HR_ERR_INVALID; }
    return hh::write_png_rgb8(path, rgb, width, height) ? HR_OK : HR_ERR_INVALID;
}

void hh_free(void *p) { free(p); }

}
