// fp32 post chain (renderer.rs:64-90): scale -> Reinhard (tonemap.rs:22-27) -> gamma (color.rs:38-48)
// -> bilateral 3x3 (filter.rs:32-58, with the release-build u32 wrap quirks) -> u8 truncation (color.rs:10-16).
#pragma once
#include <math.h>

#include "device_scene.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define HR_POST_POWF(x, y) __builtin_amdgcn_exp2f((y) * __builtin_amdgcn_logf(x))
#else
#define HR_POST_POWF(x, y) powf((x), (y))
#endif

namespace hr {

HD void tonemap_gamma(float r, float g, float b, float scale, float *out) {
    const float exposure = 1.5f, white = 20.0f * 1.5f;  // config.rs:18-19
    float cr = r * scale * exposure, cg = g * scale * exposure, cb = b * scale * exposure;
    float lum = 0.22f * cr + 0.707f * cg + 0.071f * cb;  // color.rs:63-65
    float k = (lum / (white * white) + 1.0f) / (lum + 1.0f);
    const float inv_gamma = 1.0f / 2.2f;
    out[0] = HR_POST_POWF(fminf(fmaxf(cr * k, 0.0f), 1.0f), inv_gamma);
    out[1] = HR_POST_POWF(fminf(fmaxf(cg * k, 0.0f), 1.0f), inv_gamma);
    out[2] = HR_POST_POWF(fminf(fmaxf(cb * k, 0.0f), 1.0f), inv_gamma);
}

HD float gaussianf(float x, float sigma) { return expf(-(x * x) / (2.0f * sigma * sigma)) / (2.0f * 3.14159265358979f * sigma * sigma); }

// img: W*H*3 floats (tone-mapped, gamma).  Writes 3 bytes.
HD void bilateral_quantise(const float *img, uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint8_t *out) {
    const float *c = &img[((size_t)y * W + x) * 3];
    float csum = c[0] + c[1] + c[2];
    float fr = 0.f, fg = 0.f, fb = 0.f, wp = 0.f;
    for (uint32_t i = 0; i < 3; i++)
        for (uint32_t j = 0; j < 3; j++) {
            uint32_t nx = x - (1u - i), ny = y - (1u - j);  // wrapping u32, then clamp_u32 (math.rs:13-15)
            nx = nx > W - 1 ? W - 1 : nx;
            ny = ny > H - 1 ? H - 1 : ny;
            const float *n = &img[((size_t)ny * W + nx) * 3];
            float nsum = n[0] + n[1] + n[2];
            uint32_t dx = x - nx, dy = y - ny;
            float dist = sqrtf((float)(uint32_t)(dx * dx + dy * dy));
            float w = gaussianf((1.0f / 3.0f) * (nsum - csum), 1.0f) * gaussianf(dist, 16.0f);
            fr += n[0] * w; fg += n[1] * w; fb += n[2] * w;
            wp += w;
        }
    out[0] = (uint8_t)(255.0f * fminf(fmaxf(fr / wp, 0.0f), 1.0f));
    out[1] = (uint8_t)(255.0f * fminf(fmaxf(fg / wp, 0.0f), 1.0f));
    out[2] = (uint8_t)(255.0f * fminf(fmaxf(fb / wp, 0.0f), 1.0f));
}

}  // namespace hr
