// Post chain kernels: renderer.rs:64-90 — included by hr_api.hip only (one translation unit: the kernels and the C ABI that launches them).
#pragma once
#include <hip/hip_runtime.h>

#include "device_scene.h"
#include "post_core.h"

using namespace hr;

__global__ void tonemap_gamma_kernel(const float *__restrict__ acc, float *__restrict__ out, uint32_t n, float scale) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tonemap_gamma(acc[i * 3], acc[i * 3 + 1], acc[i * 3 + 2], scale, &out[i * 3]);
}
__global__ void bilateral_quantise_kernel(const float *__restrict__ img, uint8_t *__restrict__ out, uint32_t W, uint32_t H) {
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    bilateral_quantise(img, W, H, x, y, &out[((size_t)y * W + x) * 3]);
}
