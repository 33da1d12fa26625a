// accuracy probe for the hardware transcendentals used by the trace kernel (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *x, float *s, float *c, float *p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = __builtin_amdgcn_sinf(x[i]);
    c[i] = __builtin_amdgcn_cosf(x[i]);
    p[i] = __builtin_amdgcn_exp2f(2.2f * __builtin_amdgcn_logf(x[i]));
}
int main() {
    const int n = 1 << 20;
    std::vector<float> x(n), s(n), c(n), p(n);
    for (int i = 0; i < n; i++) x[i] = (float)((i + 0.37) / n);
    float *dx, *ds, *dc, *dp;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dp, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, dp, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, ep = 0;
    for (int i = 0; i < n; i++) {
        double ph = 2 * M_PI * (double)x[i];
        es = fmax(es, fabs(s[i] - sin(ph))); ec = fmax(ec, fabs(c[i] - cos(ph)));
        double r = pow((double)x[i], 2.2);
        ep = fmax(ep, fabs(p[i] - r) / fmax(r, 1e-30));
    }
    printf("max abs err v_sin_f32 %.3e  v_cos_f32 %.3e   max rel err x^2.2 via v_exp/v_log %.3e\n", es, ec, ep);
    return 0;
}
