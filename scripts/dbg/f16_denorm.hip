// Does v_mfma_f32_16x16x32_f16 keep subnormal f16 inputs?  A[i][k] = 2^-20 (subnormal half), B[k][j] = 1024 -> each product 2^-10, the
// sum over K = 32 is 2^-5 = 0.03125 when subnormals are honoured and 0 when they are flushed.  Also prints what (_Float16)x rounds to
// for a few values (round-to-nearest-even expected) and whether a - float(half(a)) is exact.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float* cv) {
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)9.5367431640625e-07f; b[e] = (_Float16)1024.f; }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
    // subnormal x subnormal-free: A = 2^-20, B = 2^-4 -> products 2^-24 (below f16 range, fine in f32): sum 2^-19
    for (int e = 0; e < 8; ++e) b[e] = (_Float16)0.0625f;
    acc = f4{0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[1] = acc[0];
    if (threadIdx.x == 0) {
        const float xs[4] = {0.1f, 3.0004883f, 65519.f, 1e-6f};
        for (int i = 0; i < 4; ++i) { const _Float16 h = (_Float16)xs[i]; cv[2 * i] = (float)h; cv[2 * i + 1] = xs[i] - (float)h; }
    }
}
int main() {
    float *d, *c, h[2], hc[8];
    hipMalloc(&d, 8); hipMalloc(&c, 32);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 32, hipMemcpyDeviceToHost);
    printf("mfma f16 with subnormal A (2^-20) x 1024: got %g, 0.03125 if subnormal inputs are kept, 0 if flushed\n", h[0]);
    printf("mfma f16 with subnormal A (2^-20) x 2^-4: got %g, expected 1.9073486328125e-06\n", h[1]);
    for (int i = 0; i < 4; ++i) printf("half(x) = %.9g, x - half(x) = %.9g\n", hc[2 * i], hc[2 * i + 1]);
    return 0;
}
