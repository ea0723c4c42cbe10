// Does v_mfma_f32_16x16x32_f16 flush fp16 denormal inputs?  (mlp_split.hip keeps its operands out of the denormal range
// with one compare + select per value; if the matrix pipe honours denormals that guard could go.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dn tools/micro/mfma_f16_denorm.hip && /tmp/dn
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float a_val, float b_val, float *out) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float *out; hipMalloc(&out, 4);
    const float tests[][2] = {{1.0f, 1.0f}, {3.0517578125e-05f /* 2^-15: denormal */, 1024.0f}, {5.9604644775390625e-08f /* 2^-24: smallest */, 16384.0f},
                              {6.103515625e-05f /* 2^-14: smallest normal */, 1024.0f}};
    for (auto &t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, t[0], t[1], out);
        float h; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
        printf("a = %.10g  b = %g : sum over k = 32 -> %.10g   (exact %.10g)\n", t[0], t[1], h, 32.0 * t[0] * t[1]);
    }
    return 0;
}
