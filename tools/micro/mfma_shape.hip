// Micro-benchmark: sustained bf16 MFMA rate of the two gfx950 shapes on random data,
// operands in registers (optionally one ds_read_b128 per A fragment), 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, int LDS>
__global__ __launch_bounds__(512, 2) void k(const bf16x8 *in, float *out, int iters) {
    __shared__ bf16x8 sm[64 * 16];
    const int lane = threadIdx.x & 63;
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(i * 64 + lane) % 4096]; b[i] = in[((i + 8) * 64 + lane) % 4096]; }
    for (int i = threadIdx.x; i < 64 * 16; i += 512) sm[i] = in[i];
    __syncthreads();
    if constexpr (SHAPE == 32) {
        f32x16 acc0 = {}, acc1 = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16x8 w = LDS ? sm[((it + i) & 15) * 64 + lane] : a[i];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b[(i + 1) & 7], acc1, 0, 0, 0);
            }
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x4 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // same FLOPs per iteration: 4 x (16x16x32) = 2 x (32x32x16) / 2 ... doubled below
                bf16x8 w = LDS ? sm[((it + i) & 15) * 64 + lane] : a[i];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b[(i + 1) & 7], acc1, 0, 0, 0);
                bf16x8 w2 = LDS ? sm[((it + i + 8) & 15) * 64 + lane] : a[(i + 3) & 7];
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, b[i], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, b[(i + 1) & 7], acc3, 0, 0, 0);
            }
        }
        float s = 0; for (int i = 0; i < 4; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

template <int SHAPE, int LDS>
double run(const bf16x8 *in, float *out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(256 * 4), dim3(512), 0, 0, in, out, iters);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(256 * 4), dim3(512), 0, 0, in, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // FLOP per wave per iteration: 16 MFMAs x 32768 (SHAPE 32) or 32 MFMAs x 16384 (SHAPE 16)
    double flop = (double)reps * 256 * 4 * 8 * iters * 16.0 * 32768.0;
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto &v : h) { float f = (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    bf16x8 *in; float *out;
    hipMalloc(&in, h.size() * 2); hipMalloc(&out, 256 * 4 * 512 * 4);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int round = 0; round < 3; ++round) {
        printf("round %d: 32x32x16 reg %.0f TF | 16x16x32 reg %.0f TF | 32x32x16 lds %.0f TF | 16x16x32 lds %.0f TF\n", round,
               run<32, 0>(in, out, 400), run<16, 0>(in, out, 400), run<32, 1>(in, out, 400), run<16, 1>(in, out, 400));
    }
    return 0;
}
