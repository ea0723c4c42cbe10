// Micro-benchmark: do row stores overlap with MFMA work issued by the same waves?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_overlap store_overlap.hip && /tmp/store_overlap
// A workgroup of 8 waves owns 256 points per tile; per "layer" a wave issues NM MFMAs (operands in registers, four
// independent accumulators) and then its 16 row stores of 1 KiB (the training forward's pattern).  Three runs: MFMAs only,
// stores only, both.  If the machine overlapped them, "both" would take the longer of the two.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool MF, bool ST>
__global__ __launch_bounds__(512, 2) void k(char *base, const bf16x8 *in, float *out, long n_points, int layers, int nm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const long tiles = n_points / 256;
    bf16x8 a0 = in[lane], a1 = in[64 + lane], b0 = in[128 + lane], b1 = in[192 + lane];
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        for (int l = 0; l < layers; ++l) {
            if (MF) {
                for (int i = 0; i < nm; i += 4) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c3, 0, 0, 0);
                }
            }
            if (ST) {
                char *lay = base + (long)l * n_points * 512;
                const long p0 = t * 256 + wave * 32;
                const u32x4 v = {(unsigned)lane, __builtin_bit_cast(unsigned, c0[0]), 2u, 3u};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
                        *reinterpret_cast<u32x4 *>(lay + (p0 + cc * 16 + col) * 512 + kk * 64 + q * 16) = v;
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    const long n = 196608;
    const int layers = 10, nm = 256;          // 256 MFMAs per wave and layer = the field kernel's 128 fragments x 2
    char *buf; bf16x8 *in; float *out;
    hipMalloc(&buf, (size_t)layers * n * 512);
    hipMalloc(&in, 256 * 16); hipMemset(in, 0x3c, 256 * 16);
    hipMalloc(&out, 1024 * 512 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const char *names[3] = {"MFMA only ", "stores only", "both       "};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL((k<true, false>), dim3(256), dim3(512), 0, 0, buf, in, out, n, layers, nm);
            if (mode == 1) hipLaunchKernelGGL((k<false, true>), dim3(256), dim3(512), 0, 0, buf, in, out, n, layers, nm);
            if (mode == 2) hipLaunchKernelGGL((k<true, true>), dim3(256), dim3(512), 0, 0, buf, in, out, n, layers, nm);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%s: %.1f us\n", names[mode], best * 1e3);
    }
    return 0;
}
