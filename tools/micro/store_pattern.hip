// Micro-benchmark: HBM write rate of the training forward's row-store pattern, without any compute.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern store_pattern.hip && /tmp/store_pattern
// A workgroup of 8 waves owns 256 points; per "layer" every wave stores 16 x 16 B per lane: lane (col, q) writes 16 bytes at
// row(point) * 512 + k * 64 + q * 16 -- 16 points x 64 contiguous bytes per instruction (mode 0, the kernel's pattern), or the
// same bytes as 1-KiB contiguous pieces per instruction (mode 1: lane l writes piece base + 16 l), or (mode 2) the kernel's
// pattern with both halves of a 128-byte line in ONE instruction (lane pairs cover k, k+1: 8 points x 128 B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(char *base, long n_points, int layers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const long tiles = n_points / 256;
    const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        for (int l = 0; l < layers; ++l) {
            char *lay = base + (long)l * n_points * 512;
            const long p0 = t * 256 + wave * 32;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    char *dst;
                    if (MODE == 0) dst = lay + (p0 + cc * 16 + col) * 512 + kk * 64 + q * 16;
                    else if (MODE == 1) dst = lay + (p0 * 512) + (kk * 2 + cc) * 1024 + lane * 16;
                    else dst = lay + (p0 + cc * 16 + (lane >> 3) + 8 * (kk & 1)) * 512 + (kk >> 1) * 128 + (lane & 7) * 16;
                    *reinterpret_cast<u32x4 *>(dst) = v;
                }
        }
    }
}

int main() {
    const long n = 196608;
    const int layers = 10;
    char *buf;
    hipMalloc(&buf, (size_t)layers * n * 512);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {256, 768}) {
            float best = 1e9;
            for (int r = 0; r < 5; ++r) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, buf, n, layers);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, buf, n, layers);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, buf, n, layers);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("mode %d grid %4d: %.1f us, %.2f TB/s\n", mode, grid, best * 1e3, (double)layers * n * 512 / (best * 1e-3) / 1e12);
        }
    return 0;
}
