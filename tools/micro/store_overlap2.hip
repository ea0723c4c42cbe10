// Micro-benchmark 2: the training forward's skeleton -- MFMA work, a weight ring fed by LDS-DMA with counted vmcnt waits
// and a barrier per 16 "fragments", and a burst of 16 row stores per layer -- to see which ingredient serialises stores
// and compute.   hipcc --offload-arch=gfx950 -O3 -o /tmp/so2 store_overlap2.hip && /tmp/so2
// WAITMODE 0: vmcnt(2) like the kernel without a ledger (the stores are the youngest entries of the in-order queue)
//          1: vmcnt(2 + stores issued since the awaited block's DMA) = the ledger
//          2: no vmcnt wait at all before the barrier (timing only: the ring may be read before it lands)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void dma_piece(const char *g, uint32_t lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_base) : "memory");
}

template <bool ST, bool DMA, int WAITMODE, bool PACED = false, bool NT = false, bool REG = false>
__global__ __launch_bounds__(512, 2) void k(char *base, const char *wstream, float *out, long n_points, int layers) {
    extern __shared__ __attribute__((aligned(16))) char ring[];       // 4 blocks x 16 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, q = lane >> 4;
    const long tiles = n_points / 256;
    const uint32_t ring_u32 = (uint32_t)(uintptr_t)ring;
    bf16x8 b0, b1;
    for (int j = 0; j < 8; ++j) { b0[j] = (__bf16)(0.01f * lane); b1[j] = (__bf16)(0.02f * lane); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    constexpr int NB = 73;                                            // 1168 fragments of 1 KiB: the model's stream
    u32x4 stage0, stage1;                                             // REG: one block in flight through registers
    auto issue = [&](int blk) {                                       // this wave's 2 pieces of a 16-KiB block
        if (!DMA) return;
        if (REG) {
            const char *g = wstream + ((long)(blk % NB) * 16 + wave * 2) * 1024 + lane * 16;
            stage0 = *reinterpret_cast<const u32x4 *>(g);
            stage1 = *reinterpret_cast<const u32x4 *>(g + 1024);
            return;
        }
        const char *g = wstream + ((long)(blk % NB) * 16 + wave * 2) * 1024 + lane * 16;
        const uint32_t l = ring_u32 + (blk & 3) * 16384 + wave * 2048;
        dma_piece(g, l); dma_piece(g + 1024, l + 1024);
    };
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        int blk = 0;
        if (REG) issue(1); else { issue(0); issue(1); issue(2); }
        for (int l = 0; l < layers; ++l) {
#pragma unroll 1
            for (int b = 0; b < 8; ++b, ++blk) {                     // 8 blocks of 16 fragments per layer
                if (DMA && REG) {                                     // block blk + 1 arrives in registers: write it to its slot, then sync
                    char *slot = ring + ((blk + 1) & 3) * 16384 + wave * 2048 + lane * 16;
                    *reinterpret_cast<u32x4 *>(slot) = stage0;
                    *reinterpret_cast<u32x4 *>(slot + 1024) = stage1;
                    __syncthreads();
                    issue(blk + 2);
                } else if (DMA) {
                    if (WAITMODE == 0) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
                    else if (WAITMODE == 1) {
                        if (PACED) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");          // 2 blocks x (2 DMA + 2 stores) younger
                        else if (b < 3 && l > 0) asm volatile("s_waitcnt vmcnt(20)\n\ts_barrier" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
                    }
                    else asm volatile("s_barrier" ::: "memory");
                    issue(blk + 3);
                }
#pragma unroll
                for (int f = 0; f < 16; ++f) {
                    const bf16x8 a = DMA ? *reinterpret_cast<const bf16x8 *>(ring + (blk & 3) * 16384 + f * 1024 + lane * 16) : b1;
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, c1, 0, 0, 0);
                }
                if (ST && PACED) {                                   // this block's share of the layer's rows: 2 stores
                    char *lay = base + (long)l * n_points * 512;
                    const long p0 = t * 256 + wave * 32;
                    const u32x4 v = {(unsigned)lane, __builtin_bit_cast(unsigned, c0[0]), 2u, 3u};
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
                        *reinterpret_cast<u32x4 *>(lay + (p0 + cc * 16 + col) * 512 + b * 64 + q * 16) = v;
                }
            }
            if (ST && !PACED) {
                char *lay = base + (long)l * n_points * 512;
                const long p0 = t * 256 + wave * 32;
                const u32x4 v = {(unsigned)lane, __builtin_bit_cast(unsigned, c0[0]), 2u, 3u};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        u32x4 *dst = reinterpret_cast<u32x4 *>(lay + (p0 + cc * 16 + col) * 512 + kk * 64 + q * 16);
                        if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
                    }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

template <bool ST, bool DMA, int W, bool PACED = false, bool NT = false, bool REG = false>
float run(char *buf, char *ws, float *out, long n, int layers) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<ST, DMA, W, PACED, NT, REG>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<ST, DMA, W, PACED, NT, REG>), dim3(256), dim3(512), 65536, 0, buf, ws, out, n, layers);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    const long n = 196608;
    const int layers = 10;
    char *buf, *ws; float *out;
    hipMalloc(&buf, (size_t)layers * n * 512);
    hipMalloc(&ws, 73 * 16384 + 4096); hipMemset(ws, 0x3c, 73 * 16384 + 4096);
    hipMalloc(&out, 1024 * 512 * 4);
    printf("MFMA only (operands in registers)      : %.1f us\n", run<false, false, 0>(buf, ws, out, n, layers));
    printf("MFMA + weight ring (vmcnt(4) + barrier) : %.1f us\n", run<false, true, 0>(buf, ws, out, n, layers));
    printf("MFMA + stores, no ring                  : %.1f us\n", run<true, false, 0>(buf, ws, out, n, layers));
    printf("MFMA + ring + stores, plain counts      : %.1f us\n", run<true, true, 0>(buf, ws, out, n, layers));
    printf("MFMA + ring + stores, ledger counts     : %.1f us\n", run<true, true, 1>(buf, ws, out, n, layers));
    printf("MFMA + ring + stores, no vmcnt waits    : %.1f us\n", run<true, true, 2>(buf, ws, out, n, layers));
    printf("paced stores + ring, plain counts       : %.1f us\n", run<true, true, 0, true>(buf, ws, out, n, layers));
    printf("paced stores + ring, ledger counts      : %.1f us\n", run<true, true, 1, true>(buf, ws, out, n, layers));
    printf("paced stores + ring, no vmcnt waits     : %.1f us\n", run<true, true, 2, true>(buf, ws, out, n, layers));
    printf("ring through registers, no stores       : %.1f us\n", run<false, true, 0, false, false, true>(buf, ws, out, n, layers));
    printf("ring through registers + stores         : %.1f us\n", run<true, true, 0, false, false, true>(buf, ws, out, n, layers));
    printf("non-temporal stores + ring, plain counts: %.1f us\n", run<true, true, 0, false, true>(buf, ws, out, n, layers));
    printf("non-temporal stores, no ring            : %.1f us\n", run<true, false, 0, false, true>(buf, ws, out, n, layers));
    printf("paced stores, no ring                   : %.1f us\n", run<true, false, 0, true>(buf, ws, out, n, layers));
    return 0;
}
