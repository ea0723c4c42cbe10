// dw_stream.hip -- what bounds the weight-gradient products' row streaming (backward.hip dw2_body)?  Round 4 measured that
// the kernel takes the same time with its LDS reads and MFMAs removed (nerf_amd_set_tuning(0, 63): 359 vs 339 us), i.e. the
// ring itself -- LDS-DMA of 32-point chunks of two row arrays, one counted wait + one barrier per chunk -- delivers ~4.2 TB/s
// where MI355X_MICROARCH.md reports ~6 TB/s for LDS-DMA streaming.  This is that ring alone, with knobs:
//   V & 1   lane-linear source addresses instead of the XOR swizzle
//   V & 2   no workgroup barrier: every wave waits for and re-issues its own pieces only
//   V & 4   buffer_load ... lds instead of global_load_lds (one scalar offset per chunk, one VGPR offset per piece)
//   V & 8   one array of 1024-B rows instead of two of 512 B (same bytes per chunk)
//   V & 16  nt policy
//   NS      ring slots (chunks of 32 KiB)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dw_stream.hip -o /tmp/dw_stream && /tmp/dw_stream
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)(uintptr_t)base;
    rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

template <bool NT>
__device__ __forceinline__ void dma_global(const char *g, uint32_t lds) {
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void dma_buffer(unsigned voff, rsrc_t rs, unsigned soff, uint32_t lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds), "s"(soff) : "memory");
}

__device__ __forceinline__ int swz(int r) { return 2 * ((r & 3) | (((r >> 3) & 1) << 2)); }

template <int V, int NS>
__global__ __launch_bounds__(512, 2) void stream_kernel(const char *A, const char *B, int64_t n_chunks, float *sink) {
    constexpr int IMG = 32768, CNT = 4;                      // pieces per wave and chunk (32 pieces of 1 KiB)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ring = (uint32_t)(uintptr_t)smem;
    const int wg = blockIdx.x, nwg = gridDim.x;
    const int64_t n_local = wg < n_chunks ? (n_chunks - wg + nwg - 1) / nwg : 0;
    const rsrc_t rsA = make_rsrc(A, (unsigned)(n_chunks * 16384)), rsB = make_rsrc(B, (unsigned)(n_chunks * 16384));
    auto issue = [&](int64_t i) {
        int64_t ch = wg + i * nwg;
        if (ch >= n_chunks) ch = n_chunks - 1;
        const uint32_t slot = ring + (uint32_t)(i % NS) * IMG;
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            const int j = wave + 8 * k;                       // piece 0..31: 0..15 array A, 16..31 array B
            const bool isA = (V & 8) ? true : j < 16;
            const int jj = (V & 8) ? j : (isA ? j : j - 16);
            const int e = 64 * jj + lane;
            const int pr = (V & 8) ? 64 : 32;                 // 16-byte pieces per row
            const int r = e / pr, pos = e % pr;
            const int p2 = (V & 1) ? pos : (pos ^ swz(r));
            const unsigned off = (unsigned)(r * pr * 16 + (p2 << 4));
            const int64_t base = (V & 8) ? ch * 32768 : ch * 16384;
            if constexpr ((V & 4) != 0) {
                dma_buffer(off, isA ? rsA : rsB, __builtin_amdgcn_readfirstlane((unsigned)base), slot + 1024 * j);
            } else {
                const char *src = (isA ? A : B) + base + off;
                dma_global<(V & 16) != 0>(src, slot + 1024 * j);
            }
        }
    };
    float acc = 0.f;
    if (n_local > 0) {
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) issue(i);
        for (int64_t i = 0; i < n_local; ++i) {
            if constexpr ((V & 2) != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * CNT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * CNT) : "memory");
            issue(i + NS - 1);
            // touch one dword of this wave's own freshly landed piece so that nothing is optimised away
            acc += *(__attribute__((address_space(3))) const float *)(uintptr_t)(ring + (uint32_t)(i % NS) * IMG + 1024 * wave + lane * 4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 123.456f) sink[0] = acc;
}

// The one-launch path of backward.hip (DwSeq::flush): J products, each with its own pair of row arrays, ~256 / J workgroups
// per product reading chunks wg, wg + nb, ...: J x 2 windows of nb x 16 KiB that advance in lockstep, the arrays
// n_chunks x 16 KiB apart.  rot: product j starts at chunk j n / J and wraps; skew: the arrays are skew bytes further apart
template <int NS>
__global__ __launch_bounds__(512, 2) void jobs_kernel(const char *base, int64_t n_chunks, int J, int rot, int64_t plane, float *sink) {
    constexpr int IMG = 32768, CNT = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ring = (uint32_t)(uintptr_t)smem;
    const int G = gridDim.x;
    const int job = (int)((int64_t)blockIdx.x * J / G);
    const int b0 = (job * G + J - 1) / J, b1 = ((job + 1) * G + J - 1) / J;     // blocks [b0, b1) belong to this job
    const int wg = blockIdx.x - b0, nwg = b1 - b0;
    const char *A = base + (int64_t)(2 * job) * plane, *B = A + plane;
    const int64_t n_local = wg < n_chunks ? (n_chunks - wg + nwg - 1) / nwg : 0;
    const int64_t r0 = rot ? (int64_t)job * n_chunks / J : 0;
    auto issue = [&](int64_t i) {
        int64_t ch = wg + i * nwg;
        if (ch >= n_chunks) ch = n_chunks - 1;
        ch += r0;
        if (ch >= n_chunks) ch -= n_chunks;
        const uint32_t slot = ring + (uint32_t)(i % NS) * IMG;
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            const int j = wave + 8 * k;
            const bool isA = j < 16;
            const int jj = isA ? j : j - 16;
            const int e = 64 * jj + lane;
            const int r = e / 32, pos = e % 32;
            const unsigned off = (unsigned)(r * 512 + ((pos ^ swz(r)) << 4));
            dma_global<false>((isA ? A : B) + ch * 16384 + off, slot + 1024 * j);
        }
    };
    float acc = 0.f;
    if (n_local > 0) {
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) issue(i);
        for (int64_t i = 0; i < n_local; ++i) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * CNT) : "memory");
            issue(i + NS - 1);
            acc += *(__attribute__((address_space(3))) const float *)(uintptr_t)(ring + (uint32_t)(i % NS) * IMG + 1024 * wave + lane * 4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 123.456f) sink[0] = acc;
}

static int run_jobs(const char *base, int64_t n_chunks, int J, int rot, int64_t skew, float *sink, const char *what) {
    constexpr int NS = 4;
    const size_t lds = (size_t)NS * 32768;
    const int64_t plane = n_chunks * 16384 + skew;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(jobs_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((jobs_kernel<NS>), dim3(256), dim3(512), lds, 0, base, n_chunks, J, rot, plane, sink);
    CK(hipEventRecord(a, 0));
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((jobs_kernel<NS>), dim3(256), dim3(512), lds, 0, base, n_chunks, J, rot, plane, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    printf("J=%2d rot=%d skew=%-8lld %-52s %.3f ms  %.2f TB/s\n", J, rot, (long long)skew, what, ms, (double)J * n_chunks * 32768 / (ms * 1e-3) / 1e12);
    return 0;
}

template <int V, int NS>
static int run(const char *A, const char *B, int64_t n_chunks, float *sink, const char *what) {
    const size_t lds = (size_t)NS * 32768;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_kernel<V, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((stream_kernel<V, NS>), dim3(256), dim3(512), lds, 0, A, B, n_chunks, sink);
    CK(hipEventRecord(a, 0));
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((stream_kernel<V, NS>), dim3(256), dim3(512), lds, 0, A, B, n_chunks, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    printf("V=%2d NS=%d  %-58s %.3f ms  %.2f TB/s\n", V, NS, what, ms, (double)n_chunks * 32768 / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    const int64_t n_chunks = 65536;                           // 2 GiB in all: far beyond the 256 MB Infinity Cache
    char *A, *B;
    float *sink;
    CK(hipMalloc(&A, (size_t)n_chunks * 32768));
    B = A + (size_t)n_chunks * 16384;
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(A, 1, (size_t)n_chunks * 32768));
    run<0, 4>(A, B, n_chunks, sink, "as dw2_body: swizzled global DMA, barrier per chunk");
    run<1, 4>(A, B, n_chunks, sink, "lane-linear addresses");
    run<2, 4>(A, B, n_chunks, sink, "no barrier (waves independent)");
    run<3, 4>(A, B, n_chunks, sink, "lane-linear, no barrier");
    run<4, 4>(A, B, n_chunks, sink, "buffer_load lds, swizzled, barrier");
    run<6, 4>(A, B, n_chunks, sink, "buffer_load lds, no barrier");
    run<7, 4>(A, B, n_chunks, sink, "buffer_load lds, lane-linear, no barrier");
    run<8, 4>(A, B, n_chunks, sink, "one array of 1-KiB rows");
    run<16, 4>(A, B, n_chunks, sink, "nt");
    run<18, 4>(A, B, n_chunks, sink, "nt, no barrier");
    run<0, 3>(A, B, n_chunks, sink, "NS 3");
    run<0, 2>(A, B, n_chunks, sink, "NS 2");
    run<2, 3>(A, B, n_chunks, sink, "NS 3, no barrier");
    CK(hipFree(A));
    // the one-launch arrangement: 13 products x 2 arrays of 6144 chunks (196608 points, the fine network of a 1024-ray step)
    const int64_t nc = 6144;
    char *base;
    CK(hipMalloc(&base, (size_t)26 * (nc * 16384 + (1 << 20))));
    CK(hipMemset(base, 1, (size_t)26 * (nc * 16384 + (1 << 20))));
    run_jobs(base, nc, 13, 0, 0, sink, "as DwSeq::flush: arrays 96 MiB apart, lockstep");
    run_jobs(base, nc, 13, 1, 0, sink, "products start at different chunks");
    run_jobs(base, nc, 13, 0, 4096 + 256, sink, "arrays 4.25 KiB further apart");
    run_jobs(base, nc, 13, 0, 65536 + 4096, sink, "arrays 68 KiB further apart");
    run_jobs(base, nc, 13, 0, 524288 + 36864, sink, "arrays 548 KiB further apart");
    run_jobs(base, nc, 1, 0, 0, sink, "one product, all workgroups");
    run_jobs(base, nc, 4, 0, 0, sink, "4 products");
    run_jobs(base, nc, 4, 1, 0, sink, "4 products, rotated");
    return 0;
}
