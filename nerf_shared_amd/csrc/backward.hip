// backward.hip -- training support for the fused view-branch (10,4) field: workspace layout,
// training forward (activations saved), and the parameter gradients
//   dL/draw -> mlp_bwd_s16_kernel (pre-activation gradients of every layer, in registers)
//           -> weight gradients  dW_l = g_pre(l)^T h_(l-1)   and bias gradients (column sums)
// Saved activations X and gradients G are slot-major bf16 rows [P, n] (kernels.h).  dW is a GEMM
// whose contraction runs over the points (K = hundreds of thousands) with M, N <= 256: HBM-bound
// streaming of G and X.  dw_kernel splits the points over the workgroups; each stages 32-point
// chunks of G and X into LDS, reads both MFMA operands with the transposing LDS read
// (ds_read_b64_tr_b16: 8 consecutive points of one feature per lane), accumulates a full
// [n_out x n_in] fp32 tile in registers and adds it to the nn.Linear-layout gradient with
// float atomics (un-permuting the slot order on the way).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "launch_util.h"
#include "pipeline.h"
#include "program.h"
#include "split.h"

#include <utility>

namespace na {

template <class F, int... Is>
__device__ __forceinline__ void static_for_dw_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for_dw(F &&f) { static_for_dw_impl(f, std::make_integer_sequence<int, N>{}); }

enum { PERM_NAT = 0, PERM_ACC = 1, PERM_GEN = 2 };

__device__ __forceinline__ int slot_to_feature(int kind, int s, int L) {
    if (kind == PERM_NAT) return s;
    const int ks = s >> 5, q = (s >> 3) & 3, j = s & 7;
    return kind == PERM_ACC ? acc16_col(ks, q, j) : gen16_col(ks, q, j, L);
}

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct DwArgs {
    const uint16_t *G; int ldg;          // [P, ldg] bf16 gradient rows, columns [0, 16*OT) used
    const uint16_t *X; int ldx;          // [P, ldx] bf16 activation rows, columns [0, 16*IT) used
    int64_t P;
    float *slab;                         // per-workgroup partial results: [grid][OT*IT*256 + OT*16 (+ IT*256 + 16 with a head)] fp32
    const uint16_t *H;                   // head gradients transposed inside 32-point chunks (kernels.h g_rawt), or NULL
    // split-precision products (dw2s_body): the planes of fp16 lo rows of G, X and H (the pointers above are the hi planes)
    const uint16_t *G_lo, *X_lo, *H_lo;
};

struct DwReduceArgs {
    const float *slab; int n_slabs, OT, IT;
    float *dW; int ld_dw, col_off;       // nn.Linear weight gradient [n_out][ld_dw], written at column col_off + feature
    float *db;                           // bias gradient [n_out] or NULL
    int out_kind, in_kind, in_L, n_valid, m_valid;
    // a head product riding on this job (same X): one more 16-row tile whose rows are the columns of dL/draw; rows
    // [head_row0, head_row0 + head_rows) of it are the head's weight gradient [head_rows][head_ld] (+ bias gradient)
    int HT;                              // 0 or 1
    float *head_dW, *head_db;
    int head_row0, head_rows, head_ld;
    const float *inv_scale;              // split precision: 1 / loss scale (a device scalar, split.h), or NULL
};
__host__ __device__ inline int dw_slab_floats(int OT, int IT, int HT) { return OT * IT * 256 + OT * 16 + HT * (IT * 256 + 16); }

// Element e of a job's summed slabs -> its place in the nn.Linear gradients.
__device__ __forceinline__ void dw_scatter(const DwReduceArgs &a, int e, float acc) {
    if (a.inv_scale) acc *= *a.inv_scale;        // a power of two: exact
    const int n_main = a.OT * a.IT * 256;
    if (e < n_main) {
        const int r = e & 3, lane = (e >> 2) & 63, tile = e >> 8;
        const int to = tile / a.IT, ti = tile - to * a.IT;
        const int o = slot_to_feature(a.out_kind, to * 16 + 4 * (lane >> 4) + r, 0);
        const int i = slot_to_feature(a.in_kind, ti * 16 + (lane & 15), a.in_L);
        if (i >= 0 && o >= 0 && i < a.m_valid && o < a.n_valid) a.dW[(int64_t)o * a.ld_dw + a.col_off + i] = acc;
        return;
    }
    e -= n_main;
    if (e < a.OT * 16) {
        if (a.db) {
            const int o = slot_to_feature(a.out_kind, e, 0);
            if (o >= 0 && o < a.n_valid) a.db[o] = acc;
        }
        return;
    }
    e -= a.OT * 16;                      // head part: IT tiles (rows = columns of dL/draw, natural order), then 16 bias sums
    if (e < a.IT * 256) {
        const int r = e & 3, lane = (e >> 2) & 63, ti = e >> 8;
        const int row = 4 * (lane >> 4) + r - a.head_row0;
        const int i = slot_to_feature(a.in_kind, ti * 16 + (lane & 15), a.in_L);
        if (row >= 0 && row < a.head_rows && i >= 0 && i < a.m_valid) a.head_dW[(int64_t)row * a.head_ld + i] = acc;
    } else {
        const int row = e - a.IT * 256 - a.head_row0;
        if (row >= 0 && row < a.head_rows) a.head_db[row] = acc;
    }
}

__device__ __forceinline__ bf16x8 tr_frag(const char *img, int row_stride, int col0, int lane) {
    // 8 consecutive image rows (points 8g..8g+7) of column col0 + (lane & 15), as an MFMA 16x16x32 operand.
    const int i = lane & 15, g = lane >> 4;
    const char *p = img + (8 * g + (i >> 2)) * row_stride + (col0 + 4 * (i & 3)) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + 4 * row_stride));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// OT / IT: 16-wide tiles of the output / input feature axis; the 8 waves form a WO x WI grid.
template <int OT, int IT, int WO, int WI>
__device__ __forceinline__ void dw_body(const DwArgs &a, const int wg, const int nwg) {
    static_assert(WO * WI == 8 && OT % WO == 0 && IT % WI == 0, "bad wave grid");
    constexpr int TO = OT / WO, TI = IT / WI;
    constexpr int RSG = OT * 32 + 32, RSX = IT * 32 + 32;          // padded LDS row strides (bytes)
    constexpr int PG = OT * 2, PX = IT * 2;                         // 16-byte pieces per row
    constexpr int NPG = (32 * PG + 511) / 512, NPX = (32 * PX + 511) / 512;
    constexpr int BUF = 32 * (RSG + RSX);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = wave / WI, wi = wave % WI;

    f32x4 acc[TO][TI];
#pragma unroll
    for (int x = 0; x < TO; ++x)
#pragma unroll
        for (int y = 0; y < TI; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum[NPG][8];
#pragma unroll
    for (int k = 0; k < NPG; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) csum[k][j] = 0.f;

    const int64_t n_chunks = (a.P + 31) / 32;
    u32x4 rg[NPG], rx[NPX];
    auto load_chunk = [&](int64_t c) {
#pragma unroll
        for (int k = 0; k < NPG; ++k) {
            const int pc = tid + 512 * k, row = pc / PG, col = pc % PG;
            const int64_t p = c * 32 + row;
            rg[k] = (u32x4){0u, 0u, 0u, 0u};
            if (pc < 32 * PG && p < a.P) rg[k] = *reinterpret_cast<const u32x4 *>(a.G + p * a.ldg + col * 8);
        }
#pragma unroll
        for (int k = 0; k < NPX; ++k) {
            const int pc = tid + 512 * k, row = pc / PX, col = pc % PX;
            const int64_t p = c * 32 + row;
            rx[k] = (u32x4){0u, 0u, 0u, 0u};
            if (pc < 32 * PX && p < a.P) rx[k] = *reinterpret_cast<const u32x4 *>(a.X + p * a.ldx + col * 8);
        }
    };
    int buf = 0;
    int64_t c = wg;
    if (c < n_chunks) load_chunk(c);
    for (; c < n_chunks; c += nwg) {
        char *gimg = smem + buf * BUF, *ximg = gimg + 32 * RSG;
#pragma unroll
        for (int k = 0; k < NPG; ++k) {
            const int pc = tid + 512 * k, row = pc / PG, col = pc % PG;
            if (pc < 32 * PG) {
                *reinterpret_cast<u32x4 *>(gimg + row * RSG + col * 16) = rg[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) {       // bias gradient: this thread always holds the same 8 columns
                    csum[k][2 * j] += __builtin_bit_cast(float, rg[k][j] << 16);
                    csum[k][2 * j + 1] += __builtin_bit_cast(float, rg[k][j] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NPX; ++k) {
            const int pc = tid + 512 * k, row = pc / PX, col = pc % PX;
            if (pc < 32 * PX) *reinterpret_cast<u32x4 *>(ximg + row * RSX + col * 16) = rx[k];
        }
        __syncthreads();
        if (c + nwg < n_chunks) load_chunk(c + nwg);     // next chunk's loads fly under the MFMAs
        bf16x8 A[TO], B[TI];
#pragma unroll
        for (int x = 0; x < TO; ++x) A[x] = tr_frag(gimg, RSG, (wo * TO + x) * 16, lane);
#pragma unroll
        for (int y = 0; y < TI; ++y) B[y] = tr_frag(ximg, RSX, (wi * TI + y) * 16, lane);
#pragma unroll
        for (int x = 0; x < TO; ++x)
#pragma unroll
            for (int y = 0; y < TI; ++y)
                acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[x], B[y], acc[x][y], 0, 0, 0);
        buf ^= 1;
    }
    // ---- this workgroup's partial tile, as a register dump (1 KiB per 16x16 tile, fully coalesced);
    //      dw_reduce_kernel sums the dumps of all workgroups and un-permutes the slot order
    float *slab = a.slab + (int64_t)wg * (OT * IT * 256 + OT * 16);
#pragma unroll
    for (int x = 0; x < TO; ++x)
#pragma unroll
        for (int y = 0; y < TI; ++y)
            *reinterpret_cast<f32x4 *>(slab + (((wo * TO + x) * IT + (wi * TI + y)) * 64 + lane) * 4) = acc[x][y];
    // bias partials: threads that staged the same columns (different rows) reduce through LDS
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);
    for (int i = tid; i < OT * 16; i += 512) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NPG; ++k) {
        const int pc = tid + 512 * k, col = pc % PG;
        if (pc < 32 * PG)
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(red + col * 8 + j, csum[k][j]);
    }
    __syncthreads();
    for (int i = tid; i < OT * 16; i += 512) slab[OT * IT * 256 + i] = red[i];
}

template <int OT, int IT, int WO, int WI>
__global__ __launch_bounds__(512) void dw_kernel(DwArgs a) {
    dw_body<OT, IT, WO, WI>(a, blockIdx.x, gridDim.x);
}

// dW[feature(o_slot)][col_off + feature(i_slot)] = sum over workgroups of their register dumps.
// 64 elements per block; the slabs are split four ways over the block's waves (more loads in flight).
__global__ __launch_bounds__(256) void dw_reduce_kernel(DwReduceArgs a) {
    __shared__ float part[4][64];
    const int per = dw_slab_floats(a.OT, a.IT, a.HT);
    const int t = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + t;
    float acc = 0.f;
    if (e < per) {
#pragma unroll 8
        for (int b = grp; b < a.n_slabs; b += 4) acc += a.slab[(int64_t)b * per + e];
    }
    part[grp][t] = acc;
    __syncthreads();
    if (grp != 0 || e >= per) return;
    acc = part[0][t] + part[1][t] + part[2][t] + part[3][t];
    dw_scatter(a, e, acc);
}


// ---------------------------------------------------------------------------------------------------------
// dw2_kernel: the 256 x 256 weight-gradient product (16 of the 20 products of a training step) as a streaming kernel
// that keeps HBM busy.  The first version above stages every 32-point chunk through registers, has one chunk of loads
// in flight per CU (32 KB: less than the latency-bandwidth product of the memory system) and none while it writes LDS
// and waits at its barrier: 3.5 TB/s.  Here
//   * G and X rows go straight from HBM to LDS with LDS-DMA (global_load_lds_dwordx4, no VGPR staging) into a ring of
//     four 32-point chunks, three chunks (96 KB per CU) in flight, one counted-vmcnt wait + one barrier per chunk;
//   * the LDS image is XOR-swizzled at 16-byte granularity through the DMA's per-lane SOURCE addresses (the DMA's
//     destination is lane-linear, its source is not): piece j of row r sits at j ^ swz(r), which makes the transposing
//     operand reads (ds_read_b64_tr_b16, 8 rows x 32 B per 32-lane group) conflict-free on unpadded 512-byte rows;
//   * the bias gradient (column sums of G) is one more MFMA column against an all-ones operand.
// (Reduce jobs for the previous product's slabs inside this launch do not work out: every block of a launch reserves the
// launch's 128 KB of LDS, so the ~1000 small reduce blocks would each occupy a whole CU.)
// ---------------------------------------------------------------------------------------------------------
// One 32-lane group of a transposing read covers rows {0..3, 8..11} (+4 for the second half) x 32 bytes.  Rows of >= 256
// bytes all start at bank 0, so the eight rows need eight different piece pairs: XOR with 0, 2, .., 14.  Rows of 128 bytes
// (PIECES == 8) alternate between the two halves of the banks, so the four even rows {0, 2, 8, 10} (and the four odd ones)
// need four different pairs: XOR with 0, 2, 4, 6.  Both forms satisfy swz(r + 4) == swz(r).
#ifdef NERF_AMD_X_DW_STAMPS
// tools/micro/dw_stamps.py: when each workgroup of the last dw_multi_kernel launch started and ended (100 MHz clock)
__device__ unsigned long long g_dw_stamps[4 * 512];
extern "C" int nerf_amd_x_dw_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dw_stamps), sizeof(g_dw_stamps)) == hipSuccess ? 0 : -1;
}
#endif

constexpr int DW2_NS(int OT, int IT) {
    const int n = (128 * 1024) / (32 * 32 * (OT + IT));
    return n < 4 ? 4 : (n > 12 ? 12 : n);
}

template <int PIECES>
__device__ __forceinline__ int dw_swz(int r) {
    static_assert(PIECES == 4 || PIECES == 8 || (PIECES >= 16 && (PIECES & (PIECES - 1)) == 0), "unsupported row width");
    // 64-byte rows (PIECES == 4): four rows span the banks once, so rows {0..3} never collide and rows {8..11} take the
    // other piece pair
    if (PIECES == 4) return 2 * ((r >> 3) & 1);
    if (PIECES == 8) return 2 * (((r >> 1) & 1) | (((r >> 3) & 1) << 1));
    return 2 * ((r & 3) | (((r >> 3) & 1) << 2));
}

template <int PIECES>
__device__ __forceinline__ bf16x8 tr_frag_swz(uint32_t img, int col0, int lane) {
    // 8 consecutive image rows (points 8g..8g+7) of column col0 + (lane & 15), as an MFMA 16x16x32 operand, from the
    // swizzled image: 16-byte piece pc of row r lives at piece position pc ^ dw_swz(r).
    constexpr int row_bytes = PIECES * 16;
    const int i = lane & 15, g = lane >> 4;
    const int r = 8 * g + (i >> 2);
    const int pc = (col0 >> 3) + ((i & 3) >> 1);
    const uint32_t p = img + r * row_bytes + ((pc ^ dw_swz<PIECES>(r)) << 4) + ((i & 1) << 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(uintptr_t)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(uintptr_t)(p + 4 * row_bytes));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// One 1-KiB LDS-DMA piece: lane l fetches 16 bytes from g (per lane) into lds_base + 16 l.
__device__ __forceinline__ void dma_piece(const char *g, uint32_t lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t" "s_mov_b32 m0, %2\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %1, off\n\t" "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_base) : "memory");
}

__device__ __forceinline__ void dw_reduce_block(const DwReduceArgs &a, int block, int tid, float *part /* [8][64] LDS */) {
    // 64 elements of the summed register dumps per job; the slabs are split eight ways over the block's waves
    const int per = dw_slab_floats(a.OT, a.IT, a.HT);
    const int t = tid & 63, grp = tid >> 6;
    const int e = block * 64 + t;
    float acc = 0.f;
    if (e < per) {
#pragma unroll 8
        for (int b = grp; b < a.n_slabs; b += 8) acc += a.slab[(int64_t)b * per + e];
    }
    part[grp * 64 + t] = acc;
    __syncthreads();
    if (grp != 0 || e >= per) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) acc += part[k * 64 + t];
    dw_scatter(a, e, acc);
}

__global__ __launch_bounds__(512) void dw_reduce8_kernel(DwReduceArgs a) {
    __shared__ float part[8 * 64];
    dw_reduce_block(a, blockIdx.x, threadIdx.x, part);
}

// HEAD: a head product (alpha_linear / rgb_linear: G = columns of dL/draw) rides on this job -- same X, one more 16-row
// A operand per chunk.  Its operand needs no transposing read: the dX-chain kernel left dL/draw transposed inside 32-point
// chunks (kernels.h g_rawt), 16 bytes per (point group, column) = one lane's MFMA A operand, so wave 0 moves the chunk's
// 256 bytes into the ring slot with one more LDS-DMA instruction (lanes whose row is not a column of dL/draw fetch a valid
// 16 bytes too and are zeroed after the LDS read) and every wave reads its operand back with one ds_read_b128.  The 16
// head x X tiles are dealt two to a wave.
template <int OT, int IT, int WO, int WI, bool HEAD = false>
__device__ __forceinline__ void dw2_body(const DwArgs &a, const int wg, const int nwg) {
    static_assert(WO * WI == 8 && OT % WO == 0 && IT % WI == 0 && OT >= 4 && IT >= 2, "bad shape");
    constexpr int TO = OT / WO, TI = IT / WI;
    static_assert(!HEAD || TI == 2 * WO, "the head's X tiles are dealt two to a wave");
    constexpr int RG = OT * 32, RX = IT * 32;                   // row bytes (unpadded)
    constexpr int PG = OT * 2, PX = IT * 2;                     // 16-byte pieces per row
    constexpr int IMG_GX = 32 * (RG + RX);                      // G and X rows of one chunk
    constexpr int IMG = IMG_GX + (HEAD ? 1024 : 0);             // one chunk's image (+ the head operand)
    constexpr int NI = OT + IT, CNT = (NI + 7) / 8;             // 1-KiB DMA instructions per chunk; per wave at most
    constexpr int REM = NI - 8 * (CNT - 1);                     // waves below REM issue CNT per chunk, the others CNT - 1
    // ring slots: what fits in 128 KiB, at least 4 and at most 12 -- a narrow product keeps as many BYTES in flight as a
    // wide one (a workgroup's rate is bytes in flight / latency, and the one-launch path shares the CUs by bytes)
    constexpr int NS = DW2_NS(OT, IT);
    static_assert((NS - 2) * (CNT + (HEAD ? 1 : 0)) <= 63, "vmcnt field is 6 bits");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / WI, wi = wave % WI;
    const uint32_t ring = (uint32_t)(uintptr_t)smem;

    const int64_t n_chunks = (a.P + 31) / 32;
    const int64_t first = wg, stride = nwg;
    const int64_t n_local = wg < n_chunks ? (n_chunks - wg + nwg - 1) / nwg : 0;     // chunks wg, wg + nwg, ...
    // this wave's DMA instructions of a chunk: j = wave, wave + 8, ... below NI -- CNT of them, or CNT - 1 for the waves
    // from REM on (a count per wave, uniform inside it, so the counted waits stay compile-time constants: wait_chunk below)
    const bool full = REM == 8 || wave < REM;
    auto issue_chunk = [&](int64_t i) {
        int64_t ch = first + i * stride;
        if (ch >= n_chunks) ch = n_chunks - 1;                  // past the end: a harmless re-read that is never consumed
        const uint32_t slot = ring + (uint32_t)(i % NS) * IMG;
        const int64_t p0 = ch * 32;
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            const int j = wave + 8 * k;
            if (k == CNT - 1 && !full) break;
            const bool is_g = j < OT;
            const int jj = is_g ? j : j - OT;
            const int e = 64 * jj + lane;                       // 16-byte piece of the G (or X) image
            const int pr = is_g ? PG : PX;
            const int r = e / pr, pos = e % pr;
            int64_t p = p0 + r;
            if (p >= pad_points(a.P)) p = pad_points(a.P) - 1;  // rows exist up to the padded point count (kernels.h)
            const char *src = is_g ? reinterpret_cast<const char *>(a.G) + (p * a.ldg) * 2 : reinterpret_cast<const char *>(a.X) + (p * a.ldx) * 2;
            src += (pos ^ (is_g ? dw_swz<PG>(r) : dw_swz<PX>(r))) << 4;
            dma_piece(src, slot + (is_g ? 0 : 32 * RG) + 1024 * jj);
        }
        if constexpr (HEAD) {
            if (wave == 0)                                      // the chunk's head operand: lane (i, g) <- 16 bytes of (group g, column i & 3)
                dma_piece(reinterpret_cast<const char *>(a.H) + ch * 256 + ((lane >> 4) * 4 + (lane & 3)) * 16, slot + IMG_GX);
        }
    };

    f32x4 acc[TO][TI], accb[TO];
    f32x4 acch[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, acchb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < TO; ++x) {
        accb[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int y = 0; y < TI; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    if (n_local > 0) {
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) issue_chunk(i);
        for (int64_t i = 0; i < n_local; ++i) {
            // chunk i has landed (this wave's pieces; chunks i+1, i+2 stay in flight), everyone agrees, then the slot of
            // chunk i-1 -- which every wave finished reading before it came here -- is refilled with chunk i+3
            if (HEAD && wave == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * (CNT + 1)) : "memory");   // wave 0 issues one more piece per chunk
            else if (full) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * CNT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * (CNT - 1)) : "memory");
#ifdef NERF_AMD_X_DW_STAMPS
            if (tid == 0 && (i == 0 || i == n_local / 2)) {         // first chunk landed | half way: packed into one word
                const unsigned long long t = wall_clock64() - g_dw_stamps[4 * blockIdx.x];
                if (i == 0) g_dw_stamps[4 * blockIdx.x + 3] = t; else g_dw_stamps[4 * blockIdx.x + 3] |= t << 32;
            }
#endif
            issue_chunk(i + NS - 1);
            const uint32_t gimg = ring + (uint32_t)(i % NS) * IMG, ximg = gimg + 32 * RG;
            const int64_t ch = first + i * stride;
            if (ch == n_chunks - 1 && (a.P & 31)) {             // the last chunk: rows past P hold the padding points' data
                const int first = (int)(a.P & 31);
                for (int e = tid; e < (32 - first) * (RG + RX) / 16; e += 512) {
                    const int gp = (32 - first) * PG;
                    const uint32_t off = e < gp ? gimg + first * RG + e * 16 : ximg + first * RX + (e - gp) * 16;
                    *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)off = (u32x4){0u, 0u, 0u, 0u};
                }
                __syncthreads();
            }
            bf16x8 A[TO], B[TI];
#pragma unroll
            for (int x = 0; x < TO; ++x) A[x] = tr_frag_swz<PG>(gimg, (wo * TO + x) * 16, lane);
#pragma unroll
            for (int y = 0; y < TI; ++y) B[y] = tr_frag_swz<PX>(ximg, (wi * TI + y) * 16, lane);
#pragma unroll
            for (int x = 0; x < TO; ++x) {
#pragma unroll
                for (int y = 0; y < TI; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[x], B[y], acc[x][y], 0, 0, 0);
                if (wi == 0) accb[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[x], ones, accb[x], 0, 0, 0);
            }
            if constexpr (HEAD) {
                typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
                bf16x8 Ah = *(lds_bf16x8 *)(uintptr_t)(gimg + IMG_GX + lane * 16);
                if ((lane & 15) >= 4) Ah = bf16x8{};            // rows 4..15 of the head tile do not exist
                static_for_dw<WO>([&](auto x_) {                // this wave's two X tiles: 2 wo, 2 wo + 1 of its TI
                    constexpr int x = x_;
                    if (wo == x) {
                        acch[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B[2 * x], acch[0], 0, 0, 0);
                        acch[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B[2 * x + 1], acch[1], 0, 0, 0);
                    }
                });
                if (wave == 0) acchb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, ones, acchb, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing re-reads: no DMA may outlive the workgroup
    }
    // ---- this workgroup's partial tile as a register dump (1 KiB per 16x16 tile, fully coalesced) + the bias partials
    float *slab = a.slab + (int64_t)wg * dw_slab_floats(OT, IT, HEAD ? 1 : 0);
#pragma unroll
    for (int x = 0; x < TO; ++x)
#pragma unroll
        for (int y = 0; y < TI; ++y)
            *reinterpret_cast<f32x4 *>(slab + (((wo * TO + x) * IT + (wi * TI + y)) * 64 + lane) * 4) = acc[x][y];
    if (wi == 0 && (lane & 15) == 0) {                          // every column of accb holds the same sums: take column 0
#pragma unroll
        for (int x = 0; x < TO; ++x)
            *reinterpret_cast<f32x4 *>(slab + OT * IT * 256 + (wo * TO + x) * 16 + 4 * (lane >> 4)) = accb[x];
    }
    if constexpr (HEAD) {                                       // head tiles behind the main part: [IT][64 lanes][4], then 16 bias sums
        float *hs = slab + OT * IT * 256 + OT * 16;
#pragma unroll
        for (int t = 0; t < 2; ++t)
            *reinterpret_cast<f32x4 *>(hs + ((wi * TI + 2 * wo + t) * 64 + lane) * 4) = acch[t];
        if (wave == 0 && (lane & 15) == 0) *reinterpret_cast<f32x4 *>(hs + IT * 256 + 4 * (lane >> 4)) = acchb;
    }
}

// A head product alone (rgb_linear: nothing else multiplies the view layer's output): X rows only in the ring, the head
// operand as in dw2_body, one X tile per wave.
template <int IT>
__device__ __forceinline__ void dw_head_body(const DwArgs &a, const int wg, const int nwg) {
    static_assert(IT == 8, "one 16-column X tile per wave");
    constexpr int RX = IT * 32, PX = IT * 2, IMG_X = 32 * RX, IMG = IMG_X + 1024;
    constexpr int NS = 12;                                      // 12 x 9 KiB: as many bytes in flight as a wide product keeps
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ring = (uint32_t)(uintptr_t)smem;
    const int64_t n_chunks = (a.P + 31) / 32;
    const int64_t n_local = wg < n_chunks ? (n_chunks - wg + nwg - 1) / nwg : 0;
    auto issue_chunk = [&](int64_t i) {
        int64_t ch = wg + i * nwg;
        if (ch >= n_chunks) ch = n_chunks - 1;                  // past the end: a harmless re-read that is never consumed
        const uint32_t slot = ring + (uint32_t)(i % NS) * IMG;
        const int e = 64 * wave + lane;                         // 16-byte piece of the X image: this wave's 1 KiB
        const int r = e / PX, pos = e % PX;
        int64_t p = ch * 32 + r;
        if (p >= pad_points(a.P)) p = pad_points(a.P) - 1;
        dma_piece(reinterpret_cast<const char *>(a.X) + (p * a.ldx) * 2 + ((pos ^ dw_swz<PX>(r)) << 4), slot + 1024 * wave);
        if (wave == 0)
            dma_piece(reinterpret_cast<const char *>(a.H) + ch * 256 + ((lane >> 4) * 4 + (lane & 3)) * 16, slot + IMG_X);
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    if (n_local > 0) {
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) issue_chunk(i);
        for (int64_t i = 0; i < n_local; ++i) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * 2) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NS - 2) : "memory");
            issue_chunk(i + NS - 1);
            const uint32_t ximg = ring + (uint32_t)(i % NS) * IMG;
            typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
            bf16x8 Ah = *(lds_bf16x8 *)(uintptr_t)(ximg + IMG_X + lane * 16);
            if ((lane & 15) >= 4) Ah = bf16x8{};
            // rows past P of the last chunk hold the padding points' saved activations: finite, and their dL/draw is zero
            const bf16x8 B = tr_frag_swz<PX>(ximg, wave * 16, lane);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, B, acc, 0, 0, 0);
            if (wave == 0) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, ones, accb, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float *slab = a.slab + (int64_t)wg * dw_slab_floats(0, IT, 1);
    *reinterpret_cast<f32x4 *>(slab + (wave * 64 + lane) * 4) = acc;
    if (wave == 0 && (lane & 15) == 0) *reinterpret_cast<f32x4 *>(slab + IT * 256 + 4 * (lane >> 4)) = accb;
}


// ---------------------------------------------------------------------------------------------------------
// Split-precision weight gradients (NERF_AMD_PREC_FP32_SPLIT, split.h): G and X arrive as planes of fp16 hi rows and fp16
// lo rows (lo = residual x 2^11), exactly as the training forward and the dX chain hold them.  dW = G^T X keeps ONE
// accumulator per tile:
//     acc += G_hi X_hi  +  (2^-11 G_hi) X_lo  +  G_lo (2^-11 X_hi)
// where the two rescaled hi operands are made in registers (one packed fp16 multiply per register; where that product
// falls below fp16's normal range it loses bits of a term that is itself 2^-11 of the sum -- measured on the CPU before
// this was written, tools/experiments/split_bwd_sim.py "dW sym": 1e-6 of fp32 autograd at the loss scale split.h picks,
// 1.6e-4 even when it is 2^14 off).  Bias gradients: G_hi . 1 + G_lo . 2^-11.  The chunk image is twice the bf16 one
// (64 KiB for a 256 x 256 product), so the ring holds two chunks: one in flight while one is consumed.
// ---------------------------------------------------------------------------------------------------------
constexpr int DW2S_NS(int OT, int IT) {
    const int n = (128 * 1024) / (2 * 32 * 32 * (OT + IT));
    return n < 2 ? 2 : (n > 12 ? 12 : n);
}
#define MFMA_F16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0)

template <int OT, int IT, int WO, int WI, bool HEAD = false>
__device__ __forceinline__ void dw2s_body(const DwArgs &a, const int wg, const int nwg) {
    static_assert(WO * WI == 8 && OT % WO == 0 && IT % WI == 0 && OT >= 4 && IT >= 2, "bad shape");
    constexpr int TO = OT / WO, TI = IT / WI;
    static_assert(!HEAD || TI == 2 * WO, "the head's X tiles are dealt two to a wave");
    constexpr int RG = OT * 32, RX = IT * 32;                   // row bytes of one plane (unpadded)
    constexpr int PG = OT * 2, PX = IT * 2;                     // 16-byte pieces per row
    constexpr int IMG_G = 32 * RG, IMG_X = 32 * RX;             // one plane of one chunk
    constexpr int IMG_GX = 2 * (IMG_G + IMG_X);                 // G_hi, G_lo, X_hi, X_lo
    constexpr int IMG = IMG_GX + (HEAD ? 2048 : 0);             // + the head operand's two planes
    constexpr int NI = 2 * (OT + IT), CNT = (NI + 7) / 8;       // 1-KiB DMA instructions per chunk; per wave at most
    constexpr int REM = NI - 8 * (CNT - 1);                     // waves below REM issue CNT per chunk, the others CNT - 1
    constexpr int NS = DW2S_NS(OT, IT);
    static_assert((NS - 2) * (CNT + (HEAD ? 2 : 0)) <= 63, "vmcnt field is 6 bits");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / WI, wi = wave % WI;
    const uint32_t ring = (uint32_t)(uintptr_t)smem;

    const int64_t n_chunks = (a.P + 31) / 32;
    const int64_t n_local = wg < n_chunks ? (n_chunks - wg + nwg - 1) / nwg : 0;     // chunks wg, wg + nwg, ...
    // The ring DMA as buffer_load ... lds: a buffer resource per plane (scalar), ONE per-lane 32-bit offset per piece that
    // does not depend on the chunk (row inside the chunk x row bytes + swizzled position), the chunk as the scalar offset --
    // no 64-bit per-lane address arithmetic inside the loop, and rows past the arrays' end read as zero instead of needing a
    // clamp.  This wave's pieces of a chunk: j = wave, wave + 8, ... below NI of the pieces G_hi | G_lo | X_hi | X_lo: CNT of
    // them, or CNT - 1 for the waves from REM on (uniform per wave, so the counted waits stay compile-time constants).
    const bool full = REM == 8 || wave < REM;
    typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;
    const unsigned g_bytes = (unsigned)(pad_points(a.P) * a.ldg * 2), x_bytes = (unsigned)(pad_points(a.P) * a.ldx * 2);
    const rsrc_t rs_plane[4] = {make_rsrc(a.G, g_bytes), make_rsrc(a.G_lo, g_bytes), make_rsrc(a.X, x_bytes), make_rsrc(a.X_lo, x_bytes)};
    unsigned voff[CNT], lds_off[CNT];
    int plane[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        int j = wave + 8 * k;
        if (j >= NI) j = NI - 1;
        const bool is_g = j < 2 * OT;
        const int jp = is_g ? j : j - 2 * OT;                   // piece inside the two planes of G (or of X)
        const int n_pl = is_g ? OT : IT;                        // pieces per plane
        const bool lo = jp >= n_pl;
        const int jj = lo ? jp - n_pl : jp;
        const int e = 64 * jj + lane;                           // 16-byte piece of the plane's image
        const int rg = e / PG, posg = e % PG, rx = e / PX, posx = e % PX;
        const unsigned vg = (unsigned)(rg * a.ldg * 2 + ((posg ^ dw_swz<PG>(rg)) << 4));
        const unsigned vx = (unsigned)(rx * a.ldx * 2 + ((posx ^ dw_swz<PX>(rx)) << 4));
        voff[k] = is_g ? vg : vx;
        lds_off[k] = __builtin_amdgcn_readfirstlane((is_g ? 0 : 2 * IMG_G) + (lo ? (is_g ? IMG_G : IMG_X) : 0) + 1024 * jj);
        plane[k] = __builtin_amdgcn_readfirstlane((is_g ? 0 : 2) + (lo ? 1 : 0));
    }
    rsrc_t rs_head[2];
    unsigned voff_head = 0;
    if constexpr (HEAD) {
        rs_head[0] = make_rsrc(a.H, (unsigned)(n_chunks * 256));
        rs_head[1] = make_rsrc(a.H_lo, (unsigned)(n_chunks * 256));
        voff_head = ((lane >> 4) * 4 + (lane & 3)) * 16;        // lane (i, g) <- 16 bytes of (group g, column i & 3)
    }
    auto dma_buf = [&](unsigned vo, const rsrc_t &rs, unsigned so, uint32_t lds_base) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t" "s_mov_b32 m0, %3\n\t" "s_nop 0\n\t" "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t" "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(rs), "s"(lds_base), "s"(so) : "memory");
    };
    auto issue_chunk = [&](int64_t i) {
        int64_t ch = wg + i * nwg;
        if (ch >= n_chunks) ch = n_chunks - 1;                  // past the end: a harmless re-read that is never consumed
        const uint32_t slot = ring + (uint32_t)(i % NS) * IMG;
        const unsigned so_g = __builtin_amdgcn_readfirstlane((unsigned)(ch * 32 * a.ldg * 2));
        const unsigned so_x = __builtin_amdgcn_readfirstlane((unsigned)(ch * 32 * a.ldx * 2));
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            if (k == CNT - 1 && !full) break;
            const int pl = plane[k];
            const rsrc_t rs = pl == 0 ? rs_plane[0] : pl == 1 ? rs_plane[1] : pl == 2 ? rs_plane[2] : rs_plane[3];
            dma_buf(voff[k], rs, pl < 2 ? so_g : so_x, __builtin_amdgcn_readfirstlane(slot + lds_off[k]));
        }
        if constexpr (HEAD) {
            if (wave == 0) {                                    // the chunk's head operand, both planes
                const unsigned so_h = __builtin_amdgcn_readfirstlane((unsigned)(ch * 256));
                dma_buf(voff_head, rs_head[0], so_h, __builtin_amdgcn_readfirstlane(slot + IMG_GX));
                dma_buf(voff_head, rs_head[1], so_h, __builtin_amdgcn_readfirstlane(slot + IMG_GX + 1024));
            }
        }
    };

    f32x4 acc[TO][TI], accb[TO];
    f32x4 acch[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, acchb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < TO; ++x) {
        accb[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int y = 0; y < TI; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f16x8 ones, ones_s;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ones[j] = (_Float16)1.0f; ones_s[j] = (_Float16)SPLIT_INV; }
    const _Float16 kinv = (_Float16)SPLIT_INV;
    using PGc = std::integral_constant<int, PG>;
    using PXc = std::integral_constant<int, PX>;

    if (n_local > 0) {
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) issue_chunk(i);
        for (int64_t i = 0; i < n_local; ++i) {
            // chunk i has landed (this wave's pieces; younger chunks stay in flight), everyone agrees, then the slot of
            // chunk i-1 -- which every wave finished reading before it came here -- is refilled with chunk i+NS-1
            if (HEAD && wave == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * (CNT + 2)) : "memory");
            else if (full) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * CNT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * (CNT - 1)) : "memory");
            issue_chunk(i + NS - 1);
            const uint32_t g_hi = ring + (uint32_t)(i % NS) * IMG, g_lo = g_hi + IMG_G, x_hi = g_hi + 2 * IMG_G, x_lo = x_hi + IMG_X;
            const int64_t ch = wg + i * nwg;
            if (ch == n_chunks - 1 && (a.P & 31)) {             // the last chunk: rows past P hold the padding points' data
                const int first = (int)(a.P & 31);
                const int per_g = (32 - first) * PG, per_x = (32 - first) * PX;     // 16-byte pieces to clear per plane
                for (int e = tid; e < 2 * (per_g + per_x); e += 512) {
                    uint32_t off;
                    if (e < per_g) off = g_hi + first * RG + e * 16;
                    else if (e < 2 * per_g) off = g_lo + first * RG + (e - per_g) * 16;
                    else if (e < 2 * per_g + per_x) off = x_hi + first * RX + (e - 2 * per_g) * 16;
                    else off = x_lo + first * RX + (e - 2 * per_g - per_x) * 16;
                    *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)off = (u32x4){0u, 0u, 0u, 0u};
                }
                __syncthreads();
            }
            // (an opaque copy of the lane id per chunk: otherwise the 24 swizzled operand addresses of the body are loop
            // invariants, get hoisted and spill)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            auto frag = [&](auto pieces_, uint32_t img, int col0) {
                return __builtin_bit_cast(f16x8, tr_frag_swz<decltype(pieces_)::value>(img, col0, lane_o));
            };
            f16x8 A[TO], Ah = {}, Ahs = {}, Al_h = {};
            if constexpr (HEAD) {
                typedef __attribute__((address_space(3))) const f16x8 lds_f16x8;
                Ah = *(lds_f16x8 *)(uintptr_t)(g_hi + IMG_GX + lane * 16);
                Al_h = *(lds_f16x8 *)(uintptr_t)(g_hi + IMG_GX + 1024 + lane * 16);
                if ((lane & 15) >= 4) { Ah = f16x8{}; Al_h = f16x8{}; }     // rows 4..15 of the head tile do not exist
                Ahs = Ah * kinv;
            }
            // Only the G-side operands stay in registers; the X-side fragments are re-read from LDS for every term (three
            // transposing reads per X tile and chunk instead of two: LDS has the bandwidth, the register file has no room
            // for TI more fragments beside the TO x TI accumulators).  Straight-line code: every wave computes the bias
            // column and the head's bias column (only the waves that own them write them out).
            // ---- hi x hi
#pragma unroll
            for (int x = 0; x < TO; ++x) {
                A[x] = frag(PGc{}, g_hi, (wo * TO + x) * 16);
                accb[x] = MFMA_F16(A[x], ones, accb[x]);
            }
            static_for_dw<TI>([&](auto y_) {
                constexpr int y = y_;
                const f16x8 bh = frag(PXc{}, x_hi, (wi * TI + y) * 16);
#pragma unroll
                for (int x = 0; x < TO; ++x) acc[x][y] = MFMA_F16(A[x], bh, acc[x][y]);
            });
            // ---- (2^-11 G_hi) x X_lo
#pragma unroll
            for (int x = 0; x < TO; ++x) A[x] = A[x] * kinv;
            static_for_dw<TI>([&](auto y_) {
                constexpr int y = y_;
                const f16x8 bl = frag(PXc{}, x_lo, (wi * TI + y) * 16);
#pragma unroll
                for (int x = 0; x < TO; ++x) acc[x][y] = MFMA_F16(A[x], bl, acc[x][y]);
            });
            // ---- G_lo x (2^-11 X_hi)
#pragma unroll
            for (int x = 0; x < TO; ++x) {
                A[x] = frag(PGc{}, g_lo, (wo * TO + x) * 16);
                accb[x] = MFMA_F16(A[x], ones_s, accb[x]);
            }
            static_for_dw<TI>([&](auto y_) {
                constexpr int y = y_;
                const f16x8 bs = frag(PXc{}, x_hi, (wi * TI + y) * 16) * kinv;
#pragma unroll
                for (int x = 0; x < TO; ++x) acc[x][y] = MFMA_F16(A[x], bs, acc[x][y]);
            });
            if constexpr (HEAD) {                               // this wave's two X tiles of the head product: 2 wo, 2 wo + 1 of its TI
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int col = (wi * TI + 2 * wo + t) * 16;
                    const f16x8 bh = frag(PXc{}, x_hi, col), bl = frag(PXc{}, x_lo, col);
                    acch[t] = MFMA_F16(Ah, bh, acch[t]);
                    acch[t] = MFMA_F16(Ahs, bl, acch[t]);
                    acch[t] = MFMA_F16(Al_h, bh * kinv, acch[t]);
                }
                acchb = MFMA_F16(Ah, ones, acchb);
                acchb = MFMA_F16(Al_h, ones_s, acchb);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing re-reads: no DMA may outlive the workgroup
    }
    // ---- this workgroup's partial tile as a register dump, laid out exactly like dw2_body's
    float *slab = a.slab + (int64_t)wg * dw_slab_floats(OT, IT, HEAD ? 1 : 0);
#pragma unroll
    for (int x = 0; x < TO; ++x)
#pragma unroll
        for (int y = 0; y < TI; ++y)
            *reinterpret_cast<f32x4 *>(slab + (((wo * TO + x) * IT + (wi * TI + y)) * 64 + lane) * 4) = acc[x][y];
    if (wi == 0 && (lane & 15) == 0) {
#pragma unroll
        for (int x = 0; x < TO; ++x)
            *reinterpret_cast<f32x4 *>(slab + OT * IT * 256 + (wo * TO + x) * 16 + 4 * (lane >> 4)) = accb[x];
    }
    if constexpr (HEAD) {
        float *hs = slab + OT * IT * 256 + OT * 16;
#pragma unroll
        for (int t = 0; t < 2; ++t)
            *reinterpret_cast<f32x4 *>(hs + ((wi * TI + 2 * wo + t) * 64 + lane) * 4) = acch[t];
        if (wave == 0 && (lane & 15) == 0) *reinterpret_cast<f32x4 *>(hs + IT * 256 + 4 * (lane >> 4)) = acchb;
    }
}

// A head product alone in split precision (rgb_linear): the two planes of X rows + the head operand's two planes.
template <int IT>
__device__ __forceinline__ void dw_head_split_body(const DwArgs &a, const int wg, const int nwg) {
    static_assert(IT == 8, "one 16-column X tile per wave");
    constexpr int RX = IT * 32, PX = IT * 2, IMG_X = 32 * RX, IMG = 2 * IMG_X + 2048;
    constexpr int NS = 7;                                       // 7 x 18 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ring = (uint32_t)(uintptr_t)smem;
    const int64_t n_chunks = (a.P + 31) / 32;
    const int64_t n_local = wg < n_chunks ? (n_chunks - wg + nwg - 1) / nwg : 0;
    auto issue_chunk = [&](int64_t i) {
        int64_t ch = wg + i * nwg;
        if (ch >= n_chunks) ch = n_chunks - 1;
        const uint32_t slot = ring + (uint32_t)(i % NS) * IMG;
        const int e = 64 * wave + lane;                         // 16-byte piece of a plane's X image: this wave's 1 KiB
        const int r = e / PX, pos = e % PX;
        int64_t p = ch * 32 + r;
        if (p >= pad_points(a.P)) p = pad_points(a.P) - 1;
        const int64_t off = (p * a.ldx) * 2 + ((pos ^ dw_swz<PX>(r)) << 4);
        dma_piece(reinterpret_cast<const char *>(a.X) + off, slot + 1024 * wave);
        dma_piece(reinterpret_cast<const char *>(a.X_lo) + off, slot + IMG_X + 1024 * wave);
        if (wave == 0) {
            const int64_t ho = ch * 256 + ((lane >> 4) * 4 + (lane & 3)) * 16;
            dma_piece(reinterpret_cast<const char *>(a.H) + ho, slot + 2 * IMG_X);
            dma_piece(reinterpret_cast<const char *>(a.H_lo) + ho, slot + 2 * IMG_X + 1024);
        }
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
    f16x8 ones, ones_s;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ones[j] = (_Float16)1.0f; ones_s[j] = (_Float16)SPLIT_INV; }
    const _Float16 kinv = (_Float16)SPLIT_INV;
    if (n_local > 0) {
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) issue_chunk(i);
        for (int64_t i = 0; i < n_local; ++i) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * 4) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * 2) : "memory");
            issue_chunk(i + NS - 1);
            const uint32_t x_hi = ring + (uint32_t)(i % NS) * IMG, x_lo = x_hi + IMG_X;
            typedef __attribute__((address_space(3))) const f16x8 lds_f16x8;
            f16x8 Ah = *(lds_f16x8 *)(uintptr_t)(x_hi + 2 * IMG_X + lane * 16);
            f16x8 Al = *(lds_f16x8 *)(uintptr_t)(x_hi + 2 * IMG_X + 1024 + lane * 16);
            if ((lane & 15) >= 4) { Ah = f16x8{}; Al = f16x8{}; }
            // rows past P of the last chunk hold the padding points' saved activations: finite, and their dL/draw is zero
            const f16x8 Bh = __builtin_bit_cast(f16x8, tr_frag_swz<PX>(x_hi, wave * 16, lane));
            const f16x8 Bl = __builtin_bit_cast(f16x8, tr_frag_swz<PX>(x_lo, wave * 16, lane));
            acc = MFMA_F16(Ah, Bh, acc);
            acc = MFMA_F16(Ah * kinv, Bl, acc);
            acc = MFMA_F16(Al, Bh * kinv, acc);
            if (wave == 0) { accb = MFMA_F16(Ah, ones, accb); accb = MFMA_F16(Al, ones_s, accb); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float *slab = a.slab + (int64_t)wg * dw_slab_floats(0, IT, 1);
    *reinterpret_cast<f32x4 *>(slab + (wave * 64 + lane) * 4) = acc;
    if (wave == 0 && (lane & 15) == 0) *reinterpret_cast<f32x4 *>(slab + IT * 256 + 4 * (lane >> 4)) = accb;
}

template <int OT, int IT, int WO, int WI>
__global__ __launch_bounds__(512, 2) void dw2_kernel(DwArgs a) {
    dw2_body<OT, IT, WO, WI>(a, blockIdx.x, gridDim.x);
}

// Heads (alpha_linear, rgb_linear): G = NO <= 4 natural-order columns of g_rawb [P, 4], X = [P, n_in]
// slot-major.  A block walks 256-row tiles; a thread owns 8 consecutive X columns (one 16-byte load
// per row) of every (256 / groups)-th row.  Partial sums meet in LDS and leave as one slab row per
// block ([NO][n_in] weights, then NO biases); dw_small_reduce_kernel sums the rows.
template <int NO, int NT>
__device__ __forceinline__ void dw_small_body(const uint16_t *G, int ldg, int g_col0, const uint16_t *X, int ldx, int n_in, int64_t P,
                                              float *slab, float (*red)[256 + 1] /* LDS [NO][257] */, const int wg, const int nwg) {
    const int groups = n_in / 8;                       // column groups per row (32 for 256 columns, 16 for 128)
    const int rows_par = NT / groups;                  // rows in flight per block
    const int cg = threadIdx.x % groups, ty = threadIdx.x / groups;
    float acc[NO][8], bs[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        bs[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    }
    for (int64_t r0 = (int64_t)wg * 256; r0 < P; r0 += (int64_t)nwg * 256) {
        const int64_t r1 = r0 + 256 < P ? r0 + 256 : P;
#pragma unroll 4
        for (int64_t p = r0 + ty; p < r1; p += rows_par) {
            const u32x4 xv = *reinterpret_cast<const u32x4 *>(X + p * ldx + cg * 8);
            float x[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[2 * j] = __builtin_bit_cast(float, xv[j] << 16);
                x[2 * j + 1] = __builtin_bit_cast(float, xv[j] & 0xffff0000u);
            }
            // the row's four gradient values (the aligned group of four columns g_col0 lies in) as one 8-byte load
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const u32x2 gv = *reinterpret_cast<const u32x2 *>(G + p * ldg + (g_col0 & ~3));
#pragma unroll
            for (int k = 0; k < NO; ++k) {
                const int col = (g_col0 & 3) + k;
                const unsigned word = (col & 2) ? gv[1] : gv[0];
                const float g = __builtin_bit_cast(float, (col & 1) ? (word & 0xffff0000u) : (word << 16));
                if (cg == 0) bs[k] += g;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[k][j] += g * x[j];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NO; ++k)
        if (threadIdx.x < 256) red[k][threadIdx.x] = 0.f;
    if (threadIdx.x < NO) red[threadIdx.x][256] = 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NO; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&red[k][cg * 8 + j], acc[k][j]);
        if (cg == 0) atomicAdd(&red[k][256], bs[k]);
    }
    __syncthreads();
    float *row = slab + (int64_t)wg * (NO * n_in + NO);
    if ((int)threadIdx.x < n_in)
#pragma unroll
        for (int k = 0; k < NO; ++k) row[k * n_in + threadIdx.x] = red[k][threadIdx.x];
    if (threadIdx.x < NO) row[NO * n_in + threadIdx.x] = red[threadIdx.x][256];
}

template <int NO>
__global__ __launch_bounds__(256) void dw_small_kernel(const uint16_t *G, int ldg, int g_col0, const uint16_t *X, int ldx, int n_in,
                                                       int64_t P, float *slab) {
    __shared__ float red[NO][256 + 1];
    dw_small_body<NO, 256>(G, ldg, g_col0, X, ldx, n_in, P, slab, red, blockIdx.x, gridDim.x);
}

// dW[k][feature(slot)] / db[k] = sum of the slab rows dw_small_kernel left; 64 elements per block.
__global__ __launch_bounds__(1024) void dw_small_reduce_kernel(const float *slab, int n_rows, int NO, int n_in, int in_kind,
                                                               float *dW, int ld_dw, float *db) {
    __shared__ float part[16][64];
    const int per = NO * n_in + NO;
    const int t = threadIdx.x & 63, grp = threadIdx.x >> 6;      // 64 elements x 16 row groups
    const int e = blockIdx.x * 64 + t;
    float acc = 0.f;
    if (e < per) {
#pragma unroll 8
        for (int b = grp; b < n_rows; b += 16) acc += slab[(int64_t)b * per + e];
    }
    part[grp][t] = acc;
    __syncthreads();
    if (grp != 0 || e >= per) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) acc += part[k][t];
    if (e < NO * n_in) {
        const int k = e / n_in, i = slot_to_feature(in_kind, e - k * n_in, 0);
        if (i >= 0) dW[(int64_t)k * ld_dw + i] = acc;
    } else {
        db[e - NO * n_in] = acc;
    }
}

// G [P, ldg] bf16 rows; columns g_col0 .. g_col0 + NO - 1 must lie inside one aligned group of four.
template <int NO>
static void launch_dw_small(hipStream_t s, int64_t P, float *slab, const uint16_t *G, int g_col0, const uint16_t *X, int n_in,
                            float *dW, float *db, int ldg = 4) {
    int64_t g = (P + 255) / 256;
    const int64_t cap = g_variant == 50 ? 256 : 1024;  // four 256-thread blocks per CU keep enough 16-byte loads in flight; 1024 * (3 * 256 + 3) floats fit the slab
    if (g > cap) g = cap;
    hipLaunchKernelGGL(dw_small_kernel<NO>, dim3((unsigned)g), dim3(256), 0, s, G, ldg, g_col0, X, n_in, n_in, P, slab);
    hipLaunchKernelGGL(dw_small_reduce_kernel, dim3((NO * n_in + NO + 63) / 64), dim3(1024), 0, s, slab, (int)g, NO, n_in,
                       (int)PERM_ACC, dW, n_in, db);
}

// Every streaming product of one model's backward pass in ONE launch, and their reductions in a second one: a job owns a
// range of workgroups sized to its bytes per point, 256 workgroups in all -- one per CU from the first chunk to the last, all
// products in flight together (no ramp and tail per product, ~23 slabs per product to reduce instead of 128-256, and 2
// launches per model instead of 22 on a step that is host-bound at the reference's batch size).
constexpr int DW_MAX_JOBS = 16;
struct DwJob {
    DwArgs a;
    int shape;                           // 0: <16,16,4,2>  1: <8,16,4,2>  2: <16,4,8,1>  3: <8,2,8,1>  4: <16,16,4,2> + head  5: head alone on 8 X tiles
                                         // 6: <16,8,4,2>  7: <8,4,8,1>   (the encodings of a multires 15 / 6 model)
    int first_block, n_blocks;
};
struct DwMulti {
    DwJob job[DW_MAX_JOBS];
    int n;
};
struct DwReduceMulti {
    DwReduceArgs r[DW_MAX_JOBS];
    int first_block[DW_MAX_JOBS + 1];
    int n;
};

__global__ __launch_bounds__(512, 2) void dw_multi_kernel(DwMulti m) {
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.job[j + 1].first_block) ++j;
    j = __builtin_amdgcn_readfirstlane(j);
    const DwJob &J = m.job[j];
    const int wg = (int)blockIdx.x - J.first_block;
#ifdef NERF_AMD_X_DW_STAMPS
    if (threadIdx.x == 0) {
        g_dw_stamps[4 * blockIdx.x] = wall_clock64();
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_dw_stamps[4 * blockIdx.x + 2] = (unsigned long long)j << 8 | (unsigned)J.shape | (unsigned long long)(xcc & 15) << 32 | (unsigned long long)(hwid & 0xffff) << 40;
    }
#endif
    switch (J.shape) {
    case 0: dw2_body<16, 16, 4, 2>(J.a, wg, J.n_blocks); break;
    case 1: dw2_body<8, 16, 4, 2>(J.a, wg, J.n_blocks); break;
    case 2: dw2_body<16, 4, 8, 1>(J.a, wg, J.n_blocks); break;
    case 3: dw2_body<8, 2, 8, 1>(J.a, wg, J.n_blocks); break;
    case 4: dw2_body<16, 16, 4, 2, true>(J.a, wg, J.n_blocks); break;
    case 6: dw2_body<16, 8, 4, 2>(J.a, wg, J.n_blocks); break;
    case 7: dw2_body<8, 4, 8, 1>(J.a, wg, J.n_blocks); break;
    default: dw_head_body<8>(J.a, wg, J.n_blocks); break;
    }
#ifdef NERF_AMD_X_DW_STAMPS
    __syncthreads();
    if (threadIdx.x == 0) g_dw_stamps[4 * blockIdx.x + 1] = wall_clock64();
#endif
}

// The same one launch for the split-precision products (dw2s_body): job shapes 0..5 as above, 6: <16,8,4,2> (the xyz
// encoding of a multires-15 model, rows padded to 128 slots), 7: <8,4,8,1> (its view-direction encoding).
__global__ __launch_bounds__(512, 2) void dw_multi_split_kernel(DwMulti m) {
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.job[j + 1].first_block) ++j;
    j = __builtin_amdgcn_readfirstlane(j);
    const DwJob &J = m.job[j];
    const int wg = (int)blockIdx.x - J.first_block;
#ifdef NERF_AMD_X_DW_STAMPS
    if (threadIdx.x == 0) {
        g_dw_stamps[4 * blockIdx.x] = wall_clock64();
        g_dw_stamps[4 * blockIdx.x + 2] = (unsigned long long)j << 8 | (unsigned)J.shape;
    }
#endif
    switch (J.shape) {
    case 0: dw2s_body<16, 16, 4, 2>(J.a, wg, J.n_blocks); break;
    case 1: dw2s_body<8, 16, 4, 2>(J.a, wg, J.n_blocks); break;
    case 2: dw2s_body<16, 4, 8, 1>(J.a, wg, J.n_blocks); break;
    case 3: dw2s_body<8, 2, 8, 1>(J.a, wg, J.n_blocks); break;
    case 4: dw2s_body<16, 16, 4, 2, true>(J.a, wg, J.n_blocks); break;
    case 6: dw2s_body<16, 8, 4, 2>(J.a, wg, J.n_blocks); break;
    case 7: dw2s_body<8, 4, 8, 1>(J.a, wg, J.n_blocks); break;
    default: dw_head_split_body<8>(J.a, wg, J.n_blocks); break;
    }
#ifdef NERF_AMD_X_DW_STAMPS
    __syncthreads();
    if (threadIdx.x == 0) g_dw_stamps[4 * blockIdx.x + 1] = wall_clock64();
#endif
}
constexpr size_t dw_split_lds(int OT, int IT, bool head) { return (size_t)DW2S_NS(OT, IT) * (2 * 32 * 32 * (OT + IT) + (head ? 2048 : 0)); }

// output_linear's weight gradient in split precision (models without view branch, nerf.py:131-132): G = NO <= 4 columns
// of dL/draw itself (fp32, unscaled), X = the two planes of h8; fp32 FMA loops like dw_small_body.
template <int NO>
__global__ __launch_bounds__(256) void dw_small_split_kernel(const float *G, int ldg, int g_col0, const uint16_t *Xh, const uint16_t *Xl,
                                                             int ldx, int n_in, int64_t P, float *slab) {
    __shared__ float red[NO][256 + 1];
    const int groups = n_in / 8, rows_par = 256 / groups;
    const int cg = threadIdx.x % groups, ty = threadIdx.x / groups;
    const int wg = blockIdx.x, nwg = gridDim.x;
    float acc[NO][8], bs[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        bs[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    }
    for (int64_t r0 = (int64_t)wg * 256; r0 < P; r0 += (int64_t)nwg * 256) {
        const int64_t r1 = r0 + 256 < P ? r0 + 256 : P;
#pragma unroll 4
        for (int64_t p = r0 + ty; p < r1; p += rows_par) {
            const f16x8 xh = *reinterpret_cast<const f16x8 *>(Xh + p * ldx + cg * 8);
            const f16x8 xl = *reinterpret_cast<const f16x8 *>(Xl + p * ldx + cg * 8);
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf((float)xl[j], SPLIT_INV, (float)xh[j]);
#pragma unroll
            for (int k = 0; k < NO; ++k) {
                const float g = G[p * ldg + g_col0 + k];
                if (cg == 0) bs[k] += g;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[k][j] += g * x[j];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NO; ++k) red[k][threadIdx.x] = 0.f;
    if (threadIdx.x < NO) red[threadIdx.x][256] = 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NO; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&red[k][cg * 8 + j], acc[k][j]);
        if (cg == 0) atomicAdd(&red[k][256], bs[k]);
    }
    __syncthreads();
    float *row = slab + (int64_t)wg * (NO * n_in + NO);
    if ((int)threadIdx.x < n_in)
#pragma unroll
        for (int k = 0; k < NO; ++k) row[k * n_in + threadIdx.x] = red[k][threadIdx.x];
    if (threadIdx.x < NO) row[NO * n_in + threadIdx.x] = red[threadIdx.x][256];
}

// A job's slabs are few here (~23), so one thread sums one element over all of them (512 elements per block, no LDS, no
// barrier; 8x fewer blocks than the 64-element jobs of dw_reduce_block, whose count is what bounds that kernel).
constexpr int DWR_BLOCK = 512;
__global__ __launch_bounds__(DWR_BLOCK) void dw_reduce_multi_kernel(DwReduceMulti m) {
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.first_block[j + 1]) ++j;
    j = __builtin_amdgcn_readfirstlane(j);
    const DwReduceArgs &a = m.r[j];
    const int per = dw_slab_floats(a.OT, a.IT, a.HT);
    const int e = ((int)blockIdx.x - m.first_block[j]) * DWR_BLOCK + (int)threadIdx.x;
    if (e >= per) return;
    float acc = 0.f;
#pragma unroll 8
    for (int b = 0; b < a.n_slabs; ++b) acc += a.slab[(int64_t)b * per + e];
    dw_scatter(a, e, acc);
}

namespace {
struct TrainWs {
    uint16_t *sv_e, *sv_d, *sv_h, *sv_feat, *sv_hv, *g_rawb, *g_rawt, *g_hv, *g_feat, *g_h;
    uint16_t *sv_e_lo, *sv_d_lo, *sv_h_lo, *sv_feat_lo, *sv_hv_lo, *g_rawt_lo, *g_hv_lo, *g_feat_lo, *g_h_lo;   // split: the lo planes
    uint8_t *sv_bits;
    float *g_scale;     // split: GRAD_SCALE_PARTS partial maxima of |dL/draw|, then S and 1 / S (split.h)
    float *slab;        // 2 x DW_GRID partial [256 x 256 + 256] fp32 results of a weight-gradient product (alternating:
                        // the reduction of one product runs beside the next product)
};
constexpr size_t SLAB_FLOATS = (size_t)256 * (256 * 256 + 256);
constexpr int DW_GRID = 256;
size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

// Rows of a saved xyz-encoding row: 32 per k-step; a three-k-step encoding (multires 15) is padded to 128 so that its
// weight-gradient product keeps power-of-two rows and streams with the others in the one launch (dw2_body / dw2s_body; the
// extra slots are zero and map to no weight column).  (Round 4: the bf16 arrays too -- their 96-slot rows had left the two
// xyz-encoding products of a multires-15 model on the round-1 kernel, 174 us each beside a 284-us launch of everything else.)
int enc_row_slots(int k16, bool split) { (void)split; return k16 == 3 ? 128 : 32 * k16; }

int64_t carve(const Program &p, int64_t P_points, char *base, TrainWs *w, bool split) {
    const size_t P = (size_t)pad_points(P_points);     // rows for the last workgroup's padding points too
    size_t off = 0;
    auto take = [&](size_t bytes) { char *q = base ? base + off : nullptr; off += al(bytes); return q; };
    auto lo = [&](size_t bytes) { return split ? (uint16_t *)take(bytes) : nullptr; };
    const size_t e = enc_row_slots(p.KE16, split), d = 32 * p.KD16;
    TrainWs t;
    t.sv_e = (uint16_t *)take(P * e * 2);              t.sv_e_lo = lo(P * e * 2);
    t.sv_d = (uint16_t *)take(P * d * 2);              t.sv_d_lo = lo(P * d * 2);
    t.sv_h = (uint16_t *)take((size_t)8 * P * 256 * 2); t.sv_h_lo = lo((size_t)8 * P * 256 * 2);
    t.sv_feat = (uint16_t *)take(P * 256 * 2);         t.sv_feat_lo = lo(P * 256 * 2);
    t.sv_hv = (uint16_t *)take(P * 128 * 2);           t.sv_hv_lo = lo(P * 128 * 2);
    t.sv_bits = (uint8_t *)take(P * (8 * 32 + 16));
    t.g_rawb = (uint16_t *)take(P * 16 * 2);           // [P, 4] with a view branch, [P, 16] without
    t.g_rawt = (uint16_t *)take(P * 4 * 2);            t.g_rawt_lo = lo(P * 4 * 2);
    t.g_hv = (uint16_t *)take(P * 128 * 2);            t.g_hv_lo = lo(P * 128 * 2);
    t.g_feat = (uint16_t *)take(P * 256 * 2);          t.g_feat_lo = lo(P * 256 * 2);
    t.g_h = (uint16_t *)take((size_t)8 * P * 256 * 2); t.g_h_lo = lo((size_t)8 * P * 256 * 2);
    t.g_scale = (float *)take((GRAD_SCALE_PARTS + 2) * sizeof(float));
    t.slab = (float *)take(2 * SLAB_FLOATS * sizeof(float));
    if (w) *w = t;
    return (int64_t)off;
}
}  // namespace

bool train_supported(const Program &p) {
    const nerf_amd_arch &a = p.arch;
    if (!p.bf16_ok || a.i_embed != 0) return false;
    if (a.use_viewdirs) return (a.multires == 10 && a.multires_views == 4) || (a.multires == 15 && a.multires_views == 6);
    return (a.multires == 10 || a.multires == 15) && p.out_ch <= 16;       // output_linear models (nerf.py:91-94)
}

int64_t train_workspace_bytes(const Program &p, int64_t P, bool split) { return carve(p, P, nullptr, nullptr, split); }

void train_fill_args(const Program &p, int64_t P, void *workspace, MlpArgs *a, bool split) {
    TrainWs w;
    carve(p, P, static_cast<char *>(workspace), &w, split);
    a->sv_e = w.sv_e; a->sv_d = w.sv_d; a->sv_h = w.sv_h; a->sv_feat = w.sv_feat; a->sv_hv = w.sv_hv;
    a->sv_bits = w.sv_bits;
    a->g_rawb = w.g_rawb; a->g_rawt = w.g_rawt; a->g_hv = w.g_hv; a->g_feat = w.g_feat; a->g_h = w.g_h;
    a->sv_e_lo = w.sv_e_lo; a->sv_d_lo = w.sv_d_lo; a->sv_h_lo = w.sv_h_lo; a->sv_feat_lo = w.sv_feat_lo; a->sv_hv_lo = w.sv_hv_lo;
    a->g_rawt_lo = w.g_rawt_lo; a->g_hv_lo = w.g_hv_lo; a->g_feat_lo = w.g_feat_lo; a->g_h_lo = w.g_h_lo;
    a->g_scale = w.g_scale;
}

// The products of one training step in order.  Product k dumps its slabs into buffer k % 2 on the caller's stream; its
// reduction runs on the library's side stream beside product k + 1 (a small-register kernel that co-resides with the
// streaming workgroups), so the caller's stream only ever waits for the reduction two products back.
struct DwSeq {
    hipStream_t main_s = nullptr, side = nullptr;
    std::vector<hipEvent_t> ev;      // [2k] product k dumped, [2k+1] product k reduced
    float *slab = nullptr;
    int k = 0;
    bool overlap = false;
    int lanes = 0;                   // > 0: products go round-robin over this many streams (the caller's and lane_s[]), each with
                                     // its own share of the slab buffer; a product's reduction follows it on its own stream
                                     // (one fork and one join per model instead of a hand-off per product)
    hipStream_t lane_s[3] = {nullptr, nullptr, nullptr};
    int grid_cap = DW_GRID;
    bool multi = false;              // streaming products are collected and leave as one launch (flush); the others run
                                     // on lane 1 with the second half of the slab buffer
    DwMulti mj;
    DwReduceMulti mr;
    float *buffer() const {
        if (multi) return slab + SLAB_FLOATS;
        if (lanes > 0) return slab + (size_t)(k % lanes) * (2 * SLAB_FLOATS / lanes);
        return slab + (size_t)(k & 1) * SLAB_FLOATS;
    }
    hipStream_t lane() const {
        if (multi) return lane_s[0];
        const int l = k % lanes;
        return l ? lane_s[l - 1] : main_s;
    }
    int flush();
    hipStream_t begin() {            // before product k's kernel: its slab buffer is free again
        if (lanes) return lane();
        if (overlap && k >= 2) (void)hipStreamWaitEvent(main_s, ev[2 * (k - 2) + 1], 0);
        return main_s;
    }
    hipStream_t reduce_stream() {    // after product k's kernel was enqueued
        if (lanes) return lane();
        if (!overlap) return main_s;
        (void)hipEventRecord(ev[2 * k], main_s);
        (void)hipStreamWaitEvent(side, ev[2 * k], 0);
        return side;
    }
    void end() {                     // after product k's reduction was enqueued
        if (overlap) (void)hipEventRecord(ev[2 * k + 1], side);
        ++k;
    }
    void join() {                    // the caller's stream continues behind every reduction
        if (lanes) {
            for (int l = 1; l < lanes; ++l) {
                (void)hipEventRecord(ev[l], lane_s[l - 1]);
                (void)hipStreamWaitEvent(main_s, ev[l], 0);
            }
            return;
        }
        if (!overlap) return;
        for (int j = k - 2 < 0 ? 0 : k - 2; j < k; ++j) (void)hipStreamWaitEvent(main_s, ev[2 * j + 1], 0);
    }
};

template <int OT, int IT, int WO, int WI>
static int launch_dw2(const DwArgs &a, const DwReduceArgs &ra, DwSeq &q) {
    if (q.multi) {
        static_assert((OT == 16 && (IT == 16 || IT == 8 || IT == 4)) || (OT == 8 && (IT == 16 || IT == 4)), "shape not in dw_multi_kernel");
        if (q.mj.n >= DW_MAX_JOBS) return NERF_AMD_EINVAL;
        if (ra.HT && !(OT == 16 && IT == 16)) return NERF_AMD_EINVAL;      // a head rides on the 256 x 256 shape only
        DwJob &J = q.mj.job[q.mj.n];
        J.a = a;
        J.shape = OT == 16 ? (IT == 16 ? (ra.HT ? 4 : 0) : IT == 8 ? 6 : 2) : (IT == 16 ? 1 : 7);
        DwReduceArgs &r = q.mr.r[q.mj.n];
        r = ra; r.OT = OT; r.IT = IT;
        ++q.mj.n;
        return NERF_AMD_OK;
    }
    const size_t lds = (size_t)DW2_NS(OT, IT) * 32 * (OT * 32 + IT * 32);
    static DynamicLdsOptIn opt_in;
    if (opt_in.ensure(reinterpret_cast<const void *>(dw2_kernel<OT, IT, WO, WI>), lds) != hipSuccess) return NERF_AMD_EHIP;
    // at least ~16 chunks (512 points) per workgroup so the partial tiles are worth their dump and reduction (measured at
    // 1024 rays: 2.35 ms per step with 16, 2.37 with 10, 2.40 with 6 chunks per workgroup)
    const int64_t n_chunks = (a.P + 31) / 32;
    int64_t g = n_chunks / 16;
    if (g < 1) g = 1;
    if (g > q.grid_cap) g = q.grid_cap;
    const unsigned grid = (unsigned)g;
    DwArgs a2 = a;
    a2.slab = q.buffer();
    hipLaunchKernelGGL((dw2_kernel<OT, IT, WO, WI>), dim3(grid), dim3(512), lds, q.begin(), a2);
    DwReduceArgs r = ra;
    r.slab = a2.slab; r.n_slabs = (int)grid; r.OT = OT; r.IT = IT;
    hipLaunchKernelGGL(dw_reduce8_kernel, dim3((OT * IT * 256 + OT * 16 + 63) / 64), dim3(512), 0, q.reduce_stream(), r);
    q.end();
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

template <int OT, int IT, int WO, int WI>
static int launch_dw(const DwArgs &a, const DwReduceArgs &ra, DwSeq &q) {
    if constexpr (OT == 8 && IT == 2) {
        if (q.multi) {
            if (q.mj.n >= DW_MAX_JOBS) return NERF_AMD_EINVAL;
            DwJob &J = q.mj.job[q.mj.n];
            J.a = a; J.shape = 3;
            DwReduceArgs &r = q.mr.r[q.mj.n];
            r = ra; r.OT = OT; r.IT = IT;
            ++q.mj.n;
            return NERF_AMD_OK;
        }
    }
    constexpr int RSG = OT * 32 + 32, RSX = IT * 32 + 32;
    const size_t lds = 2 * 32 * (RSG + RSX);
    static DynamicLdsOptIn opt_in;
    if (opt_in.ensure(reinterpret_cast<const void *>(dw_kernel<OT, IT, WO, WI>), lds) != hipSuccess) return NERF_AMD_EHIP;
    // at least ~16 chunks (512 points) per workgroup so the partial tiles are worth their reduction
    const int64_t n_chunks = (a.P + 31) / 32;
    int64_t g = n_chunks / 16;
    if (g < 1) g = 1;
    if (g > q.grid_cap) g = q.grid_cap;
    const unsigned grid = (unsigned)g;
    DwArgs a2 = a;
    a2.slab = q.buffer();
    hipLaunchKernelGGL((dw_kernel<OT, IT, WO, WI>), dim3(grid), dim3(512), lds, q.begin(), a2);
    DwReduceArgs r = ra;
    r.slab = a2.slab; r.n_slabs = (int)grid; r.OT = OT; r.IT = IT;
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((OT * IT * 256 + OT * 16 + 63) / 64), dim3(256), 0, q.reduce_stream(), r);
    q.end();
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// Workgroups per product of the one-launch paths.  Every product streams the same number of chunks, but what a workgroup
// needs per chunk depends on the product's shape -- bytes, MFMAs, and a cost per ring step that does not shrink with the
// chunk -- so sharing the workgroups by bytes let the narrow products finish last: at 196608 points the <8,2> product ended
// at 495 us and the head-alone one at 440 us when the 256 x 256 products were done at 385 us, and the launch takes as long as
// its last workgroup.  The table is that measurement (tools/micro/dw_stamps.py: us per chunk and workgroup by job shape,
// every CU busy, bf16 and split-precision kernels); the next workgroup always goes to the product that would end last.
static void dw_share_workgroups(const DwMulti &mj, bool split, int cap, int *nb) {
    static const float cost_bf16[8] = {1.49f, 1.06f, 0.90f, 0.56f, 1.64f, 0.43f, 1.06f, 0.62f};
    static const float cost_split[8] = {2.88f, 1.91f, 1.43f, 0.80f, 3.11f, 0.544f, 1.91f, 0.95f};
    const float *cost = split ? cost_split : cost_bf16;
    int used = 0;
    for (int j = 0; j < mj.n; ++j) { nb[j] = 1; ++used; }
    for (; used < DW_GRID; ++used) {
        int best = -1;
        float worst = 0.f;
        for (int j = 0; j < mj.n; ++j) {
            const float t = cost[mj.job[j].shape & 7] / (float)nb[j];
            if (nb[j] < cap && t > worst) { worst = t; best = j; }
        }
        if (best < 0) break;                                    // every product has as many workgroups as 8-chunk pieces
        ++nb[best];
    }
}

int DwSeq::flush() {
    if (!multi || mj.n == 0) return NERF_AMD_OK;
    const size_t lds = 4 * (32 * (16 * 32 + 16 * 32) + 1024);     // the largest job shape: 256 x 256 with a head operand per ring slot
    static DynamicLdsOptIn opt_in;
    if (opt_in.ensure(reinterpret_cast<const void *>(dw_multi_kernel), lds) != hipSuccess) return NERF_AMD_EHIP;
    // DW_GRID workgroups in all, shared by dw_share_workgroups (never more than a product has 8-chunk pieces); every
    // product's slabs follow the previous product's
    const int64_t n_chunks = (mj.job[0].a.P + 31) / 32;
    int nb[DW_MAX_JOBS], per[DW_MAX_JOBS];
    for (int j = 0; j < mj.n; ++j) per[j] = dw_slab_floats(mr.r[j].OT, mr.r[j].IT, mr.r[j].HT);
    const int cap = n_chunks / 8 < 1 ? 1 : (int)(n_chunks / 8 > DW_GRID ? DW_GRID : n_chunks / 8);
    dw_share_workgroups(mj, false, cap, nb);
    float *sl = slab;
    int first = 0, rfirst = 0;
    for (int j = 0; j < mj.n; ++j) {
        mj.job[j].a.slab = sl;
        mj.job[j].first_block = first; mj.job[j].n_blocks = nb[j];
        mr.r[j].slab = sl; mr.r[j].n_slabs = nb[j];
        mr.first_block[j] = rfirst;
        sl += (size_t)nb[j] * per[j];
        first += nb[j];
        rfirst += (per[j] + DWR_BLOCK - 1) / DWR_BLOCK;
    }
    mr.first_block[mj.n] = rfirst;
    mr.n = mj.n;
    hipLaunchKernelGGL(dw_multi_kernel, dim3((unsigned)first), dim3(512), lds, main_s, mj);
    hipLaunchKernelGGL(dw_reduce_multi_kernel, dim3((unsigned)rfirst), dim3(512), 0, main_s, mr);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// dW[:, col_off : col_off + m_valid] (+ db) of one Linear from G [P, 16*OT] and X [P, 16*IT].
// A head product that shares X with a streaming product (or stands alone): rows [row0, row0 + rows) of dL/draw's columns.
struct HeadSpec {
    const uint16_t *H = nullptr;         // kernels.h g_rawt
    int row0 = 0, rows = 0, ld = 0;
    float *dW = nullptr, *db = nullptr;
};

static int weight_grad(DwSeq &s, int64_t P, float *slab, const uint16_t *X, int n_in_slots, int in_kind, int in_L,
                       int m_valid, const uint16_t *G, int n_out_slots, int n_valid, float *dW, int ld_dw, int col_off,
                       float *db, const HeadSpec *head = nullptr) {
    DwArgs a;
    a.G = G; a.ldg = n_out_slots; a.X = X; a.ldx = n_in_slots; a.P = P; a.slab = slab; a.H = head ? head->H : nullptr;
    DwReduceArgs r;
    r.slab = slab; r.n_slabs = 0; r.OT = 0; r.IT = 0;
    a.G_lo = a.X_lo = a.H_lo = nullptr; r.inv_scale = nullptr;
    r.HT = head ? 1 : 0; r.head_dW = head ? head->dW : nullptr; r.head_db = head ? head->db : nullptr;
    r.head_row0 = head ? head->row0 : 0; r.head_rows = head ? head->rows : 0; r.head_ld = head ? head->ld : 0;
    if (head && !(s.multi && n_out_slots == 256 && n_in_slots == 256)) return NERF_AMD_EINVAL;
    r.dW = dW; r.ld_dw = ld_dw; r.col_off = col_off; r.db = db;
    r.out_kind = PERM_ACC; r.in_kind = in_kind; r.in_L = in_L; r.n_valid = n_valid; r.m_valid = m_valid;
    if (n_out_slots == 256 && n_in_slots == 256) return g_variant == 50 ? launch_dw<16, 16, 4, 2>(a, r, s) : launch_dw2<16, 16, 4, 2>(a, r, s);
    if (n_out_slots == 128 && n_in_slots == 256 && g_variant != 50) return launch_dw2<8, 16, 4, 2>(a, r, s);
    if (n_out_slots == 256 && n_in_slots == 64) return g_variant == 50 ? launch_dw<16, 4, 8, 1>(a, r, s) : launch_dw2<16, 4, 8, 1>(a, r, s);
    if (n_out_slots == 256 && n_in_slots == 128) return g_variant == 50 ? launch_dw<16, 8, 4, 2>(a, r, s) : launch_dw2<16, 8, 4, 2>(a, r, s);
    if (n_out_slots == 128 && n_in_slots == 64) return g_variant == 50 ? launch_dw<8, 4, 8, 1>(a, r, s) : launch_dw2<8, 4, 8, 1>(a, r, s);
    if (n_out_slots == 128 && n_in_slots == 256) return launch_dw<8, 16, 4, 2>(a, r, s);
    if (n_out_slots == 128 && n_in_slots == 32) return launch_dw<8, 2, 8, 1>(a, r, s);
    return NERF_AMD_EUNSUPPORTED;
}

// A head product alone, as a job of the one launch (shape 5): X [P, 128] slot-major rows.
static int head_grad(DwSeq &s, int64_t P, const uint16_t *X, int n_in_slots, int in_kind, int m_valid, const HeadSpec &head) {
    if (!s.multi || n_in_slots != 128 || s.mj.n >= DW_MAX_JOBS) return NERF_AMD_EINVAL;
    DwJob &J = s.mj.job[s.mj.n];
    J.a.G = nullptr; J.a.ldg = 0; J.a.X = X; J.a.ldx = n_in_slots; J.a.P = P; J.a.slab = nullptr; J.a.H = head.H;
    J.a.G_lo = J.a.X_lo = J.a.H_lo = nullptr;
    J.shape = 5;
    DwReduceArgs &r = s.mr.r[s.mj.n];
    r.slab = nullptr; r.n_slabs = 0; r.OT = 0; r.IT = n_in_slots / 16;
    r.dW = nullptr; r.ld_dw = 0; r.col_off = 0; r.db = nullptr;
    r.out_kind = PERM_NAT; r.in_kind = in_kind; r.in_L = 0; r.n_valid = 0; r.m_valid = m_valid; r.inv_scale = nullptr;
    r.HT = 1; r.head_dW = head.dW; r.head_db = head.db; r.head_row0 = head.row0; r.head_rows = head.rows; r.head_ld = head.ld;
    ++s.mj.n;
    return NERF_AMD_OK;
}

// Parameter gradients of the view-branch model from the saved activations and the
// pre-activation gradients the dX-chain kernel left in the workspace.  Every product overwrites
// its destination (no accumulation into gw / gb).
static int train_param_grads_split(const Program &p, int64_t P, const TrainWs &w, float *const *gw, float *const *gb, hipStream_t stream,
                                   const float *g_raw);

int train_param_grads(const Program &p, int64_t P, void *workspace, float *const *gw, float *const *gb, int device, hipStream_t stream,
                      bool split, const float *g_raw) {
    TrainWs w;
    carve(p, P, static_cast<char *>(workspace), &w, split);
    if (split) return train_param_grads_split(p, P, w, gw, gb, stream, g_raw);
    const int D = p.arch.D, W = p.arch.W, E = enc_row_slots(p.KE16, false), Dd = 32 * p.KD16, ic = p.input_ch, icv = p.input_ch_views;
    const bool vd = p.arch.use_viewdirs != 0;
    if (W != 256 || (E != 64 && E != 128) || (vd && Dd != 32 && Dd != 64)) return NERF_AMD_EUNSUPPORTED;
    DwSeq s;
    s.main_s = stream; s.slab = w.slab;
    constexpr int MAX_PRODUCTS = 24;
    // Measured: with the reductions on the side stream a 1024-ray step takes 2.78 ms instead of 2.38 -- 24 cross-stream
    // event hand-offs per model cost more than the 9-us reductions they hide.  Off; nerf_amd_set_tuning(0, 52) turns it on.
    s.overlap = g_variant == 52 && lane_acquire(device, 2 * MAX_PRODUCTS + 1, &s.side, &s.ev) == NERF_AMD_OK;
    // Two lanes of 128 workgroups: the products of one model are independent, and one product alone cannot keep the
    // memory system busy through its ramp, its tail and its 9-us reduction (the coarse pass even has only 128 workgroups'
    // worth of points).  Measured at 1024 rays (ms per step, same session): one lane x 256 workgroups 2.25, two x 256 2.19,
    // two x 128 2.08, two x 64 2.26, three x 128 2.28, three x 96 2.27, four x 128 2.50, four x 64 2.41.
    int n_lanes = 2;
    bool heads_first = false;        // the two head products open lane 1 instead of following the join
    bool heads_tail = true;          // ... or close lane 1, beside lane 0's last product
    s.grid_cap = 128;
    bool multi = true;               // all streaming products of the model as one launch (DwSeq::flush)
    switch (g_variant) {             // A/B
    case 56: multi = false; break;   // two lanes x 128 workgroups, one launch per product
    case 50: case 51: case 52: multi = false; n_lanes = 0; s.grid_cap = DW_GRID; heads_tail = false; break;   // one lane (51: this round's kernels)
    case 53: multi = false; s.grid_cap = DW_GRID; heads_tail = false; break;
    case 54: multi = false; heads_tail = false; break;
    case 55: multi = false; heads_tail = false; heads_first = true; break;
    default: break;
    }
    // A/B 58: a side stream per head product, both beside the one launch from its start -- 1.683 ms per step against 1.653
    // with one side stream (four interleaved rounds of 500 steps): the heads' blocks take CU slots from the streaming jobs
    const bool two_side = multi && g_variant == 58;
    if (two_side) n_lanes = 3;
    if (n_lanes > 0 && lane_acquire(device, n_lanes, &s.side, &s.ev) == NERF_AMD_OK) {
        if (lane_streams(device, n_lanes - 1, s.lane_s) == NERF_AMD_OK) s.lanes = n_lanes;
        else lane_release(device, s.ev);
    }
    if (!s.lanes) s.grid_cap = DW_GRID;
    s.multi = multi && s.lanes >= 2;
    s.mj.n = 0;
    if (s.multi) s.grid_cap = DW_GRID;
    if (s.lanes) {                                     // the other lanes start behind the backward-chain kernel
        (void)hipEventRecord(s.ev[0], stream);
        for (int l = 1; l < s.lanes; ++l) (void)hipStreamWaitEvent(s.lane_s[l - 1], s.ev[0], 0);
    }
    if (s.overlap) {                                   // the side stream starts behind the backward-chain kernel
        (void)hipEventRecord(s.ev[2 * MAX_PRODUCTS], stream);
        (void)hipStreamWaitEvent(s.side, s.ev[2 * MAX_PRODUCTS], 0);
    }
    const int64_t HS = pad_points(P) * 256;
    heads_first = heads_first && s.lanes >= 2;
    heads_tail = heads_tail && s.lanes >= 2;
    auto heads = [&](hipStream_t hs, float *slab) {    // alpha_linear (column 3 of g_rawb) and rgb_linear (columns 0..2)
        if (!vd) {
            // output_linear [out_ch, W]: X = h8, G = the out_ch <= 16 columns of g_rawb [P, 16], four rows per launch
            const uint16_t *h8x = w.sv_h + (D - 1) * HS;
            for (int r0 = 0; r0 < p.out_ch; r0 += 4) {
                const int n = p.out_ch - r0 < 4 ? p.out_ch - r0 : 4;
                float *dWr = gw[D] + (int64_t)r0 * W, *dbr = gb[D] + r0;
                if (n == 1) launch_dw_small<1>(hs, P, slab, w.g_rawb, r0, h8x, W, dWr, dbr, 16);
                else if (n == 2) launch_dw_small<2>(hs, P, slab, w.g_rawb, r0, h8x, W, dWr, dbr, 16);
                else if (n == 3) launch_dw_small<3>(hs, P, slab, w.g_rawb, r0, h8x, W, dWr, dbr, 16);
                else launch_dw_small<4>(hs, P, slab, w.g_rawb, r0, h8x, W, dWr, dbr, 16);
            }
            return;
        }
        launch_dw_small<1>(hs, P, slab, w.g_rawb, 3, w.sv_h + (D - 1) * HS, W, gw[D + 1], gb[D + 1]);
        launch_dw_small<3>(hs, P, slab, w.g_rawb, 0, w.sv_hv, W / 2, gw[D + 3], gb[D + 3]);
    };
    if (heads_first) heads(s.lane_s[0], w.slab + 2 * SLAB_FLOATS / s.lanes);   // lane 1's share of the slab, ahead of its products
    const int Lx = p.arch.multires, Ld = p.arch.multires_views;
    int rc = 0;
    for (int l = 0; l < D && !rc; ++l) {
        const uint16_t *G = w.g_h + l * HS;
        const int n_in = p.tensors[l].n_in;
        if (l == 0) {
            rc = weight_grad(s, P, w.slab, w.sv_e, E, PERM_GEN, Lx, ic, G, W, W, gw[l], n_in, 0, gb[l]);
        } else if (n_in == W + ic) {      // the layer after the skip: [input_pts | h]
            rc = weight_grad(s, P, w.slab, w.sv_e, E, PERM_GEN, Lx, ic, G, W, W, gw[l], n_in, 0, nullptr);
            if (!rc) rc = weight_grad(s, P, w.slab, w.sv_h + (l - 1) * HS, W, PERM_ACC, 0, W, G, W, W, gw[l], n_in, ic, gb[l]);
        } else {
            rc = weight_grad(s, P, w.slab, w.sv_h + (l - 1) * HS, W, PERM_ACC, 0, W, G, W, W, gw[l], n_in, 0, gb[l]);
        }
    }
    if (!vd) {
        // output_linear models: the hidden layers' products are the one launch; the head (out_ch <= 16 rows over h8) runs on
        // the side stream beside it (fp32 FMA kernels: this model family is not the one the step time is tuned on)
        if (s.lanes >= 2) heads(s.lane_s[0], w.slab + (s.multi ? SLAB_FLOATS : 2 * SLAB_FLOATS / s.lanes));
        if (!rc) rc = s.flush();
        s.join();
        if (s.overlap || s.lanes) lane_release(device, s.ev);
        if (rc) return rc;
        if (s.lanes < 2) heads(stream, w.slab);
        return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
    }
    const uint16_t *h8 = w.sv_h + (D - 1) * HS;
    // feature_linear -- and alpha_linear, whose product has the same X (h8): in the one launch its gradient column rides
    // along as a head tile (row 3 of dL/draw's columns) instead of a separate fp32 FMA kernel re-reading h8
    HeadSpec alpha_head, rgb_head;
    alpha_head.H = w.g_rawt; alpha_head.row0 = 3; alpha_head.rows = 1; alpha_head.ld = W; alpha_head.dW = gw[D + 1]; alpha_head.db = gb[D + 1];
    rgb_head.H = w.g_rawt; rgb_head.row0 = 0; rgb_head.rows = 3; rgb_head.ld = W / 2; rgb_head.dW = gw[D + 3]; rgb_head.db = gb[D + 3];
    const bool fold_heads = s.multi && g_variant != 57;       // A/B 57: the round-2 head kernels beside the one launch
    if (!rc) rc = weight_grad(s, P, w.slab, h8, W, PERM_ACC, 0, W, w.g_feat, W, W, gw[D], W, 0, gb[D], fold_heads ? &alpha_head : nullptr);
    // views_linears.0: [feature | dirs]
    if (!rc) rc = weight_grad(s, P, w.slab, w.sv_feat, W, PERM_ACC, 0, W, w.g_hv, W / 2, W / 2, gw[D + 2], W + icv, 0, gb[D + 2]);
    if (!rc) rc = weight_grad(s, P, w.slab, w.sv_d, Dd, PERM_GEN, Ld, icv, w.g_hv, W / 2, W / 2, gw[D + 2], W + icv, W, nullptr);
    // (the two head products as jobs of the one launch: 2.13 ms per step instead of 1.69 -- their fp32 FMA loops want a
    // thousand small blocks in flight, not a twentieth of the CUs)
    if (fold_heads) {
        // rgb_linear: nothing else multiplies the view layer's output, so its three gradient columns are a job of their own
        if (!rc) rc = head_grad(s, P, w.sv_hv, W / 2, PERM_ACC, W / 2, rgb_head);
    } else if (s.multi && s.lanes == 3) {
        // both heads start with the one launch and run beside it on a stream each (their blocks are small enough to share a
        // CU with a streaming workgroup); second half of the slab buffer, a quarter each
        launch_dw_small<1>(s.lane_s[0], P, w.slab + SLAB_FLOATS, w.g_rawb, 3, w.sv_h + (D - 1) * HS, W, gw[D + 1], gb[D + 1]);
        launch_dw_small<3>(s.lane_s[1], P, w.slab + SLAB_FLOATS + SLAB_FLOATS / 2, w.g_rawb, 0, w.sv_hv, W / 2, gw[D + 3], gb[D + 3]);
    } else if (heads_tail) {
        heads(s.lane_s[0], w.slab + (s.multi ? SLAB_FLOATS : 2 * SLAB_FLOATS / s.lanes));    // behind lane 1's last reduction, same slab share
    }
    if (!rc) rc = s.flush();
    s.join();
    if (s.overlap || s.lanes) lane_release(device, s.ev);
    if (rc) return rc;
    if (!fold_heads && !heads_first && !heads_tail) heads(stream, w.slab);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// The split-precision counterpart: every streaming product of the model (and, with a view branch, both head products) as
// jobs of ONE dw_multi_split_kernel launch, one reduction launch that also takes the loss scale off; output_linear's
// gradient (no view branch) from fp32 FMA kernels on dL/draw itself.
static int train_param_grads_split(const Program &p, int64_t P, const TrainWs &w, float *const *gw, float *const *gb, hipStream_t stream,
                                   const float *g_raw) {
    const int D = p.arch.D, W = p.arch.W, E = enc_row_slots(p.KE16, true), Dd = 32 * p.KD16, ic = p.input_ch, icv = p.input_ch_views;
    const bool vd = p.arch.use_viewdirs != 0;
    if (W != 256 || (E != 64 && E != 128) || (vd && Dd != 32 && Dd != 64)) return NERF_AMD_EUNSUPPORTED;
    const int Lx = p.arch.multires, Ld = p.arch.multires_views;
    const int64_t HS = pad_points(P) * 256;
    DwMulti mj;
    DwReduceMulti mr;
    mj.n = 0;
    size_t lds = 0;
    int rc = NERF_AMD_OK;
    struct Plane { const uint16_t *hi, *lo; };
    // dW[:, col_off : col_off + m_valid] (+ db) of one Linear from G [P, n_out_slots] and X [P, n_in_slots] (hi / lo planes)
    auto product = [&](Plane X, int n_in_slots, int in_kind, int in_L, int m_valid, Plane G, int n_out_slots, int n_valid, float *dW,
                       int ld_dw, int col_off, float *db, int head_row0 = -1, int head_rows = 0, int head_ld = 0, float *head_dW = nullptr,
                       float *head_db = nullptr) {
        if (rc) return;
        if (mj.n >= DW_MAX_JOBS) { rc = NERF_AMD_EINVAL; return; }
        const bool head = head_row0 >= 0;
        int shape = -1, OT = n_out_slots / 16, IT = n_in_slots / 16;
        if (OT == 16 && IT == 16) shape = head ? 4 : 0;
        else if (OT == 8 && IT == 16) shape = 1;
        else if (OT == 16 && IT == 4) shape = 2;
        else if (OT == 8 && IT == 2) shape = 3;
        else if (OT == 16 && IT == 8) shape = 6;
        else if (OT == 8 && IT == 4) shape = 7;
        else if (OT == 0 && IT == 8 && head) shape = 5;
        if (shape < 0 || (head && shape != 4 && shape != 5)) { rc = NERF_AMD_EUNSUPPORTED; return; }
        DwJob &J = mj.job[mj.n];
        J.a.G = G.hi; J.a.G_lo = G.lo; J.a.ldg = n_out_slots; J.a.X = X.hi; J.a.X_lo = X.lo; J.a.ldx = n_in_slots; J.a.P = P;
        J.a.slab = nullptr; J.a.H = head ? w.g_rawt : nullptr; J.a.H_lo = head ? w.g_rawt_lo : nullptr;
        J.shape = shape;
        DwReduceArgs &r = mr.r[mj.n];
        r.slab = nullptr; r.n_slabs = 0; r.OT = OT; r.IT = IT;
        r.dW = dW; r.ld_dw = ld_dw; r.col_off = col_off; r.db = db;
        r.out_kind = shape == 5 ? PERM_NAT : PERM_ACC; r.in_kind = in_kind; r.in_L = in_L; r.n_valid = n_valid; r.m_valid = m_valid;
        r.HT = head ? 1 : 0; r.head_dW = head_dW; r.head_db = head_db; r.head_row0 = head ? head_row0 : 0; r.head_rows = head_rows; r.head_ld = head_ld;
        r.inv_scale = w.g_scale + GRAD_SCALE_PARTS + 1;
        const size_t need = shape == 5 ? (size_t)7 * (2 * 32 * 32 * 8 + 2048) : dw_split_lds(OT, IT, head);
        if (need > lds) lds = need;
        ++mj.n;
    };
    for (int l = 0; l < D; ++l) {
        const Plane G{w.g_h + l * HS, w.g_h_lo + l * HS};
        const Plane Xe{w.sv_e, w.sv_e_lo}, Xh{w.sv_h + (l > 0 ? l - 1 : 0) * HS, w.sv_h_lo + (l > 0 ? l - 1 : 0) * HS};
        const int n_in = p.tensors[l].n_in;
        if (l == 0) {
            product(Xe, E, PERM_GEN, Lx, ic, G, W, W, gw[l], n_in, 0, gb[l]);
        } else if (n_in == W + ic) {      // the layer after the skip: [input_pts | h]
            product(Xe, E, PERM_GEN, Lx, ic, G, W, W, gw[l], n_in, 0, nullptr);
            product(Xh, W, PERM_ACC, 0, W, G, W, W, gw[l], n_in, ic, gb[l]);
        } else {
            product(Xh, W, PERM_ACC, 0, W, G, W, W, gw[l], n_in, 0, gb[l]);
        }
    }
    const Plane h8{w.sv_h + (D - 1) * HS, w.sv_h_lo + (D - 1) * HS};
    if (vd) {
        const Plane Gf{w.g_feat, w.g_feat_lo}, Ghv{w.g_hv, w.g_hv_lo};
        // feature_linear, with alpha_linear's gradient (same X = h8) riding along as a head tile: row 3 of dL/draw's columns
        product(h8, W, PERM_ACC, 0, W, Gf, W, W, gw[D], W, 0, gb[D], 3, 1, W, gw[D + 1], gb[D + 1]);
        // views_linears.0: [feature | dirs]
        product(Plane{w.sv_feat, w.sv_feat_lo}, W, PERM_ACC, 0, W, Ghv, W / 2, W / 2, gw[D + 2], W + icv, 0, gb[D + 2]);
        product(Plane{w.sv_d, w.sv_d_lo}, Dd, PERM_GEN, Ld, icv, Ghv, W / 2, W / 2, gw[D + 2], W + icv, W, nullptr);
        // rgb_linear: rows 0..2 of dL/draw's columns over the view layer's output
        product(Plane{w.sv_hv, w.sv_hv_lo}, W / 2, PERM_ACC, 0, W / 2, Plane{nullptr, nullptr}, 0, 0, nullptr, 0, 0, nullptr, 0, 3, W / 2,
                gw[D + 3], gb[D + 3]);
    }
    if (rc) return rc;
    // ---- the one launch: DW_GRID workgroups in all, shared by dw_share_workgroups
    static DynamicLdsOptIn opt_in;
    if (opt_in.ensure(reinterpret_cast<const void *>(dw_multi_split_kernel), 135168) != hipSuccess) return NERF_AMD_EHIP;
    const int64_t n_chunks = (P + 31) / 32;
    int nb[DW_MAX_JOBS], per[DW_MAX_JOBS];
    for (int j = 0; j < mj.n; ++j) per[j] = dw_slab_floats(mr.r[j].OT, mr.r[j].IT, mr.r[j].HT);
    const int cap = n_chunks / 8 < 1 ? 1 : (int)(n_chunks / 8 > DW_GRID ? DW_GRID : n_chunks / 8);
    dw_share_workgroups(mj, true, cap, nb);
    float *sl = w.slab;
    int first = 0, rfirst = 0;
    for (int j = 0; j < mj.n; ++j) {
        mj.job[j].a.slab = sl;
        mj.job[j].first_block = first; mj.job[j].n_blocks = nb[j];
        mr.r[j].slab = sl; mr.r[j].n_slabs = nb[j];
        mr.first_block[j] = rfirst;
        sl += (size_t)nb[j] * per[j];
        first += nb[j];
        rfirst += (per[j] + DWR_BLOCK - 1) / DWR_BLOCK;
    }
    mr.first_block[mj.n] = rfirst;
    mr.n = mj.n;
    if (lds > 135168) return NERF_AMD_EINVAL;
    hipLaunchKernelGGL(dw_multi_split_kernel, dim3((unsigned)first), dim3(512), lds, stream, mj);
    hipLaunchKernelGGL(dw_reduce_multi_kernel, dim3((unsigned)rfirst), dim3(512), 0, stream, mr);
    if (!vd) {
        // output_linear [out_ch, W]: X = h8, G = the columns of dL/draw (fp32), four rows per launch; the slabs go behind
        // the one launch's (its reduction has consumed them in stream order, and these are 1024 x (4 x 256 + 4) floats)
        for (int r0 = 0; r0 < p.out_ch; r0 += 4) {
            const int n = p.out_ch - r0 < 4 ? p.out_ch - r0 : 4;
            float *dWr = gw[D] + (int64_t)r0 * W, *dbr = gb[D] + r0;
            int64_t g = (P + 255) / 256;
            if (g > 1024) g = 1024;
            float *slab2 = w.slab + SLAB_FLOATS;
#define NA_LAUNCH_SMALL_SPLIT(NO)                                                                                                        \
    hipLaunchKernelGGL(dw_small_split_kernel<NO>, dim3((unsigned)g), dim3(256), 0, stream, g_raw, p.out_ch, r0, h8.hi, h8.lo, W, W, P, slab2); \
    hipLaunchKernelGGL(dw_small_reduce_kernel, dim3((NO * W + NO + 63) / 64), dim3(1024), 0, stream, slab2, (int)g, NO, W, (int)PERM_ACC, dWr, W, dbr)
            if (n == 1) { NA_LAUNCH_SMALL_SPLIT(1); }
            else if (n == 2) { NA_LAUNCH_SMALL_SPLIT(2); }
            else if (n == 3) { NA_LAUNCH_SMALL_SPLIT(3); }
            else { NA_LAUNCH_SMALL_SPLIT(4); }
#undef NA_LAUNCH_SMALL_SPLIT
        }
    }
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

}  // namespace na
