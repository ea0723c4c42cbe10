// mlp_fp32.hip -- exact-fp32 NeRF field for any reference architecture, and the
// standalone positional encoding.
//
// Replaces Embedder.embed (/root/reference/nerf_shared/nerf.py:16-41) and
// NeRF.forward + NeRF.MLP (nerf.py:96-134) for every D / W (2..1024, any value; 32 points per workgroup instead of 64
// where the feature rows would not fit LDS) /
// skips / multires / viewdirs combination the reference constructor accepts.
// This is the parity path (v_mfma_f32_32x32x2_f32 is a bit-exact fp32 fma
// chain) and the fallback for architectures the fused bf16 kernel does not
// cover; it is still MFMA code, just at the fp32 rate (1/16 of bf16).
//
// The network is evaluated transposed (weights = A operand, points = B operand, v_mfma_f32_32x32x2_f32);
// LDS rows are features:
//   [0, input_ch)                    encoded xyz      (kept for the skip concat)
//   [input_ch, input_ch+W)           hidden
//   [input_ch+W, +input_ch_views)    encoded view dir (kept for the view concat)
// so both torch.cat calls of the reference (nerf.py:117-118, :123) are just a different first row.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "kernels.h"
#include "launch_util.h"
#include "program.h"

namespace na {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Feature `col` of the reference embedding of x (nerf.py:32-41): accurate sinf/cosf.
__device__ __forceinline__ float embed_feature(const float x[3], int col, int i_embed) {
    if (col < 3 || i_embed == -1) return x[col];
    const int g = col - 3, f = g / 6, rem = g % 6;
    const float arg = x[rem % 3] * __builtin_ldexpf(1.0f, f);   // 2^f exact, as 2.**linspace in fp32
    return rem < 3 ? sinf(arg) : cosf(arg);
}

__global__ __launch_bounds__(256) void embed_kernel(const float *x, int64_t n, int multires, float *out) {
    const int dim = 3 + 6 * multires;
    const int64_t total = n * dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / dim;
        const int col = (int)(i - p * dim);
        const float v[3] = {x[3 * p], x[3 * p + 1], x[3 * p + 2]};
        out[i] = embed_feature(v, col, 0);
    }
}

int launch_embed(const float *x, int64_t n, int multires, float *out, hipStream_t s) {
    if (n <= 0) return NERF_AMD_OK;
    const int64_t total = n * (3 + 6 * multires);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, multires, out);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// One workgroup (8 waves) evaluates 64 points = two 32-point column halves (HALVES = 2), or one half when the model's
// feature rows x 64 points would not fit the CU's 160 KiB (HALVES = 1: W above ~540, up to 1024 with the largest
// encodings).  Activations live in ONE LDS buffer [feature row][PTS points] fp32 and layers update it in place: every wave first computes all of its
// output tiles of a layer into registers (tile t belongs to wave t % 8; both column halves share the weight
// fragment loaded from L2, which is what bounds this kernel), a barrier ends the reads, the tiles are
// written over the layer's input rows, a second barrier publishes them.  The encoded xyz and view rows are
// never overwritten, so both concats of the reference are a different first row.
constexpr int F32_MAX_TILES_PER_WAVE = 4;            // n_out <= 8 * 4 * 32 = 1024

#ifdef NERF_AMD_STAMPS
// tools/micro/f32_xcd_ends.py: when the first workgroup on each XCD started and the last one ended (100 MHz clock)
__device__ unsigned long long g_f32_xcd[2 * 8];
extern "C" int nerf_amd_x_f32_xcd(unsigned long long *out, int reset) {
    if (reset) {
        unsigned long long init[16];
        for (int i = 0; i < 8; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
        return hipMemcpyToSymbol(HIP_SYMBOL(g_f32_xcd), init, sizeof(init)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f32_xcd), sizeof(g_f32_xcd)) == hipSuccess ? 0 : -1;
}
#endif

// TPW: 32-row output tiles per wave (n_out <= 8 * 32 * TPW): the accumulators of all of a wave's tiles stay in registers from
// the compute phase to the write-back, so the instantiation for W <= 256 holds 32 of them instead of 128.
template <int TPW, int HALVES>
__global__ __launch_bounds__(512) void mlp_f32_kernel(MlpArgs a) {
    constexpr int PTS = 32 * HALVES;                     // points per workgroup = floats per LDS row
    extern __shared__ __attribute__((aligned(16))) float act[];
    const int rows = a.lds_rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = lane & 31, h = lane >> 5;

#ifdef NERF_AMD_STAMPS
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    xcc_id &= 7;
    if (tid == 0) atomicMin(&g_f32_xcd[2 * xcc_id], wall_clock64());
#endif
    for (int i = tid; i < rows * PTS; i += 512) act[i] = 0.0f;
    __syncthreads();

    // ---- encode this tile's points
    const int64_t p0 = (int64_t)blockIdx.x * PTS;
    const int n_in_rows = a.input_ch + a.input_ch_views;
    for (int i = tid; i < n_in_rows * PTS; i += 512) {
        const int row = i / PTS, q = i - row * PTS;
        int64_t p = p0 + q;
        if (p >= a.P) p = a.P - 1;
        const int64_t ray = (int64_t)((uint32_t)p / (uint32_t)a.S);
        const int dst = row < a.input_ch ? row : a.input_ch + a.W + (row - a.input_ch);
        float v[3];
        float val;
        if (a.embedded) {
            val = a.embedded[p * n_in_rows + row];
        } else if (row < a.input_ch) {
            if (a.pts) {
                v[0] = a.pts[3 * p]; v[1] = a.pts[3 * p + 1]; v[2] = a.pts[3 * p + 2];
            } else {
                const float *r = a.rays + ray * a.ray_stride;
                const float z = a.z_vals[p];
                v[0] = __fadd_rn(r[0], __fmul_rn(r[3], z));
                v[1] = __fadd_rn(r[1], __fmul_rn(r[4], z));
                v[2] = __fadd_rn(r[2], __fmul_rn(r[5], z));
            }
            val = embed_feature(v, row, a.i_embed);
        } else {
            const float *d = a.viewdirs + ray * a.vd_stride;
            v[0] = d[0]; v[1] = d[1]; v[2] = d[2];
            val = embed_feature(v, row - a.input_ch, a.i_embed);
        }
        act[dst * PTS + q] = val;
    }
    __syncthreads();

    for (int li = 0; li < a.n_layers; ++li) {
        const LayerF32 L = a.layers[li];
        if (TPW < F32_MAX_TILES_PER_WAVE && L.out_row < 0 && L.n_out <= 8) {      // (the widest instantiation has no registers to spare)
            // A head that leaves for global memory (alpha_linear: 1 row, rgb_linear: 3, output_linear <= 8): as a 32 x 32 x 2 tile
            // it is 1-8 live rows of 32 on one wave, 64 matrix-pipe cycles per k-pair, while seven waves wait at the layer's
            // barriers (in-kernel stamps: 7 % of a tile's cycles for the two heads of the view-branch model).
            // v_mfma_f32_4x4x1_16B_f32 fits it: sixteen 4 x 4 blocks = 4 output rows x 64 points per instruction, one k at a
            // time, 8 cycles each -- the B operand of lane l is simply point l's activation (one conflict-free ds_read_b32
            // per k), D register i of lane l is output row i of point l, and lane l's A operand is W[l & 3][k], eight k's
            // per pair of 16-byte loads from the fragment stream.  Wave 0 does it alone and no barrier is needed: a head
            // only reads rows the previous layer's trailing barrier published and writes to global memory; the next
            // layer's first barrier (which wave 0 joins after its own tile) still precedes any overwrite of those rows.
            if (wave == 0) {
                const float *wrow = a.stream_f32 + L.frag_off + 4 * (lane & 3);        // rows 0..3 of tile 0; rows 4..7 are 16 floats on
                const int groups8 = (L.n_in + 7) >> 3;
                const bool two = L.n_out > 4;
#pragma unroll
                for (int pg = 0; pg < (PTS + 63) / 64; ++pg) {                         // 64 points per pass (three halves: two passes)
                    // accumulators start at the bias, like every other layer of the chain and like this head in the widest
                    // instantiation (D register i = output row i; the bias table is zero-padded past n_out)
                    f32x4 d0 = *reinterpret_cast<const f32x4 *>(a.bias_f32 + L.bias_off);
                    f32x4 d1 = *reinterpret_cast<const f32x4 *>(a.bias_f32 + L.bias_off + 4);
                    const int q = 64 * pg + lane;                                      // this lane's point of the workgroup
                    const float *xcol = act + L.in_row * PTS + (q < PTS ? q : q - 32);  // lanes past the last point repeat real ones
                    for (int g = 0; g < groups8; ++g) {
                        const f32x4 we = *reinterpret_cast<const f32x4 *>(wrow + g * 256);          // W[o][8g + 0, 2, 4, 6]
                        const f32x4 wo = *reinterpret_cast<const f32x4 *>(wrow + g * 256 + 128);    // W[o][8g + 1, 3, 5, 7]
                        f32x4 we2 = we, wo2 = wo;
                        if (two) {
                            we2 = *reinterpret_cast<const f32x4 *>(wrow + g * 256 + 16);
                            wo2 = *reinterpret_cast<const f32x4 *>(wrow + g * 256 + 128 + 16);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {              // k = 8g + 2i, 8g + 2i + 1 (columns past n_in are zero in the stream)
                            const float x0 = xcol[(8 * g + 2 * i) * PTS], x1 = xcol[(8 * g + 2 * i + 1) * PTS];
                            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(we[i], x0, d0, 0, 0, 0);
                            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wo[i], x1, d0, 0, 0, 0);
                            if (two) {
                                d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(we2[i], x0, d1, 0, 0, 0);
                                d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wo2[i], x1, d1, 0, 0, 0);
                            }
                        }
                    }
                    const int64_t p = p0 + q;
                    if (q < PTS && p < a.P) {
#pragma unroll
                        for (int o = 0; o < 8; ++o)
                            if (o < L.n_out) {
                                float v = o < 4 ? d0[o & 3] : d1[o & 3];
                                if (L.relu) v = fmaxf(v, 0.0f);
                                a.out[(int64_t)a.out_ch * p + L.out_col + o] = v;
                            }
                    }
                }
            }
            continue;
        }
        const float *in = act + L.in_row * PTS + pt;
        const int tiles = (L.n_out + 31) >> 5, groups = (L.n_in + 7) >> 3;
        f32x16 acc[TPW][HALVES];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int t = wave + 8 * u;
            if (t < tiles) {                                   // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bias = a.bias_f32[L.bias_off + 32 * t + acc_row(r, h)];
#pragma unroll
                    for (int c = 0; c < HALVES; ++c) acc[u][c][r] = bias;
                }
                const f32x4 *wf = reinterpret_cast<const f32x4 *>(a.stream_f32 + L.frag_off + (int64_t)t * groups * 256) + lane;
#pragma unroll 2
                for (int g = 0; g < groups; ++g) {
                    const f32x4 w = wf[(int64_t)g * 64];
                    const float *bp = in + (8 * g + h) * PTS;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {              // rows 8g + h + 2k: the k-th pair of this 8-row group
#pragma unroll
                        for (int c = 0; c < HALVES; ++c)
                            acc[u][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[k], bp[2 * PTS * k + 32 * c], acc[u][c], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                                       // every wave has read this layer's inputs
        // Write-back.  Everything that decides HOW a value is written is uniform over the wave (the layer's ReLU flag, LDS or
        // global destination, whether the 32-row tile is complete), so it is decided once per tile and the common case -- a
        // complete tile of a hidden layer -- is 32 unconditional LDS stores per lane.  (Written as one loop with the tests
        // inside, the compiler emitted ~20 instructions and three branches per value: tools/tile stamps put the write-back at
        // 10 % of a tile's cycles.)
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int t = wave + 8 * u;
            if (t < tiles) {
                if (L.relu) {
#pragma unroll
                    for (int c = 0; c < HALVES; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[u][c][r] = fmaxf(acc[u][c][r], 0.0f);
                }
                const bool complete = 32 * t + 32 <= L.n_out;
                if (L.out_row >= 0 && complete) {
                    float *dst = act + (L.out_row + 32 * t + 4 * h) * PTS + pt;
#pragma unroll
                    for (int c = 0; c < HALVES; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * PTS + 32 * c] = acc[u][c][r];
                } else {
#pragma unroll
                    for (int c = 0; c < HALVES; ++c) {
                        const int64_t p = p0 + 32 * c + pt;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = 32 * t + acc_row(r, h);
                            if (row < L.n_out) {
                                if (L.out_row >= 0) act[(L.out_row + row) * PTS + 32 * c + pt] = acc[u][c][r];
                                else if (p < a.P) a.out[(int64_t)a.out_ch * p + L.out_col + row] = acc[u][c][r];
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
#ifdef NERF_AMD_STAMPS
    if (tid == 0) atomicMax(&g_f32_xcd[2 * xcc_id + 1], wall_clock64());
#endif
}

int launch_mlp_f32(const MlpArgs &a, hipStream_t s) {
    if (a.P <= 0) return NERF_AMD_OK;
    if (a.P >= (int64_t)1 << 31) return NERF_AMD_EINVAL;
    if (a.W > 8 * F32_MAX_TILES_PER_WAVE * 32) return NERF_AMD_EUNSUPPORTED;
    // 64 points per workgroup while the activation rows fit the CU's LDS, 32 for the widest models (build_program bounds the
    // rows).  nerf_amd_set_tuning(0, 61) = A/B: 96 points (three 32-point column halves, every weight fragment fetched from
    // L2 feeds three MFMAs per k) where the rows fit -- measured in round 4 (tools/micro/f32_pts_ab.py, same process,
    // bit-identical outputs): 106.0 TFLOP/s against 114.6 for 64 points, i.e. weight delivery per point is NOT what holds
    // this kernel at 0.73 of its peak.
    const bool half = (size_t)a.lds_rows * 64 * sizeof(float) > 160 * 1024;
    const bool three = !half && g_variant == 61 && (size_t)a.lds_rows * 96 * sizeof(float) <= 160 * 1024 && (a.W > a.out_ch ? a.W : a.out_ch) <= 256;
    const int pts = half ? 32 : three ? 96 : 64;
    const size_t lds = (size_t)a.lds_rows * pts * sizeof(float);
    if (lds > 160 * 1024) return NERF_AMD_EUNSUPPORTED;
    const int64_t blocks = (a.P + pts - 1) / pts;
    const int widest = a.W > a.out_ch ? a.W : a.out_ch;      // rows of the widest layer
    auto go = [&](auto tpw_, auto halves_) -> int {
        constexpr int TPW = decltype(tpw_)::value, HALVES = decltype(halves_)::value;
        static DynamicLdsOptIn opt_in;     // the size depends on the model: raise the limit to the CU's 160 KiB once per device
        if (opt_in.ensure(reinterpret_cast<const void *>(mlp_f32_kernel<TPW, HALVES>), 160 * 1024) != hipSuccess) return NERF_AMD_EHIP;
        hipLaunchKernelGGL((mlp_f32_kernel<TPW, HALVES>), dim3((unsigned)blocks), dim3(512), lds, s, a);
        return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
    };
    using std::integral_constant;
    if (half) {
        if (widest <= 512) return go(integral_constant<int, 2>{}, integral_constant<int, 1>{});
        return go(integral_constant<int, F32_MAX_TILES_PER_WAVE>{}, integral_constant<int, 1>{});
    }
    if (three) return go(integral_constant<int, 1>{}, integral_constant<int, 3>{});
    if (widest <= 256) return go(integral_constant<int, 1>{}, integral_constant<int, 2>{});
    if (widest <= 512) return go(integral_constant<int, 2>{}, integral_constant<int, 2>{});
    return go(integral_constant<int, F32_MAX_TILES_PER_WAVE>{}, integral_constant<int, 2>{});
}

}  // namespace na
