// train_f32.hip -- forward-with-saves, dX chain and weight gradients in exact fp32 for ANY architecture the reference's
// constructor accepts (NERF_AMD_PREC_FP32 training).
//
// The fused training kernels (mlp_bf16_s16.hip / mlp_split.hip SAVE, mlp_bwd_*.hip, backward.hip) cover the 8 x 256 family the
// reference's configs use; NeRF(D, W, skips, ...) of any other shape (config_parser.py:18-25 netdepth / netwidth) rendered on
// the exact-fp32 kernel and raised from backward().  This is that kernel's structure carried through the backward pass
// (nerf.py:110-134 under torch.autograd): v_mfma_f32_32x32x2_f32 everywhere, activations / gradients as LDS rows of 64 (or 32)
// points, weights as fp32 fragments straight from L2 -- fp32 MFMA rate, i.e. 1/16 of the bf16 kernels: the any-architecture
// fallback, not a tuned path.  Gradients with respect to points, rays and view directions (pose estimation) on request: the
// chain then also carries the rows of the two encodings, and their derivative is taken at the end of the kernel.
//
//   forward   f32_fwd_save_kernel   mlp_fp32.hip's layer loop; every layer's input rows X_l and, for ReLU layers, output rows
//                                   go to the workspace as [rows][pad64(P)] fp32
//   dX chain  f32_bwd_kernel        layers in reverse: g_pre(l) = relu'(y_l) g_out(l) -> workspace; g_in = W_l^T g_pre(l) on
//                                   the transposed fragment stream; written over (or added to) the hidden rows of the input
//   dW, db    f32_dw_kernel         [dW_l | db_l] = g_pre(l) [X_l ; 1]^T, 64 x 64 output blocks, the point axis split over workgroups,
//             f32_dw_reduce_kernel  partial blocks summed in a fixed order (deterministic)
#include <hip/hip_runtime.h>
#include <type_traits>

#include "kernels.h"
#include "launch_util.h"
#include "pipeline.h"
#include "program.h"

namespace na {

typedef __attribute__((ext_vector_type(16))) float f32x16t;
typedef __attribute__((ext_vector_type(4))) float f32x4t;

// Feature `col` of the reference embedding of x (nerf.py:32-41), as mlp_fp32.hip evaluates it (accurate sinf / cosf).
__device__ __forceinline__ float embed_feature_t(const float x[3], int col, int i_embed) {
    if (col < 3 || i_embed == -1) return x[col];
    const int g = col - 3, f = g / 6, rem = g % 6;
    const float arg = x[rem % 3] * __builtin_ldexpf(1.0f, f);
    return rem < 3 ? sinf(arg) : cosf(arg);
}

constexpr int64_t pad64(int64_t P) { return (P + 63) & ~(int64_t)63; }
constexpr int TF32_MAX_TPW = 5;      // 32-row tiles per wave: n_out (forward) or the hidden rows of n_in (backward) <= 8 * 5 * 32

struct TrainF32Args {
    MlpArgs a;                       // the forward's inputs and the packed fp32 streams
    const TrainLayerF32 *tl;         // device copy of Program::tlayers
    const float *stream_t;           // transposed fragment stream
    float *ws;                       // workspace [train_f32_rows][Pp]
    int64_t Pp;                      // pad64(P)
    int32_t lds_rows_bwd;
    int32_t want_enc;                // backward: dL/d(encoding rows) too (someone asked for point / ray / view-direction gradients)
};

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <int TPW, int HALVES>
__global__ __launch_bounds__(512) void f32_fwd_save_kernel(TrainF32Args t) {
    constexpr int PTS = 32 * HALVES;
    const MlpArgs &a = t.a;
    extern __shared__ __attribute__((aligned(16))) float act[];
    const int rows = a.lds_rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = lane & 31, h = lane >> 5;
    for (int i = tid; i < rows * PTS; i += 512) act[i] = 0.0f;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * PTS;
    const int n_in_rows = a.input_ch + a.input_ch_views;
    for (int i = tid; i < n_in_rows * PTS; i += 512) {
        const int row = i / PTS, q = i - row * PTS;
        int64_t p = p0 + q;
        if (p >= a.P) p = a.P - 1;
        const int64_t ray = (int64_t)((uint32_t)p / (uint32_t)a.S);
        const int dst = row < a.input_ch ? row : a.input_ch + a.W + (row - a.input_ch);
        float v[3], val;
        if (row < a.input_ch) {
            if (a.pts) {
                v[0] = a.pts[3 * p]; v[1] = a.pts[3 * p + 1]; v[2] = a.pts[3 * p + 2];
            } else {
                const float *r = a.rays + ray * a.ray_stride;
                const float z = a.z_vals[p];
                v[0] = mul_then_add(r[3], z, r[0]);               // o + d z, the product rounded first (render_utils.py:131)
                v[1] = mul_then_add(r[4], z, r[1]);
                v[2] = mul_then_add(r[5], z, r[2]);
            }
            val = embed_feature_t(v, row, a.i_embed);
        } else {
            const float *d = a.viewdirs + ray * a.vd_stride;
            v[0] = d[0]; v[1] = d[1]; v[2] = d[2];
            val = embed_feature_t(v, row - a.input_ch, a.i_embed);
        }
        act[dst * PTS + q] = val;
    }
    __syncthreads();

    for (int li = 0; li < a.n_layers; ++li) {
        const LayerF32 L = a.layers[li];
        const TrainLayerF32 T = t.tl[li];
        // ---- this layer's input rows, as the weight-gradient product will read them
        for (int i = tid; i < L.n_in * PTS; i += 512) {
            const int r = i / PTS, q = i - r * PTS;
            t.ws[(int64_t)(T.x_row + r) * t.Pp + p0 + q] = act[(L.in_row + r) * PTS + q];
        }
        const float *in = act + L.in_row * PTS + pt;
        const int tiles = (L.n_out + 31) >> 5, groups = (L.n_in + 7) >> 3;
        f32x16t acc[TPW][HALVES];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int tt = wave + 8 * u;
            if (tt < tiles) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bias = a.bias_f32[L.bias_off + 32 * tt + acc_row(r, h)];
#pragma unroll
                    for (int c = 0; c < HALVES; ++c) acc[u][c][r] = bias;
                }
                const f32x4t *wf = reinterpret_cast<const f32x4t *>(a.stream_f32 + L.frag_off + (int64_t)tt * groups * 256) + lane;
#pragma unroll 2
                for (int g = 0; g < groups; ++g) {
                    const f32x4t w = wf[(int64_t)g * 64];
                    const float *bp = in + (8 * g + h) * PTS;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int c = 0; c < HALVES; ++c)
                            acc[u][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[k], bp[2 * PTS * k + 32 * c], acc[u][c], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                       // every wave has read (and saved) this layer's inputs
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int tt = wave + 8 * u;
            if (tt < tiles) {
#pragma unroll
                for (int c = 0; c < HALVES; ++c) {
                    const int64_t p = p0 + 32 * c + pt;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * tt + acc_row(r, h);
                        if (row >= L.n_out) continue;
                        float v = acc[u][c][r];
                        if (L.relu) {
                            v = fmaxf(v, 0.0f);
                            t.ws[(int64_t)(T.y_row + row) * t.Pp + p] = v;
                        }
                        if (L.out_row >= 0) act[(L.out_row + row) * PTS + 32 * c + pt] = v;
                        else if (p < a.P) a.out[(int64_t)a.out_ch * p + L.out_col + row] = v;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dX chain
// ------------------------------------------------------------------------------------------------------------------
template <int TPW, int HALVES>
__global__ __launch_bounds__(512) void f32_bwd_kernel(TrainF32Args t) {
    constexpr int PTS = 32 * HALVES;
    const MlpArgs &a = t.a;
    extern __shared__ __attribute__((aligned(16))) float G[];
    const int rows = t.lds_rows_bwd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = lane & 31, h = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * PTS;
    for (int i = tid; i < rows * PTS; i += 512) G[i] = 0.0f;
    __syncthreads();
    // dL/draw -> the heads' output-gradient rows (zero for the padding points: they then contribute nothing anywhere)
    for (int i = tid; i < a.out_ch * PTS; i += 512) {
        const int c = i / PTS, q = i - c * PTS;
        const int64_t p = p0 + q;
        G[(a.lds_rows + c) * PTS + q] = p < a.P ? a.g_raw[(int64_t)a.out_ch * p + c] : 0.0f;
    }
    __syncthreads();
    for (int li = a.n_layers - 1; li >= 0; --li) {
        const LayerF32 L = a.layers[li];
        const TrainLayerF32 T = t.tl[li];
        // ---- g_pre = relu'(y) g_out, in place, and out to the workspace for the weight-gradient product
        for (int i = tid; i < L.n_out * PTS; i += 512) {
            const int r = i / PTS, q = i - r * PTS;
            float v = G[(T.lds_g_row + r) * PTS + q];
            if (L.relu && !(t.ws[(int64_t)(T.y_row + r) * t.Pp + p0 + q] > 0.0f)) v = 0.0f;
            G[(T.lds_g_row + r) * PTS + q] = v;
            t.ws[(int64_t)(T.g_row + r) * t.Pp + p0 + q] = v;
        }
        __syncthreads();
        // rows of the input whose gradient goes on: its hidden rows [lo, hi) (what an earlier layer produced) and, when point /
        // ray gradients are wanted, its encoding rows (LDS rows below input_ch or from input_ch + W on: always accumulated,
        // two layers read the xyz encoding)
        const int enc_lo = t.want_enc ? 0 : T.lo, enc_hi = t.want_enc ? L.n_in : T.hi;
        const int r_lo = T.hi > T.lo ? (T.lo < enc_lo ? T.lo : enc_lo) : enc_lo, r_hi = T.hi > T.lo ? (T.hi > enc_hi ? T.hi : enc_hi) : enc_hi;
        if (r_hi <= r_lo) continue;                            // nothing upstream of this layer's input wants a gradient
        // ---- g_in = W^T g_pre for the 32-row tiles of the input that touch those rows
        const int t_first = r_lo >> 5, t_last = (r_hi - 1) >> 5;
        const int groups = (L.n_out + 7) >> 3, groups_all = groups;
        const float *in = G + T.lds_g_row * PTS + pt;
        f32x16t acc[TPW][HALVES];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int tt = t_first + wave + 8 * u;
            if (tt <= t_last) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int c = 0; c < HALVES; ++c) acc[u][c][r] = 0.0f;
                const f32x4t *wf = reinterpret_cast<const f32x4t *>(t.stream_t + T.frag_off_t + (int64_t)tt * groups_all * 256) + lane;
#pragma unroll 2
                for (int g = 0; g < groups; ++g) {
                    const f32x4t w = wf[(int64_t)g * 64];
                    const float *bp = in + (8 * g + h) * PTS;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int c = 0; c < HALVES; ++c)
                            acc[u][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[k], bp[2 * PTS * k + 32 * c], acc[u][c], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                       // every wave has read g_pre (its rows may be the ones written next)
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int tt = t_first + wave + 8 * u;
            if (tt <= t_last) {
#pragma unroll
                for (int c = 0; c < HALVES; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ri = 32 * tt + acc_row(r, h);            // row of the input
                        if (ri >= L.n_in) continue;
                        const bool hidden = ri >= T.lo && ri < T.hi;
                        if (!hidden && !t.want_enc) continue;
                        float *dst = G + (L.in_row + ri) * PTS + 32 * c + pt;
                        *dst = (hidden && !T.accumulate) ? acc[u][c][r] : *dst + acc[u][c][r];
                    }
            }
        }
        __syncthreads();
    }
    if (!t.want_enc) return;
    // ---- through the two encodings (nerf.py:32-41): d sin(2^f x) = 2^f cos, d cos = -2^f sin, identity columns 1; one thread
    // per (point, coordinate, encoding); LDS rows [0, input_ch) hold dL/d(xyz encoding), [input_ch + W, ...) the directions'
    for (int i = tid; i < 2 * 3 * PTS; i += 512) {
        const int which = i / (3 * PTS), c = (i / PTS) % 3, q = i % PTS;
        if (which == 1 && !a.viewdirs) continue;
        const int64_t p = p0 + q;
        if (p >= a.P) continue;
        const int64_t ray = (int64_t)((uint32_t)p / (uint32_t)a.S);
        float x, z = 0.0f;
        if (which == 0) {
            if (a.pts) x = a.pts[3 * p + c];
            else { const float *r = a.rays + ray * a.ray_stride; z = a.z_vals[p]; x = mul_then_add(r[3 + c], z, r[c]); }
        } else {
            x = a.viewdirs[ray * a.vd_stride + c];
        }
        const int row0 = which == 0 ? 0 : a.input_ch + a.W;
        const int Lf = a.i_embed == -1 ? 0 : (which == 0 ? a.multires : a.multires_views);
        float g = G[(row0 + c) * PTS + q];                     // the raw coordinate's column
        for (int f = 0; f < Lf; ++f) {
            const float sc = __builtin_ldexpf(1.0f, f), arg = x * sc;
            g += G[(row0 + 3 + 6 * f + c) * PTS + q] * (sc * cosf(arg));
            g -= G[(row0 + 3 + 6 * f + 3 + c) * PTS + q] * (sc * sinf(arg));
        }
        if (which == 0) {
            if (a.g_pts) a.g_pts[3 * p + c] = g;
            if (a.g_rays) { atomicAdd(a.g_rays + ray * 6 + c, g); atomicAdd(a.g_rays + ray * 6 + 3 + c, g * z); }
        } else if (a.g_vd) {
            atomicAdd(a.g_vd + ray * 3 + c, g);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dW_l = g_pre(l) X_l^T over the points; db_l = row sums of g_pre(l)
// ------------------------------------------------------------------------------------------------------------------
// One workgroup (4 waves, a 2 x 2 arrangement of 32 x 32 MFMA tiles): the 64 x 64 block (bo, bi) of [dW | db] over the points
// [slice * chunk_pts, ...).  X has one more row than the layer has inputs, all ones: column n_in of the product is db (the
// padding points' g_pre is zero).  Rows of G (n_out) and X past that read as zero.
constexpr int DWF_KP = 32;           // points per staged chunk
__global__ __launch_bounds__(256) void f32_dw_kernel(const float *G, const float *X, int n_out, int n_in, int64_t Pp, int64_t pts_per_slice,
                                                     int blocks_i, float *slab) {
    __shared__ float sg[64][DWF_KP + 1], sx[64][DWF_KP + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bo = blockIdx.x / blocks_i, bi = blockIdx.x % blocks_i, slice = blockIdx.y;
    const int wo = wave >> 1, wi = wave & 1;
    const int64_t pa = (int64_t)slice * pts_per_slice, pb = pa + pts_per_slice < Pp ? pa + pts_per_slice : Pp;
    f32x16t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int l32 = lane & 31, hh = lane >> 5;
    // 64 rows x 32 points of each operand per step (coalesced: 32 consecutive points of a row per half wave), the next
    // step's values fetched into registers while this step's are multiplied
    float rg[8], rx[8];
    auto fetch = [&](int64_t p) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j, r = i / DWF_KP, q = i % DWF_KP;
            const int ro = 64 * bo + r, rxi = 64 * bi + r;
            const bool in = p + q < pb;
            rg[j] = (in && ro < n_out) ? G[(int64_t)ro * Pp + p + q] : 0.0f;
            rx[j] = in ? (rxi < n_in ? X[(int64_t)rxi * Pp + p + q] : rxi == n_in ? 1.0f : 0.0f) : 0.0f;     // row n_in: ones -> the bias gradient
        }
    };
    if (pa < pb) fetch(pa);
    for (int64_t p = pa; p < pb; p += DWF_KP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j, r = i / DWF_KP, q = i % DWF_KP;
            sg[r][q] = rg[j];
            sx[r][q] = rx[j];
        }
        __syncthreads();
        if (p + DWF_KP < pb) fetch(p + DWF_KP);
#pragma unroll
        for (int k = 0; k < DWF_KP; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sg[32 * wo + l32][k + hh], sx[32 * wi + l32][k + hh], acc, 0, 0, 0);
        __syncthreads();
    }
    // partial block: [slice][block][64][64]
    float *out = slab + ((int64_t)slice * gridDim.x + blockIdx.x) * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(32 * wo + acc_row(r, hh)) * 64 + 32 * wi + l32] = acc[r];
}

__global__ __launch_bounds__(256) void f32_dw_reduce_kernel(const float *slab, int n_slices, int n_blocks, int blocks_i, int n_out, int n_in,
                                                            float *dW, float *db) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;        // element of the [n_blocks][64][64] result
    if (e >= (int64_t)n_blocks * 4096) return;
    const int b = (int)(e / 4096), r = (int)(e % 4096) / 64, c = (int)(e % 64);
    const int o = 64 * (b / blocks_i) + r, i = 64 * (b % blocks_i) + c;
    if (o >= n_out || i > n_in) return;
    float s = 0.0f;
    for (int k = 0; k < n_slices; ++k) s += slab[((int64_t)k * n_blocks + b) * 4096 + r * 64 + c];      // fixed order: deterministic
    if (i < n_in) dW[(int64_t)o * n_in + i] = s;
    else db[o] = s;
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct Shape { int halves, tpw_f, tpw_b; };

bool pick_shape(const Program &p, Shape *s) {
    const size_t rows_f = p.lds_rows, rows_b = p.lds_rows_bwd;
    s->halves = rows_b * 64 * sizeof(float) <= 160 * 1024 ? 2 : 1;
    if (rows_b * 32 * s->halves * sizeof(float) > 160 * 1024) return false;
    (void)rows_f;
    int widest = 0, widest_b = 0;
    for (size_t l = 0; l < p.layers.size(); ++l) {
        widest = std::max(widest, (p.layers[l].n_out + 31) / 32);
        widest_b = std::max(widest_b, (p.layers[l].n_in + 31) / 32);      // with encoding gradients the whole input is written
    }
    s->tpw_f = (widest + 7) / 8;
    s->tpw_b = std::max(1, (widest_b + 7) / 8);
    return s->tpw_f <= TF32_MAX_TPW && s->tpw_b <= TF32_MAX_TPW;
}

// points per weight-gradient workgroup: 1024, more when the point axis would otherwise need more slices than a grid has rows
int64_t dw_slice_pts(int64_t Pp) {
    int64_t pts = 1024;
    while ((Pp + pts - 1) / pts > 32768) pts *= 2;
    return pts;
}
int64_t slab_floats(const Program &p, int64_t Pp) {
    const int64_t slices = (Pp + dw_slice_pts(Pp) - 1) / dw_slice_pts(Pp);
    int64_t worst = 0;
    for (const LayerF32 &L : p.layers) worst = std::max<int64_t>(worst, (int64_t)((L.n_out + 63) / 64) * ((L.n_in + 1 + 63) / 64));
    return slices * worst * 4096;
}

template <class F>
int dispatch(int tpw, int halves, F &&f) {
    using std::integral_constant;
#define NA_TF32_CASE(T)                                                                                         \
    case T: return halves == 2 ? f(integral_constant<int, T>{}, integral_constant<int, 2>{}) : f(integral_constant<int, T>{}, integral_constant<int, 1>{});
    switch (tpw) {
        NA_TF32_CASE(1) NA_TF32_CASE(2) NA_TF32_CASE(3) NA_TF32_CASE(4) NA_TF32_CASE(5)
    }
#undef NA_TF32_CASE
    return NERF_AMD_EUNSUPPORTED;
}
}  // namespace

bool train_f32_supported(const Program &p) {
    Shape s;
    return !p.tlayers.empty() && pick_shape(p, &s);
}

int64_t train_f32_workspace_bytes(const Program &p, int64_t P) {
    const int64_t Pp = pad64(P);
    return ((int64_t)p.train_f32_rows * Pp + slab_floats(p, Pp)) * (int64_t)sizeof(float) + 256;
}

int launch_train_f32_forward(const Program &p, const MlpArgs &a, const TrainLayerF32 *d_tl, void *workspace, hipStream_t s) {
    Shape sh;
    if (!pick_shape(p, &sh)) return NERF_AMD_EUNSUPPORTED;
    if (a.P <= 0) return NERF_AMD_OK;
    if (a.P >= (int64_t)1 << 31) return NERF_AMD_EINVAL;
    TrainF32Args t;
    t.a = a; t.tl = d_tl; t.stream_t = nullptr; t.ws = static_cast<float *>(workspace); t.Pp = pad64(a.P); t.lds_rows_bwd = p.lds_rows_bwd;
    t.want_enc = 0;
    const int pts = 32 * sh.halves;
    const size_t lds = (size_t)p.lds_rows * pts * sizeof(float);
    const int64_t blocks = t.Pp / pts;
    return dispatch(sh.tpw_f, sh.halves, [&](auto tpw_, auto halves_) -> int {
        constexpr int TPW = decltype(tpw_)::value, HALVES = decltype(halves_)::value;
        static DynamicLdsOptIn opt_in;
        if (opt_in.ensure(reinterpret_cast<const void *>(f32_fwd_save_kernel<TPW, HALVES>), 160 * 1024) != hipSuccess) return NERF_AMD_EHIP;
        hipLaunchKernelGGL((f32_fwd_save_kernel<TPW, HALVES>), dim3((unsigned)blocks), dim3(512), lds, s, t);
        return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
    });
}

int launch_train_f32_backward(const Program &p, const MlpArgs &a, const TrainLayerF32 *d_tl, const float *stream_t, void *workspace,
                              float *const *gw, float *const *gb, hipStream_t s) {
    Shape sh;
    if (!pick_shape(p, &sh)) return NERF_AMD_EUNSUPPORTED;
    if (a.P <= 0) return NERF_AMD_OK;
    TrainF32Args t;
    t.a = a; t.tl = d_tl; t.stream_t = stream_t; t.ws = static_cast<float *>(workspace); t.Pp = pad64(a.P); t.lds_rows_bwd = p.lds_rows_bwd;
    t.want_enc = (a.g_pts || a.g_rays || a.g_vd) ? 1 : 0;
    const int pts = 32 * sh.halves;
    const size_t lds = (size_t)p.lds_rows_bwd * pts * sizeof(float);
    const int64_t blocks = t.Pp / pts;
    int rc = dispatch(sh.tpw_b, sh.halves, [&](auto tpw_, auto halves_) -> int {
        constexpr int TPW = decltype(tpw_)::value, HALVES = decltype(halves_)::value;
        static DynamicLdsOptIn opt_in;
        if (opt_in.ensure(reinterpret_cast<const void *>(f32_bwd_kernel<TPW, HALVES>), 160 * 1024) != hipSuccess) return NERF_AMD_EHIP;
        hipLaunchKernelGGL((f32_bwd_kernel<TPW, HALVES>), dim3((unsigned)blocks), dim3(512), lds, s, t);
        return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
    });
    if (rc) return rc;
    // ---- weight and bias gradients, layer by layer (the slab is re-used: everything is in stream order)
    float *slab = t.ws + (int64_t)p.train_f32_rows * t.Pp;
    const int64_t slice_pts = dw_slice_pts(t.Pp), slices = (t.Pp + slice_pts - 1) / slice_pts;
    for (size_t l = 0; l < p.layers.size(); ++l) {
        const LayerF32 &L = p.layers[l];
        const TrainLayerF32 &T = p.tlayers[l];
        const float *G = t.ws + (int64_t)T.g_row * t.Pp, *X = t.ws + (int64_t)T.x_row * t.Pp;
        const int blocks_o = (L.n_out + 63) / 64, blocks_i = (L.n_in + 1 + 63) / 64, nb = blocks_o * blocks_i;      // + 1: the ones row
        hipLaunchKernelGGL(f32_dw_kernel, dim3((unsigned)nb, (unsigned)slices), dim3(256), 0, s, G, X, L.n_out, L.n_in, t.Pp, slice_pts, blocks_i, slab);
        hipLaunchKernelGGL(f32_dw_reduce_kernel, dim3((unsigned)(((int64_t)nb * 4096 + 255) / 256)), dim3(256), 0, s, slab, (int)slices, nb, blocks_i,
                           L.n_out, L.n_in, gw[L.tensor], gb[L.tensor]);
    }
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

}  // namespace na
