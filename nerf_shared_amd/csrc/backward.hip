// backward.hip -- training support for the fused view-branch (10,4) field: workspace layout,
// training forward (activations saved), and the backward pass
//   dL/draw -> mlp_bwd_s16_kernel (pre-activation gradients of every layer, in registers)
//           -> weight gradients  dW_l = g_pre(l)^T h_(l-1)   (plain GEMMs with K = #points: rocBLAS)
//           -> bias gradients    db_l = column sums of g_pre(l)
// Saved activations and gradients are slot-major bf16 rows (kernels.h), so dW comes out with
// permuted rows/columns; unpermute_kernel scatters it into the nn.Linear layout.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <mutex>

#include "kernels.h"
#include "program.h"

namespace na {

enum { PERM_NAT = 0, PERM_ACC = 1, PERM_GEN = 2 };

__device__ __forceinline__ int slot_to_feature(int kind, int s, int L) {
    if (kind == PERM_NAT) return s;
    const int ks = s >> 5, q = (s >> 3) & 3, j = s & 7;
    return kind == PERM_ACC ? acc16_col(ks, q, j) : gen16_col(ks, q, j, L);
}

// dst[out_feature][col_off + in_feature] = src[o_slot][i_slot]   (src is n x m row-major, ld = m)
__global__ __launch_bounds__(256) void unpermute_kernel(const float *src, int n, int m, int out_kind, int in_kind,
                                                        int in_L, int n_valid, int m_valid, float *dst, int dst_ld,
                                                        int col_off) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * m) return;
    const int o_slot = idx / m, i_slot = idx - o_slot * m;
    const int o = slot_to_feature(out_kind, o_slot, 0), i = slot_to_feature(in_kind, i_slot, in_L);
    if (o < 0 || i < 0 || o >= n_valid || i >= m_valid) return;
    dst[(int64_t)o * dst_ld + col_off + i] = src[idx];
}

// out[feature(slot)] = sum over rows of G[row][col0 + slot]  (G bf16, ld elements per row)
__global__ __launch_bounds__(256) void colsum_kernel(const uint16_t *G, int64_t P, int ld, int col0, int n, int kind,
                                                     int n_valid, float *out) {
    const int t = threadIdx.x;
    if (t >= n) return;
    const int64_t r0 = (int64_t)blockIdx.x * 2048, r1 = r0 + 2048 < P ? r0 + 2048 : P;
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        const unsigned u = (unsigned)G[r * ld + col0 + t] << 16;
        acc += __builtin_bit_cast(float, u);
    }
    const int o = slot_to_feature(kind, t, 0);
    if (o >= 0 && o < n_valid) atomicAdd(out + o, acc);
}

namespace {
std::mutex g_blas_mu;
rocblas_handle g_blas = nullptr;

struct TrainWs {
    uint16_t *sv_e, *sv_d, *sv_h, *sv_feat, *sv_hv, *g_rawb, *g_hv, *g_feat, *g_h;
    float *scratch;     // 256 x 320 fp32 GEMM output in slot order
};
size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

int64_t carve(const Program &p, int64_t P, char *base, TrainWs *w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { char *q = base ? base + off : nullptr; off += al(bytes); return q; };
    const size_t e = 32 * p.KE16, d = 32 * p.KD16;
    TrainWs t;
    t.sv_e = (uint16_t *)take(P * e * 2);
    t.sv_d = (uint16_t *)take(P * d * 2);
    t.sv_h = (uint16_t *)take((size_t)8 * P * 256 * 2);
    t.sv_feat = (uint16_t *)take(P * 256 * 2);
    t.sv_hv = (uint16_t *)take(P * 128 * 2);
    t.g_rawb = (uint16_t *)take(P * 4 * 2);
    t.g_hv = (uint16_t *)take(P * 128 * 2);
    t.g_feat = (uint16_t *)take(P * 256 * 2);
    t.g_h = (uint16_t *)take((size_t)8 * P * 256 * 2);
    t.scratch = (float *)take(256 * 320 * sizeof(float));
    if (w) *w = t;
    return (int64_t)off;
}
}  // namespace

bool train_supported(const Program &p) {
    const nerf_amd_arch &a = p.arch;
    return p.bf16_ok && a.use_viewdirs && a.i_embed == 0 && a.multires == 10 && a.multires_views == 4;
}

int64_t train_workspace_bytes(const Program &p, int64_t P) { return carve(p, P, nullptr, nullptr); }

void train_fill_args(const Program &p, int64_t P, void *workspace, MlpArgs *a) {
    TrainWs w;
    carve(p, P, static_cast<char *>(workspace), &w);
    a->sv_e = w.sv_e; a->sv_d = w.sv_d; a->sv_h = w.sv_h; a->sv_feat = w.sv_feat; a->sv_hv = w.sv_hv;
    a->g_rawb = w.g_rawb; a->g_hv = w.g_hv; a->g_feat = w.g_feat; a->g_h = w.g_h;
}

// dW (n_out x n_in slice) = G^T X over P rows, then scatter to the nn.Linear layout.
static int weight_grad(rocblas_handle h, hipStream_t s, int64_t P, const uint16_t *X, int ldx, int m, int in_kind,
                       int in_L, int m_valid, const uint16_t *G, int ldg, int n, int out_kind, int n_valid,
                       float *scratch, float *dst, int dst_ld, int col_off) {
    const float one = 1.0f, zero = 0.0f;
    // column-major view: C[m x n] = X'[m x P] * (G'[n x P])^T, i.e. row-major C[n][m]
    rocblas_status st = rocblas_gemm_ex(h, rocblas_operation_none, rocblas_operation_transpose, m, n, (rocblas_int)P, &one,
                                        X, rocblas_datatype_bf16_r, ldx, G, rocblas_datatype_bf16_r, ldg, &zero,
                                        scratch, rocblas_datatype_f32_r, m, scratch, rocblas_datatype_f32_r, m,
                                        rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    if (st != rocblas_status_success) return NERF_AMD_EHIP;
    hipLaunchKernelGGL(unpermute_kernel, dim3((n * m + 255) / 256), dim3(256), 0, s, scratch, n, m, out_kind, in_kind,
                       in_L, n_valid, m_valid, dst, dst_ld, col_off);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

static int bias_grad(hipStream_t s, int64_t P, const uint16_t *G, int ldg, int col0, int n, int kind, int n_valid, float *dst) {
    if (hipMemsetAsync(dst, 0, n_valid * sizeof(float), s) != hipSuccess) return NERF_AMD_EHIP;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((P + 2047) / 2048)), dim3(256), 0, s, G, P, ldg, col0, n, kind, n_valid, dst);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// Parameter gradients of the view-branch (10,4) model from the saved activations and the
// pre-activation gradients the dX-chain kernel left in the workspace.
int train_param_grads(const Program &p, int64_t P, void *workspace, float *const *gw, float *const *gb, hipStream_t s) {
    TrainWs w;
    carve(p, P, static_cast<char *>(workspace), &w);
    std::lock_guard<std::mutex> lk(g_blas_mu);
    if (!g_blas && rocblas_create_handle(&g_blas) != rocblas_status_success) return NERF_AMD_EHIP;
    if (rocblas_set_stream(g_blas, s) != rocblas_status_success) return NERF_AMD_EHIP;
    rocblas_set_pointer_mode(g_blas, rocblas_pointer_mode_host);
    const int D = p.arch.D, W = p.arch.W, E = 32 * p.KE16, Dd = 32 * p.KD16, ic = p.input_ch, icv = p.input_ch_views;
    const int64_t HS = P * 256;
    const int Lx = p.arch.multires, Ld = p.arch.multires_views;
    int rc = 0;
    for (int l = 0; l < D && !rc; ++l) {
        const uint16_t *G = w.g_h + l * HS;
        const int n_in = p.tensors[l].n_in;
        if (l == 0) {
            rc = weight_grad(g_blas, s, P, w.sv_e, E, E, PERM_GEN, Lx, ic, G, W, W, PERM_ACC, W, w.scratch, gw[l], n_in, 0);
        } else if (n_in == W + ic) {      // the layer after the skip: [input_pts | h]
            rc = weight_grad(g_blas, s, P, w.sv_e, E, E, PERM_GEN, Lx, ic, G, W, W, PERM_ACC, W, w.scratch, gw[l], n_in, 0);
            if (!rc) rc = weight_grad(g_blas, s, P, w.sv_h + (l - 1) * HS, W, W, PERM_ACC, 0, W, G, W, W, PERM_ACC, W, w.scratch, gw[l], n_in, ic);
        } else {
            rc = weight_grad(g_blas, s, P, w.sv_h + (l - 1) * HS, W, W, PERM_ACC, 0, W, G, W, W, PERM_ACC, W, w.scratch, gw[l], n_in, 0);
        }
        if (!rc) rc = bias_grad(s, P, G, W, 0, W, PERM_ACC, W, gb[l]);
    }
    const uint16_t *h8 = w.sv_h + (D - 1) * HS;
    // feature_linear
    if (!rc) rc = weight_grad(g_blas, s, P, h8, W, W, PERM_ACC, 0, W, w.g_feat, W, W, PERM_ACC, W, w.scratch, gw[D], W, 0);
    if (!rc) rc = bias_grad(s, P, w.g_feat, W, 0, W, PERM_ACC, W, gb[D]);
    // alpha_linear: G = column 3 of g_rawb
    if (!rc) rc = weight_grad(g_blas, s, P, h8, W, W, PERM_ACC, 0, W, w.g_rawb + 3, 4, 1, PERM_NAT, 1, w.scratch, gw[D + 1], W, 0);
    if (!rc) rc = bias_grad(s, P, w.g_rawb, 4, 3, 1, PERM_NAT, 1, gb[D + 1]);
    // views_linears.0: [feature | dirs]
    if (!rc) rc = weight_grad(g_blas, s, P, w.sv_feat, W, W, PERM_ACC, 0, W, w.g_hv, W / 2, W / 2, PERM_ACC, W / 2, w.scratch, gw[D + 2], W + icv, 0);
    if (!rc) rc = weight_grad(g_blas, s, P, w.sv_d, Dd, Dd, PERM_GEN, Ld, icv, w.g_hv, W / 2, W / 2, PERM_ACC, W / 2, w.scratch, gw[D + 2], W + icv, W);
    if (!rc) rc = bias_grad(s, P, w.g_hv, W / 2, 0, W / 2, PERM_ACC, W / 2, gb[D + 2]);
    // rgb_linear: G = columns 0..2 of g_rawb
    if (!rc) rc = weight_grad(g_blas, s, P, w.sv_hv, W / 2, W / 2, PERM_ACC, 0, W / 2, w.g_rawb, 4, 3, PERM_NAT, 3, w.scratch, gw[D + 3], W / 2, 0);
    if (!rc) rc = bias_grad(s, P, w.g_rawb, 4, 0, 3, PERM_NAT, 3, gb[D + 3]);
    return rc;
}

}  // namespace na
