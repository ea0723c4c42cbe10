// render.hip -- the per-ray stages of Renderer.render_rays around the field MLP:
// stratified depths, alpha compositing, hierarchical resampling, ray generation.
//
// All of these are HBM-/latency-bound per-ray scans and reductions; each ray is
// owned by one 64-lane wavefront so the cumprod / cumsum / sort never leave the
// wave (DPP shuffles + a few hundred bytes of LDS).
//
// Compiled with -ffp-contract=off: the reference's ATen ops round after every
// multiply and add, so no fused multiply-add may be formed here.
#include <hip/hip_runtime.h>
#include <cstring>
#include <math.h>

#include "kernels.h"
#include "launch_util.h"

namespace na {

constexpr int RAYS_PER_WG = 4;   // one wave per ray, 4 waves per workgroup

// Dynamic LDS of the per-ray kernels grows with the sample counts: reject what the CU cannot hold
// (EINVAL, before any launch) and raise the kernel's limit above the 64-KiB default when needed.
constexpr size_t LDS_PER_CU = 160 * 1024;
static int pow2_at_least(int n) {
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}
size_t composite_bwd_lds_bytes(int S) { return (size_t)RAYS_PER_WG * 3 * S * sizeof(float); }
size_t sample_pdf_lds_bytes(int n_bins) { return (size_t)RAYS_PER_WG * 2 * n_bins * sizeof(float); }
size_t resample_lds_bytes(int Nc, int Ni, bool with_composite) {
    return (size_t)RAYS_PER_WG * ((with_composite ? Nc : 0) + 2 * (size_t)(Nc - 1) + pow2_at_least(Nc) + pow2_at_least(Ni)) * sizeof(float);
}
static int reserve_lds(DynamicLdsOptIn &opt_in, const void *kernel, size_t bytes) {
    if (bytes > LDS_PER_CU) return NERF_AMD_EINVAL;
    if (bytes > 64 * 1024 && opt_in.ensure(kernel, LDS_PER_CU) != hipSuccess) return NERF_AMD_EHIP;
    return NERF_AMD_OK;
}

// Each ray's scratch lives in LDS that only its own wave touches, so ordering LDS
// traffic inside the wave is all that is needed: wait for this wave's LDS
// operations and stop the compiler from moving memory accesses across.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double wave_incl_prod(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_up(v, d);
        if (lane >= d) v *= o;
    }
    return v;
}
__device__ __forceinline__ double wave_incl_sum(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}
template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ---------------------------------------------------------------------------
// z_vals of the coarse pass: render_utils.py:105-129
// ---------------------------------------------------------------------------
__device__ __forceinline__ float z_of_t(float near, float far, float t, int lindisp) {
    if (!lindisp) return near * (1.0f - t) + far * t;
    return 1.0f / (1.0f / near * (1.0f - t) + 1.0f / far * t);
}

__global__ __launch_bounds__(256) void coarse_z_kernel(const float *rays, int ray_stride, const float *t_vals,
                                                       const float *t_rand, int64_t R, int Nc, int lindisp,
                                                       int perturb, float *z) {
    const int64_t total = R * Nc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Nc;
        const int s = (int)(i - r * Nc);
        const float near = rays[r * ray_stride + 6], far = rays[r * ray_stride + 7];
        float zc = z_of_t(near, far, t_vals[s], lindisp);
        if (perturb) {
            float upper = zc, lower = zc;
            if (s + 1 < Nc) upper = 0.5f * (z_of_t(near, far, t_vals[s + 1], lindisp) + zc);
            if (s > 0) lower = 0.5f * (zc + z_of_t(near, far, t_vals[s - 1], lindisp));
            zc = lower + (upper - lower) * t_rand[i];
        }
        z[i] = zc;
    }
}

int launch_coarse_z(const float *rays, int ray_stride, const float *t_vals, const float *t_rand,
                    int64_t R, int Nc, int lindisp, int perturb, float *z, hipStream_t s) {
    if (R <= 0) return NERF_AMD_OK;
    int64_t blocks = (R * Nc + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(coarse_z_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rays, ray_stride, t_vals, t_rand,
                       R, Nc, lindisp, perturb, z);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// ---------------------------------------------------------------------------
// raw2outputs: render_utils.py:241-290.  One wave per ray, 64 samples per step.
// The transmittance product is carried in fp64 like ATen's CPU cumprod
// (acc_type<float> = double) and rounded to fp32 per sample.
// ---------------------------------------------------------------------------
// raw2outputs for ray r by one wave; store_w(s, w) receives every sample's weight.
template <class StoreW>
__device__ __forceinline__ void composite_ray(const float *raw, int raw_ch, const float *z, const float *rays_d,
                                              int rays_d_stride, const float *noise, int64_t r, int S, int white_bkgd,
                                              float *rgb_map, float *disp_map, float *acc_map, float *depth_map, int lane,
                                              StoreW store_w) {
    const float *d = rays_d + r * rays_d_stride;
    const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float *zr = z + r * S;
    double carry = 1.0;                      // product of (1 - alpha + 1e-10) over all previous samples
    float sr = 0.f, sg = 0.f, sb = 0.f, sdepth = 0.f, sacc = 0.f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool in = s < S;
        float w = 0.f, zc = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
        double term = 1.0;
        float alpha = 0.f;
        if (in) {
            zc = zr[s];
            float dist = (s + 1 < S) ? zr[s + 1] - zc : 1e10f;
            dist = dist * dnorm;
            const float *q = raw + (r * S + s) * raw_ch;
            cr = 1.0f / (1.0f + expf(-q[0]));
            cg = 1.0f / (1.0f + expf(-q[1]));
            cb = 1.0f / (1.0f + expf(-q[2]));
            float sigma = q[3];
            if (noise) sigma = sigma + noise[r * S + s];
            sigma = fmaxf(sigma, 0.0f);
            alpha = 1.0f - expf(-sigma * dist);
            // One sample: the reference's `dists` comes out EMPTY (the 1e10 tail is expanded to dists[..., :1].shape = [R, 0],
            // render_utils.py:256-258), so do alpha and the weights; every sum over them is 0: background colour, acc 0, disp NaN.
            if (S == 1) alpha = 0.0f;
            term = (double)(1.0f - alpha + 1e-10f);
        }
        const double incl = wave_incl_prod(term, lane);
        double excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0;
        // ATen rounds every running product to fp32 on output; the running value itself stays fp64.
        const float T = (float)(carry * excl);
        carry = carry * __shfl(incl, 63);
        if (in) {
            w = alpha * T;
            store_w(s, w);
        }
        sr += w * cr; sg += w * cg; sb += w * cb;
        sdepth += w * zc; sacc += w;
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb);
    sdepth = wave_sum(sdepth); sacc = wave_sum(sacc);
    if (lane == 0) {
        if (white_bkgd) { const float bg = 1.0f - sacc; sr = sr + bg; sg = sg + bg; sb = sb + bg; }
        if (rgb_map) { rgb_map[3 * r] = sr; rgb_map[3 * r + 1] = sg; rgb_map[3 * r + 2] = sb; }
        if (depth_map) depth_map[r] = sdepth;
        if (acc_map) acc_map[r] = sacc;
        if (disp_map) {
            const float q = sdepth / sacc;               // 0/0 -> NaN, which torch.max propagates (:284)
            const float m = (q != q) ? q : (q > 1e-10f ? q : 1e-10f);
            disp_map[r] = 1.0f / m;
        }
    }
}

__global__ __launch_bounds__(64 * RAYS_PER_WG) void composite_kernel(
    const float *raw, int raw_ch, const float *z, const float *rays_d, int rays_d_stride, const float *noise,
    int64_t R, int S, int white_bkgd, float *rgb_map, float *disp_map, float *acc_map, float *weights,
    float *depth_map) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
    if (r >= R) return;
    composite_ray(raw, raw_ch, z, rays_d, rays_d_stride, noise, r, S, white_bkgd, rgb_map, disp_map, acc_map, depth_map, lane,
                  [&](int s, float w) { if (weights) weights[r * S + s] = w; });
}

int launch_composite(const float *raw, int raw_ch, const float *z, const float *rays_d, int rays_d_stride,
                     const float *noise, int64_t R, int S, int white_bkgd, float *rgb, float *disp, float *acc,
                     float *weights, float *depth, hipStream_t s) {
    if (R <= 0) return NERF_AMD_OK;
    if (S < 1 || raw_ch < 4) return NERF_AMD_EINVAL;
    const int64_t blocks = (R + RAYS_PER_WG - 1) / RAYS_PER_WG;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)blocks), dim3(64 * RAYS_PER_WG), 0, s, raw, raw_ch, z,
                       rays_d, rays_d_stride, noise, R, S, white_bkgd, rgb, disp, acc, weights, depth);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// ---------------------------------------------------------------------------
// Backward of raw2outputs (render_utils.py:241-290) with respect to raw: what
// torch.autograd produces for the reference's expression graph.  With
//   w_s = a_s T_s,  T_s = prod_{j<s}(1 - a_j + 1e-10),  a_s = 1 - exp(-relu(sigma_s + n_s) d_s),
//   rgb = sum w c + white (1 - sum w),  acc = sum w,  depth = sum w z,  disp = 1 / max(1e-10, depth / acc)
// and v_s = dL/dw_s (direct dependence only), u_s = v_s w_s:
//   dL/dsigma_s = d_s [sigma_s + n_s > 0] (1 - a_s) ( v_s T_s - (sum_{k>s} u_k) / (1 - a_s + 1e-10) )
//   dL/draw_rgb = g_rgb w_s c (1 - c)
// One wave per ray; a, T, w live in this wave's LDS slice between the two sweeps.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_suffix_excl_sum(double v, int lane) {
    double incl = v;                       // inclusive suffix sum: lanes lane..63
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_down(incl, d);
        if (lane + d < 64) incl += o;
    }
    return incl - v;
}

__global__ __launch_bounds__(64 * RAYS_PER_WG) void composite_bwd_kernel(
    const float *raw, int raw_ch, const float *z, const float *rays_d, int rays_d_stride, const float *noise,
    int64_t R, int S, int white_bkgd, const float *g_rgb, const float *g_disp, const float *g_acc,
    const float *g_depth, const float *g_weights, float *g_raw, float *g_rays_d) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_WG + wv;
    if (r >= R) return;
    float *sa = lds + wv * 3 * S, *sT = sa + S, *sw = sT + S;
    const float *d = rays_d + r * rays_d_stride;
    const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float *zr = z + r * S;
    // ---- sweep 1: alpha, transmittance, weights; acc and depth totals
    double carry = 1.0;
    float sacc = 0.f, sdepth = 0.f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool in = s < S;
        float alpha = 0.f, w = 0.f, zc = 0.f;
        double term = 1.0;
        if (in) {
            zc = zr[s];
            float dist = (s + 1 < S) ? zr[s + 1] - zc : 1e10f;
            dist = dist * dnorm;
            float sigma = raw[(r * S + s) * raw_ch + 3];
            if (noise) sigma = sigma + noise[r * S + s];
            sigma = fmaxf(sigma, 0.0f);
            alpha = 1.0f - expf(-sigma * dist);
            if (S == 1) alpha = 0.0f;                          // as in the forward: no weights, no gradient
            term = (double)(1.0f - alpha + 1e-10f);
        }
        const double incl = wave_incl_prod(term, lane);
        double excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0;
        const float T = (float)(carry * excl);
        carry = carry * __shfl(incl, 63);
        if (in) {
            w = alpha * T;
            sa[s] = alpha; sT[s] = T; sw[s] = w;
        }
        sacc += w; sdepth += w * zc;
    }
    sacc = wave_sum(sacc); sdepth = wave_sum(sdepth);
    wave_lds_sync();
    // ---- upstream gradients of the per-ray scalars
    const float gr = g_rgb ? g_rgb[3 * r] : 0.f, gg = g_rgb ? g_rgb[3 * r + 1] : 0.f, gb = g_rgb ? g_rgb[3 * r + 2] : 0.f;
    float gacc = g_acc ? g_acc[r] : 0.f, gdep = g_depth ? g_depth[r] : 0.f;
    if (g_disp) {
        const float q = sdepth / sacc;
        const float m = (q != q) ? q : (q > 1e-10f ? q : 1e-10f);
        const float gq = (q > 1e-10f || q != q) ? -g_disp[r] / (m * m) : 0.0f;   // d(1/max(1e-10, q))/dq
        gdep += gq / sacc;
        gacc += -gq * sdepth / (sacc * sacc);
    }
    if (white_bkgd) gacc -= gr + gg + gb;
    // ---- sweep 2, last chunk first: suffix sums of u = v w
    double tail = 0.0;                     // sum of u over all later chunks
    float gdn = 0.f;                       // dL/d|rays_d| (dists = dz * |d|, render_utils.py:259)
    for (int s0 = ((S - 1) / 64) * 64; s0 >= 0; s0 -= 64) {
        const int s = s0 + lane;
        const bool in = s < S;
        float v = 0.f, w = 0.f, a = 0.f, T = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, sig = 0.f, dist = 0.f;
        if (in) {
            const float *qv = raw + (r * S + s) * raw_ch;
            cr = 1.0f / (1.0f + expf(-qv[0]));
            cg = 1.0f / (1.0f + expf(-qv[1]));
            cb = 1.0f / (1.0f + expf(-qv[2]));
            sig = qv[3];
            if (noise) sig = sig + noise[r * S + s];
            const float zc = zr[s];
            dist = ((s + 1 < S) ? zr[s + 1] - zc : 1e10f) * dnorm;
            a = sa[s]; T = sT[s]; w = sw[s];
            v = gr * cr + gg * cg + gb * cb + gacc + gdep * zc;
            if (g_weights) v += g_weights[r * S + s];
        }
        const double u = (double)v * (double)w;
        const double later = wave_suffix_excl_sum(u, lane) + tail;
        tail += wave_sum(u);
        if (in) {
            float *o = g_raw + (r * S + s) * raw_ch;
            o[0] = gr * w * cr * (1.0f - cr);
            o[1] = gg * w * cg * (1.0f - cg);
            o[2] = gb * w * cb * (1.0f - cb);
            const float oma = 1.0f - a;
            const float core = (sig > 0.0f && S > 1) ? oma * (v * T - (float)later / (oma + 1e-10f)) : 0.0f;   // dL/dalpha * (1 - alpha)
            o[3] = dist * core;
            for (int c = 4; c < raw_ch; ++c) o[c] = 0.0f;
            gdn += core * sig * (dist / dnorm);                  // dL/ddist * dz
        }
    }
    if (g_rays_d) {
        gdn = wave_sum(gdn);
        if (lane == 0) {
            g_rays_d[3 * r] = gdn * d[0] / dnorm;
            g_rays_d[3 * r + 1] = gdn * d[1] / dnorm;
            g_rays_d[3 * r + 2] = gdn * d[2] / dnorm;
        }
    }
}

int launch_composite_bwd(const float *raw, int raw_ch, const float *z, const float *rays_d, int rays_d_stride,
                         const float *noise, int64_t R, int S, int white_bkgd, const float *g_rgb, const float *g_disp,
                         const float *g_acc, const float *g_depth, const float *g_weights, float *g_raw, float *g_rays_d,
                         hipStream_t s) {
    if (R <= 0) return NERF_AMD_OK;
    if (S < 1 || raw_ch < 4) return NERF_AMD_EINVAL;
    const int64_t blocks = (R + RAYS_PER_WG - 1) / RAYS_PER_WG;
    const size_t lds = composite_bwd_lds_bytes(S);
    static DynamicLdsOptIn opt_in;
    if (int rc = reserve_lds(opt_in, reinterpret_cast<const void *>(composite_bwd_kernel), lds)) return rc;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)blocks), dim3(64 * RAYS_PER_WG), lds, s, raw, raw_ch, z,
                       rays_d, rays_d_stride, noise, R, S, white_bkgd, g_rgb, g_disp, g_acc, g_depth, g_weights, g_raw,
                       g_rays_d);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// ---------------------------------------------------------------------------
// sample_pdf: utils.py:74-117.  cdf (fp64 running sum rounded to fp32 per bin,
// like ATen's CPU cumsum) and bins of one ray live in this wave's LDS slice.
// ---------------------------------------------------------------------------
// Builds cdf[0..nb) in LDS from w[0..nb-1) (weights already offset so w[i] is bin i).
template <class LoadW>
__device__ __forceinline__ void build_cdf(LoadW load_w, int nb, float *cdf, int lane) {
    double total = 0.0;
    for (int i0 = 0; i0 < nb - 1; i0 += 64) {
        const int i = i0 + lane;
        total += (i < nb - 1) ? (double)(load_w(i) + 1e-5f) : 0.0;
    }
    const float wsum = (float)wave_sum(total);
    double carry = 0.0;
    if (lane == 0) cdf[0] = 0.0f;
    for (int i0 = 0; i0 < nb - 1; i0 += 64) {
        const int i = i0 + lane;
        const float pdf = (i < nb - 1) ? (load_w(i) + 1e-5f) / wsum : 0.0f;
        const double incl = wave_incl_sum((double)pdf, lane);
        if (i < nb - 1) cdf[i + 1] = (float)(carry + incl);
        carry += __shfl(incl, 63);
    }
}

__device__ __forceinline__ float invert_cdf(const float *cdf, const float *bins, int nb, float u) {
    int lo = 0, hi = nb;                      // first index with cdf[idx] > u   (searchsorted right=True)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    const int below = lo - 1 > 0 ? lo - 1 : 0;
    const int above = lo < nb - 1 ? lo : nb - 1;
    const float c0 = cdf[below], c1 = cdf[above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;
    const float t = (u - c0) / denom;
    const float b0 = bins[below], b1 = bins[above];
    return b0 + t * (b1 - b0);
}

__global__ __launch_bounds__(64 * RAYS_PER_WG) void sample_pdf_kernel(const float *bins, const float *weights,
                                                                      const float *u, const float *t_lin, int64_t R,
                                                                      int nb, int N, float *samples) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t r = (int64_t)blockIdx.x * RAYS_PER_WG + wv;
    const bool live = r < R;
    if (!live) r = R - 1;
    float *cdf = lds + wv * 2 * nb, *bl = cdf + nb;
    const float *wr = weights + r * (nb - 1);
    build_cdf([&](int i) { return wr[i]; }, nb, cdf, lane);
    for (int i = lane; i < nb; i += 64) bl[i] = bins[r * nb + i];
    wave_lds_sync();
    if (!live) return;
    for (int i = lane; i < N; i += 64) {
        const float uu = u ? u[r * N + i] : t_lin[i];
        samples[r * N + i] = invert_cdf(cdf, bl, nb, uu);
    }
}

int launch_sample_pdf(const float *bins, const float *weights, const float *u, const float *t_lin,
                      int64_t R, int n_bins, int n_samples, float *samples, hipStream_t s) {
    if (R <= 0 || n_samples <= 0) return NERF_AMD_OK;
    if (n_bins < 2) return NERF_AMD_EINVAL;      // one bin edge, no weights: the reference's cdf is empty and its gather fails
    const int64_t blocks = (R + RAYS_PER_WG - 1) / RAYS_PER_WG;
    const size_t lds = sample_pdf_lds_bytes(n_bins);
    static DynamicLdsOptIn opt_in;
    if (int rc = reserve_lds(opt_in, reinterpret_cast<const void *>(sample_pdf_kernel), lds)) return rc;
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)blocks), dim3(64 * RAYS_PER_WG), lds, s, bins, weights, u,
                       t_lin, R, n_bins, n_samples, samples);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// ---------------------------------------------------------------------------
// The resampling step of render_rays (render_utils.py:140-148, :168) fused:
//   z_mid -> sample_pdf(z_mid, weights[1:-1]) -> z_std -> sort(cat[z, z_samples])
// The concatenation is two runs that are almost always sorted already (coarse depths are monotone; the
// samples are monotone in u, and u is sorted in the deterministic mode), so torch.sort's result is
// produced as a merge: each run is checked, sorted in place only if it has to be (bitonic network
// private to the wave), and every element finds its output position with one binary search in the
// other run.  Comparisons use the floats' bits mapped to an unsigned total order (NaN last, as
// torch.sort places it).
// Per-ray LDS: cdf[Nc-1] + bins[Nc-1] + run A[pow2 >= Nc] + run B[pow2 >= Ni].
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned order_key(float v) {
    const unsigned b = __builtin_bit_cast(unsigned, v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// true when x[0..n) is non-decreasing (wave-uniform result)
__device__ __forceinline__ bool wave_is_sorted(const float *x, int n, int lane) {
    bool ok = true;
    for (int i = lane; i + 1 < n; i += 64) ok = ok && order_key(x[i]) <= order_key(x[i + 1]);
    return __all(ok);
}

// ascending bitonic sort of x[0..npad), npad a power of two (padding = +inf), private to this wave.
// Every lane owns one compare-exchange per pass: pair t <-> elements (i, i + j), i = 2j*(t / j) + t % j.
__device__ __forceinline__ void wave_bitonic_sort(float *x, int npad, int lane) {
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (npad >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const float a = x[i], b = x[i + j];
                const bool up = (i & k) == 0;
                if ((order_key(a) > order_key(b)) == up) { x[i] = b; x[i + j] = a; }
            }
            wave_lds_sync();
        }
    }
}

// number of elements of the sorted run x[0..n) whose key is < k (STRICT) or <= k
template <bool STRICT>
__device__ __forceinline__ int count_below(const float *x, int n, unsigned k) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const unsigned km = order_key(x[mid]);
        if (STRICT ? km < k : km <= k) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The resampling of ray r by one wave (scratch: cdf[nb] + bins[nb] + run A[pad_c] + run B[pad_s] floats of
// LDS); load_w(i) returns weights[r][1 + i].  `live` is false for the padding waves of the last workgroup.
template <class LoadW>
__device__ __forceinline__ void resample_ray(const float *z_coarse, LoadW load_w, const float *u, const float *t_lin,
                                             int64_t r, bool live, int Nc, int Ni, int pad_c, int pad_s, float *scratch,
                                             float *z_fine, float *z_std, int lane) {
    const int nb = Nc - 1;
    float *cdf = scratch, *bl = cdf + nb, *za = bl + nb, *zs = za + pad_c;
    const float *zr = z_coarse + r * Nc;
    build_cdf(load_w, nb, cdf, lane);
    for (int i = lane; i < nb; i += 64) bl[i] = 0.5f * (zr[i + 1] + zr[i]);
    for (int i = lane; i < pad_c; i += 64) za[i] = i < Nc ? zr[i] : INFINITY;
    for (int i = Ni + lane; i < pad_s; i += 64) zs[i] = INFINITY;
    wave_lds_sync();
    double sum = 0.0;
    for (int i = lane; i < Ni; i += 64) {
        const float uu = u ? u[r * Ni + i] : t_lin[i];
        const float v = invert_cdf(cdf, bl, nb, uu);
        zs[i] = v;
        sum += (double)v;
    }
    // std(z_samples, unbiased=False): two-pass in fp64
    const double mean = wave_sum(sum) / (double)Ni;
    double ss = 0.0;
    for (int i = lane; i < Ni; i += 64) {
        const double dv = (double)zs[i] - mean;
        ss += dv * dv;
    }
    ss = wave_sum(ss);
    if (live && lane == 0 && z_std) z_std[r] = (float)sqrt(ss / (double)Ni);
    wave_lds_sync();
    if (!wave_is_sorted(zs, Ni, lane)) wave_bitonic_sort(zs, pad_s, lane);      // random u (perturb > 0)
    if (!wave_is_sorted(za, Nc, lane)) wave_bitonic_sort(za, pad_c, lane);      // an ulp of jitter rounding, if ever
    if (!live) return;
    // merge: ties keep the coarse value first (any consistent rule yields the same sorted values)
    float *out = z_fine + r * (Nc + Ni);
    for (int i = lane; i < Nc; i += 64) {
        const float v = za[i];
        out[i + count_below<true>(zs, Ni, order_key(v))] = v;
    }
    for (int j = lane; j < Ni; j += 64) {
        const float v = zs[j];
        out[j + count_below<false>(za, Nc, order_key(v))] = v;
    }
}

__global__ __launch_bounds__(64 * RAYS_PER_WG) void resample_kernel(const float *z_coarse, const float *weights,
                                                                    const float *u, const float *t_lin, int64_t R,
                                                                    int Nc, int Ni, int pad_c, int pad_s, float *z_fine,
                                                                    float *z_std) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t r = (int64_t)blockIdx.x * RAYS_PER_WG + wv;
    const bool live = r < R;
    if (!live) r = R - 1;
    const float *wr = weights + r * Nc + 1;                       // weights[..., 1:-1]
    resample_ray(z_coarse, [&](int i) { return wr[i]; }, u, t_lin, r, live, Nc, Ni, pad_c, pad_s,
                 lds + wv * (2 * (Nc - 1) + pad_c + pad_s), z_fine, z_std, lane);
}

// Between the two field kernels of a chunk, ONE launch does two independent jobs (the weights go from the
// compositing to the inverse-CDF sampling through LDS): the coarse compositing + resampling of chunk k (blocks [0, cr_blocks)) and the final
// compositing of chunk k-1 (the remaining blocks) -- one dependent launch less per chunk than running the
// final compositing on its own.
__global__ __launch_bounds__(64 * RAYS_PER_WG) void mid_stage_kernel(CompositeJob cj, ResampleJob rj, CompositeJob fj,
                                                                     int cr_blocks) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if ((int)blockIdx.x < cr_blocks) {
        int64_t r = (int64_t)blockIdx.x * RAYS_PER_WG + wv;
        const bool live = r < cj.R;
        if (!live) r = cj.R - 1;
        const int Nc = cj.S;
        float *wl = lds + wv * (Nc + 2 * (Nc - 1) + rj.pad_c + rj.pad_s);
        composite_ray(cj.raw, cj.raw_ch, cj.z, cj.rays_d, cj.rays_d_stride, cj.noise, r, Nc, cj.white_bkgd,
                      live ? cj.rgb : nullptr, live ? cj.disp : nullptr, live ? cj.acc : nullptr, nullptr, lane,
                      [&](int s, float w) { wl[s] = w; if (live && cj.weights) cj.weights[r * Nc + s] = w; });
        wave_lds_sync();
        resample_ray(cj.z, [&](int i) { return wl[1 + i]; }, rj.u, rj.t_lin, r, live, Nc, rj.Ni, rj.pad_c, rj.pad_s,
                     wl + Nc, rj.z_fine, rj.z_std, lane);
    } else {
        const int64_t r = (int64_t)((int)blockIdx.x - cr_blocks) * RAYS_PER_WG + wv;
        if (r >= fj.R) return;
        composite_ray(fj.raw, fj.raw_ch, fj.z, fj.rays_d, fj.rays_d_stride, fj.noise, r, fj.S, fj.white_bkgd, fj.rgb, fj.disp,
                      fj.acc, nullptr, lane, [&](int s, float w) { if (fj.weights) fj.weights[r * fj.S + s] = w; });
    }
}

int launch_mid_stage(const CompositeJob &cj, const ResampleJob &rj, const CompositeJob *fj, hipStream_t s) {
    if (cj.R <= 0) return NERF_AMD_EINVAL;
    const int Nc = cj.S, Ni = rj.Ni;
    if (cj.raw_ch < 4 || Nc < 3 || Ni < 1) return NERF_AMD_EINVAL;
    ResampleJob r2 = rj;
    r2.pad_c = pow2_at_least(Nc); r2.pad_s = pow2_at_least(Ni);
    CompositeJob f2;
    std::memset(&f2, 0, sizeof(f2));
    if (fj) f2 = *fj;
    const int64_t cr_blocks = (cj.R + RAYS_PER_WG - 1) / RAYS_PER_WG;
    const int64_t f_blocks = (f2.R + RAYS_PER_WG - 1) / RAYS_PER_WG;
    if (cr_blocks + f_blocks >= ((int64_t)1 << 31)) return NERF_AMD_EINVAL;
    const size_t lds = resample_lds_bytes(Nc, Ni, true);
    static DynamicLdsOptIn opt_in;
    if (int rc = reserve_lds(opt_in, reinterpret_cast<const void *>(mid_stage_kernel), lds)) return rc;
    hipLaunchKernelGGL(mid_stage_kernel, dim3((unsigned)(cr_blocks + f_blocks)), dim3(64 * RAYS_PER_WG), lds, s, cj, r2, f2,
                       (int)cr_blocks);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

int launch_resample(const float *z_coarse, const float *weights, const float *u, const float *t_lin,
                    int64_t R, int Nc, int Ni, float *z_fine, float *z_std, hipStream_t s) {
    if (R <= 0) return NERF_AMD_OK;
    if (Nc < 3 || Ni < 1) return NERF_AMD_EINVAL;
    const int pad_c = pow2_at_least(Nc), pad_s = pow2_at_least(Ni);
    const int64_t blocks = (R + RAYS_PER_WG - 1) / RAYS_PER_WG;
    const size_t lds = resample_lds_bytes(Nc, Ni, false);
    static DynamicLdsOptIn opt_in;
    if (int rc = reserve_lds(opt_in, reinterpret_cast<const void *>(resample_kernel), lds)) return rc;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(64 * RAYS_PER_WG), lds, s, z_coarse, weights, u,
                       t_lin, R, Nc, Ni, pad_c, pad_s, z_fine, z_std);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// ---------------------------------------------------------------------------
// get_rays + viewdirs + ndc_rays + batch assembly: utils.py:33-71,
// render_utils.py:200-226.  One thread per pixel.
// ---------------------------------------------------------------------------
// |v| of a 3-vector, the library's one definition (make_rays, assemble_rays): separately rounded squares summed as
// (x^2 + z^2) + y^2, correctly rounded sqrt (this file is built with -ffp-contract=off).  It is the order torch.norm(v, dim=-1)
// uses for 3 elements on the build this was written against (torch 2.10 / ROCm 7, tools/micro/norm_probe.py), so the fast
// rays= path equals the autograd-tracked torch expression bit for bit there; on another torch the two may differ by one
// ulp of the norm, which is why parity tests compare view directions with a tolerance (2e-7), not bit for bit.
__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf((x * x + z * z) + y * y); }

struct RayGen {
    float fx, fy, cx, cy;
    float c2w[12], c2ws[12];
    int has_static, use_viewdirs, ndc, H, W;
    float near, far;
    float sx, sy;          // -1/(W/(2 focal)), -1/(H/(2 focal)) evaluated in fp64 like the Python scalars
};

__device__ __forceinline__ void cam_ray(const float *c, float dx, float dy, float dz, float *o, float *d) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d[k] = dx * c[4 * k] + dy * c[4 * k + 1] + dz * c[4 * k + 2];
        o[k] = c[4 * k + 3];
    }
}

__global__ __launch_bounds__(256) void make_rays_kernel(RayGen g, int64_t pix0, int64_t n, float *out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int64_t pix = pix0 + idx;
    const float i = (float)(pix % g.W), j = (float)(pix / g.W);
    const float dx = (i - g.cx) / g.fx, dy = -(j - g.cy) / g.fy, dz = -1.0f;
    float o[3], d[3], vd[3];
    cam_ray(g.c2w, dx, dy, dz, o, d);
    vd[0] = d[0]; vd[1] = d[1]; vd[2] = d[2];
    if (g.has_static) cam_ray(g.c2ws, dx, dy, dz, o, d);
    const int ch = g.use_viewdirs ? 11 : 8;
    float *row = out + idx * ch;
    if (g.use_viewdirs) {
        const float nrm = norm3(vd[0], vd[1], vd[2]);      // one summation order for every path that normalises a view direction
        row[8] = vd[0] / nrm; row[9] = vd[1] / nrm; row[10] = vd[2] / nrm;
    }
    if (g.ndc) {
        // ndc_rays(H, W, focal = K[0][0], near = 1.)
        const float nearp = 1.0f;
        const float t = -(nearp + o[2]) / d[2];
        o[0] = o[0] + t * d[0]; o[1] = o[1] + t * d[1]; o[2] = o[2] + t * d[2];
        const float sx = g.sx, sy = g.sy;
        const float o0 = sx * o[0] / o[2], o1 = sy * o[1] / o[2], o2 = 1.0f + 2.0f * nearp / o[2];
        const float d0 = sx * (d[0] / d[2] - o[0] / o[2]);
        const float d1 = sy * (d[1] / d[2] - o[1] / o[2]);
        const float d2 = -2.0f * nearp / o[2];
        o[0] = o0; o[1] = o1; o[2] = o2; d[0] = d0; d[1] = d1; d[2] = d2;
    }
    row[0] = o[0]; row[1] = o[1]; row[2] = o[2];
    row[3] = d[0]; row[4] = d[1]; row[5] = d[2];
    row[6] = g.near; row[7] = g.far;
}

// ---------------------------------------------------------------------------
// utils.ndc_rays (utils.py:54-71) on explicit ray arrays; one thread per ray.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ndc_rays_kernel(float sx, float sy, float nearp, const float *ro, const float *rd,
                                                       int64_t n, float *oo, float *od) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[3] = {ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]};
    const float d[3] = {rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]};
    const float t = -(nearp + o[2]) / d[2];
    o[0] = o[0] + t * d[0]; o[1] = o[1] + t * d[1]; o[2] = o[2] + t * d[2];
    oo[3 * i] = sx * o[0] / o[2];
    oo[3 * i + 1] = sy * o[1] / o[2];
    oo[3 * i + 2] = 1.0f + 2.0f * nearp / o[2];
    od[3 * i] = sx * (d[0] / d[2] - o[0] / o[2]);
    od[3 * i + 1] = sy * (d[1] / d[2] - o[1] / o[2]);
    od[3 * i + 2] = -2.0f * nearp / o[2];
}

// Backward of the NDC warp above: gradients of (oo, od) -> gradients of (ro, rd), per ray, following the
// forward's own sequence of operations (what torch.autograd differentiates in utils.py:54-71).
__global__ __launch_bounds__(256) void ndc_rays_bwd_kernel(float sx, float sy, float nearp, const float *ro, const float *rd,
                                                           const float *g_oo, const float *g_od, int64_t n, float *g_ro,
                                                           float *g_rd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float o[3] = {ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]};
    const float d[3] = {rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]};
    const float t = -(nearp + o[2]) / d[2];
    const float p[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
    const float go[3] = {g_oo ? g_oo[3 * i] : 0.f, g_oo ? g_oo[3 * i + 1] : 0.f, g_oo ? g_oo[3 * i + 2] : 0.f};
    const float gd[3] = {g_od ? g_od[3 * i] : 0.f, g_od ? g_od[3 * i + 1] : 0.f, g_od ? g_od[3 * i + 2] : 0.f};
    const float ipz = 1.0f / p[2], idz = 1.0f / d[2];
    const float ax = sx * (go[0] - gd[0]), ay = sy * (go[1] - gd[1]);       // weights of px/pz and py/pz
    float gp[3];
    gp[0] = ax * ipz;
    gp[1] = ay * ipz;
    gp[2] = -(ax * p[0] + ay * p[1]) * ipz * ipz + 2.0f * nearp * ipz * ipz * (gd[2] - go[2]);
    float gdir[3];
    gdir[0] = sx * gd[0] * idz;
    gdir[1] = sy * gd[1] * idz;
    gdir[2] = -(sx * gd[0] * d[0] + sy * gd[1] * d[1]) * idz * idz;
    const float gt = gp[0] * d[0] + gp[1] * d[1] + gp[2] * d[2];            // p = o + t d
    float gor[3] = {gp[0], gp[1], gp[2]};
    gdir[0] += t * gp[0]; gdir[1] += t * gp[1]; gdir[2] += t * gp[2];
    gor[2] += -gt * idz;                                                     // t = -(near + oz) / dz
    gdir[2] += -gt * t * idz;
    if (g_ro) { g_ro[3 * i] = gor[0]; g_ro[3 * i + 1] = gor[1]; g_ro[3 * i + 2] = gor[2]; }
    if (g_rd) { g_rd[3 * i] = gdir[0]; g_rd[3 * i + 1] = gdir[1]; g_rd[3 * i + 2] = gdir[2]; }
}

int launch_ndc_rays_bwd(int H, int W, double focal, float near, const float *rays_o, const float *rays_d, const float *g_oo,
                        const float *g_od, int64_t n, float *g_ro, float *g_rd, hipStream_t s) {
    if (n <= 0) return NERF_AMD_OK;
    const float sx = (float)(-1.0 / ((double)W / (2.0 * focal)));
    const float sy = (float)(-1.0 / ((double)H / (2.0 * focal)));
    hipLaunchKernelGGL(ndc_rays_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sx, sy, near, rays_o, rays_d,
                       g_oo, g_od, n, g_ro, g_rd);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

int launch_ndc_rays(int H, int W, double focal, float near, const float *rays_o, const float *rays_d, int64_t n,
                    float *out_o, float *out_d, hipStream_t s) {
    if (n <= 0) return NERF_AMD_OK;
    const float sx = (float)(-1.0 / ((double)W / (2.0 * focal)));
    const float sy = (float)(-1.0 / ((double)H / (2.0 * focal)));
    hipLaunchKernelGGL(ndc_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sx, sy, near, rays_o, rays_d,
                       n, out_o, out_d);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// Backward of get_rays with respect to c2w (utils.py:33-42): rays_d = R dirs, rays_o = t, so
// dL/dR[k][m] = sum_pix g_d[pix][k] dirs[pix][m] and dL/dt[k] = sum_pix g_o[pix][k].
__global__ __launch_bounds__(256) void get_rays_bwd_kernel(RayGen g, int64_t pix0, int64_t n, const float *g_o,
                                                           const float *g_d, float *g_c2w) {
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = pix0 + idx;
        const float i = (float)(pix % g.W), j = (float)(pix / g.W);
        const float dir[3] = {(i - g.cx) / g.fx, -(j - g.cy) / g.fy, -1.0f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gd = g_d ? g_d[3 * idx + k] : 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[4 * k + m] += gd * dir[m];
            if (g_o) acc[4 * k + 3] += g_o[3 * idx + k];
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float v = wave_sum(acc[i]);
        if ((threadIdx.x & 63) == 0) atomicAdd(g_c2w + i, v);
    }
}

int launch_get_rays_bwd(int H, int W, const double *K4, int64_t pix0, int64_t n, const float *g_o, const float *g_d,
                        float *g_c2w, hipStream_t s) {
    if (hipMemsetAsync(g_c2w, 0, 12 * sizeof(float), s) != hipSuccess) return NERF_AMD_EHIP;
    if (n <= 0) return NERF_AMD_OK;
    RayGen g;
    g.fx = (float)K4[0]; g.fy = (float)K4[1]; g.cx = (float)K4[2]; g.cy = (float)K4[3];
    g.H = H; g.W = W;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(get_rays_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, pix0, n, g_o, g_d, g_c2w);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// utils.to8b (/root/reference/nerf_shared/utils.py:30): uint8(255 * clip(x, 0, 1)), truncating.
// HBM-bound: 4 B read + 1 B written per element; four elements per thread (one 16-byte load, one
// dword store).  NaN quantises to 0 (what numpy's float->uint8 cast yields on x86).
__device__ __forceinline__ unsigned quant8(float x) {
    const float c = fminf(fmaxf(x, 0.0f), 1.0f);        // fmaxf(NaN, 0) = 0
    return (unsigned)__fmul_rn(255.0f, c);              // fp32 product like numpy's float32 * 255, then truncation
}

__global__ __launch_bounds__(256) void to8b_kernel(const float *x, int64_t n, uint8_t *out) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        reinterpret_cast<unsigned *>(out)[i] = quant8(v.x) | (quant8(v.y) << 8) | (quant8(v.z) << 16) | (quant8(v.w) << 24);
    }
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = (uint8_t)quant8(x[t]);
}

int launch_to8b(const float *x, int64_t n, uint8_t *out, hipStream_t s) {
    if (n <= 0) return NERF_AMD_OK;
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(to8b_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, out);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// ---- utils.img2mse (utils.py:24): mean((x - y)^2) and its gradient, one launch each --------------------------------
constexpr int MSE_BLOCK = 1024, MSE_PER_BLOCK = 16384, MSE_MAX_BLOCKS = 256;

__device__ __forceinline__ float block_sum_1024(float v, float *part /* [16] LDS */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < MSE_BLOCK / 64; ++k) t += part[k];
    return t;                        // valid in thread 0
}

__global__ __launch_bounds__(MSE_BLOCK) void img2mse_kernel(const float *x, const float *y, int64_t n, float *out, float *partials) {
    __shared__ float part[MSE_BLOCK / 64];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * MSE_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * MSE_BLOCK + threadIdx.x; i < n; i += stride) {
        const float d = x[i] - y[i];
        acc += d * d;
    }
    const float t = block_sum_1024(acc, part);
    if (threadIdx.x == 0) {
        if (gridDim.x == 1) out[0] = t / (float)n;
        else partials[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(MSE_MAX_BLOCKS) void img2mse_finish_kernel(const float *partials, int n_parts, int64_t n, float *out) {
    __shared__ float part[MSE_MAX_BLOCKS];
    part[threadIdx.x] = (int)threadIdx.x < n_parts ? partials[threadIdx.x] : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < n_parts; ++k) t += part[k];        // fixed order: the result does not depend on scheduling
        out[0] = t / (float)n;
    }
}

__global__ __launch_bounds__(256) void img2mse_bwd_kernel(const float *x, const float *y, int64_t n, const float *g, float *gx, float *gy) {
    const float s = g[0] * (2.0f / (float)n);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = s * (x[i] - y[i]);
        if (gx) gx[i] = v;
        if (gy) gy[i] = -v;
    }
}

int launch_img2mse(const float *x, const float *y, int64_t n, float *out, float *partials, hipStream_t s) {
    int64_t blocks = (n + MSE_PER_BLOCK - 1) / MSE_PER_BLOCK;
    if (blocks < 1) blocks = 1;
    if (blocks > MSE_MAX_BLOCKS) blocks = MSE_MAX_BLOCKS;
    if (blocks > 1 && !partials) return NERF_AMD_EINVAL;
    hipLaunchKernelGGL(img2mse_kernel, dim3((unsigned)blocks), dim3(MSE_BLOCK), 0, s, x, y, n, out, partials);
    if (blocks > 1) hipLaunchKernelGGL(img2mse_finish_kernel, dim3(1), dim3(MSE_MAX_BLOCKS), 0, s, partials, (int)blocks, n, out);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

int launch_img2mse_bwd(const float *x, const float *y, int64_t n, const float *g, float *gx, float *gy, hipStream_t s) {
    if (n <= 0) return NERF_AMD_OK;
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(img2mse_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, n, g, gx, gy);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// ---- Renderer.render(rays=...) batch assembly (render_utils.py:205-222): [o | d | near | far | v / |v|] per ray ------
__global__ __launch_bounds__(256) void assemble_rays_kernel(const float *o, const float *d, const float *v, int64_t n, float near,
                                                            float far, float *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int ch = v ? 11 : 8;
    float *r = out + i * ch;
    r[0] = o[3 * i]; r[1] = o[3 * i + 1]; r[2] = o[3 * i + 2];
    r[3] = d[3 * i]; r[4] = d[3 * i + 1]; r[5] = d[3 * i + 2];
    r[6] = near; r[7] = far;
    if (v) {
        const float x = v[3 * i], y = v[3 * i + 1], z = v[3 * i + 2];
        const float nrm = norm3(x, y, z);
        r[8] = x / nrm; r[9] = y / nrm; r[10] = z / nrm;
    }
}

int launch_assemble_rays(const float *o, const float *d, const float *v, int64_t n, float near, float far, float *out, hipStream_t s) {
    if (n <= 0) return NERF_AMD_OK;
    hipLaunchKernelGGL(assemble_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, o, d, v, n, near, far, out);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

int launch_make_rays(int H, int W, const double *K4, const float *c2w, const float *c2w_static,
                     int64_t pix0, int64_t n, float near, float far, int use_viewdirs, int ndc,
                     float *rays_out, hipStream_t s) {
    if (n <= 0) return NERF_AMD_OK;
    RayGen g;
    g.fx = (float)K4[0]; g.fy = (float)K4[1]; g.cx = (float)K4[2]; g.cy = (float)K4[3];
    for (int i = 0; i < 12; ++i) { g.c2w[i] = c2w[i]; g.c2ws[i] = c2w_static ? c2w_static[i] : 0.0f; }
    g.has_static = c2w_static != nullptr;
    g.use_viewdirs = use_viewdirs; g.ndc = ndc; g.H = H; g.W = W; g.near = near; g.far = far;
    g.sx = (float)(-1.0 / ((double)W / (2.0 * K4[0])));
    g.sy = (float)(-1.0 / ((double)H / (2.0 * K4[0])));
    hipLaunchKernelGGL(make_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, pix0, n, rays_out);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

}  // namespace na
