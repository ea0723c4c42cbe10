// mlp_bf16.hip -- fused positional-encoding + 8x256 NeRF MLP for gfx950 (MI355X).
//
// Replaces NeRF.forward + NeRF.MLP (/root/reference/nerf_shared/nerf.py:96-134)
// together with the point construction pts = o + d*z of render_rays
// (/root/reference/nerf_shared/render_utils.py:131,148) for the canonical
// architecture D=8, W=256, skips=[4].
//
// Design:
//  * The network is evaluated transposed, H_out^T = W . H_in^T, with
//    v_mfma_f32_32x32x16_bf16.  One wave owns 32 points (MFMA columns) and keeps
//    the whole 256-wide activation of those points in registers: the fp32
//    accumulator tile of layer l, converted in place to bf16, *is* the B operand
//    of layer l+1 (no LDS round trip, no lane movement) because the weight
//    stream is pre-permuted to the accumulator's row order (program.h).
//  * Weights are the A operand.  A workgroup of 8 waves (256 points) streams the
//    packed 1-KiB fragments once from L2 into a 3 x 16 KiB LDS ring with
//    global_load_lds_dwordx4 (LDS-DMA, no VGPR staging); every wave then reads
//    each fragment with one conflict-free ds_read_b128 per MFMA.
//  * Positional encodings are generated in registers straight into B-operand
//    layout (sin features on lanes 0-31, cos features on lanes 32-63).
//  * bias add = accumulator initialisation from an LDS table; ReLU + bf16
//    conversion happen on the accumulator registers.
#include <hip/hip_runtime.h>
#include <utility>

#include "kernels.h"
#include "launch_util.h"
#include "pipeline.h"
#include "program.h"

namespace na {

// One 32-row output tile for the wave's NP column tiles: acc[i] = bias + sum over K1 k-steps
// of x1 and K2 of x2.  Activation fragment k of column tile i is x[k * NP + i].
template <int F0, int T, int K1, int K2, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tile(C &c, const bf16x8 *x1, const bf16x8 *x2, f32x16 (&acc)[C::NP]) {
    {
        const f32x4 *b = reinterpret_cast<const f32x4 *>(c.bias_half + T * 32);
        f32x4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
#pragma unroll
        for (int p = 0; p < C::NP; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[p][i] = b0[i]; acc[p][4 + i] = b1[i]; acc[p][8 + i] = b2[i]; acc[p][12 + i] = b3[i]; }
    }
    static_for<K1>([&](auto k_) {
        constexpr int k = k_, n = F0 + k;
        const bf16x8 w = take<n, NB, NFRAGS>(c);
        static_for<C::NP>([&](auto p_) { constexpr int p = p_;
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x1[k * C::NP + p], acc[p], 0, 0, 0); });
    });
    static_for<K2>([&](auto k_) {
        constexpr int k = k_, n = F0 + K1 + k;
        const bf16x8 w = take<n, NB, NFRAGS>(c);
        static_for<C::NP>([&](auto p_) { constexpr int p = p_;
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x2[k * C::NP + p], acc[p], 0, 0, 0); });
    });
}

template <bool RELU>
__device__ __forceinline__ void pack_tile(const f32x16 &acc, bf16x8 &lo, bf16x8 &hi) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float a = acc[r], b = acc[8 + r];
        if (RELU) { a = relu_bits(a); b = relu_bits(b); }
        lo[r] = (__bf16)a;
        hi[r] = (__bf16)b;
    }
}

// A full hidden layer: NT output tiles -> y[2*NT*NP] (next layer's B fragments).
template <int F0, int T0, int NT, int K1, int K2, bool RELU, int NB, int NFRAGS, class C>
__device__ __forceinline__ void layer(C &c, const bf16x8 *x1, const bf16x8 *x2, bf16x8 *y) {
    static_for<NT>([&](auto t_) {
        constexpr int t = t_;
        f32x16 acc[C::NP];
        tile<F0 + t * (K1 + K2), T0 + t, K1, K2, NB, NFRAGS>(c, x1, x2, acc);
        static_for<C::NP>([&](auto p_) { constexpr int p = p_;
            pack_tile<RELU>(acc[p], y[(2 * t) * C::NP + p], y[(2 * t + 1) * C::NP + p]); });
    });
}

// Positional encoding of (x0,x1,x2) into B-operand layout (program.h, FRAG_GEN):
// lane half 0 evaluates sin(2^f x), half 1 cos(2^f x) = sin(2^f x + pi/2).
// The argument is reduced exactly: t = x/(2 pi) is kept as an unevaluated fp32
// sum th + tl, doubling th and taking v_fract is exact, and v_sin_f32 takes
// revolutions.  Absolute error is ~1e-6, far below the bf16 quantum (4e-3).
template <int L, int K, int STRIDE>
__device__ __forceinline__ void encode(float x0, float x1, float x2, int h, bf16x8 *out) {
    constexpr float INV2PI_HI = 0.15915494f;                       // fl32(1/(2 pi))
    constexpr float INV2PI_LO = (float)(0.15915494309189535 - (double)INV2PI_HI);
    float x[3] = {x0, x1, x2};
    const float phase = h ? 0.25f : 0.0f;
    // The reduction runs on |x| (v_fract of a negative number rounds, and that ulp would be doubled with every
    // frequency); sin is odd, cos even, so the sin family gets x's sign bit back at the end.
    float ra[3], tl[3];
    unsigned sgn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ax = __builtin_fabsf(x[c]);
        sgn[c] = h ? 0u : (__builtin_bit_cast(unsigned, x[c]) & 0x80000000u);
        float th = ax * INV2PI_HI;
        tl[c] = __builtin_fmaf(ax, INV2PI_HI, -th) + ax * INV2PI_LO;
        ra[c] = __builtin_amdgcn_fractf(th);
    }
    float vals[8 * K];
#pragma unroll
    for (int e = 0; e < 8 * K; ++e) vals[e] = 0.0f;
#pragma unroll
    for (int f = 0; f < L; ++f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            vals[3 * f + c] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, __builtin_amdgcn_sinf(ra[c] + (tl[c] + phase))) ^ sgn[c]);
            ra[c] = __builtin_amdgcn_fractf(ra[c] * 2.0f);   // exact
            tl[c] *= 2.0f;                                   // exact
        }
    }
    vals[3 * L] = h ? x2 : x0;
    vals[3 * L + 1] = h ? 0.0f : x1;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) out[k * STRIDE][j] = (__bf16)vals[8 * k + j];
}

template <int LX, int LD, bool VD>
struct Layout {
    static constexpr int KE = gen_ksteps(LX);
    static constexpr int KD = VD ? gen_ksteps(LD) : 0;
    static constexpr int F_L0 = 0;
    static constexpr int F_L1 = F_L0 + 8 * KE;
    static constexpr int F_L5 = F_L1 + 4 * 128;
    static constexpr int F_L6 = F_L5 + 8 * (KE + 16);
    static constexpr int F_HEAD = F_L6 + 2 * 128;
    // viewdirs head
    static constexpr int F_FEAT = F_HEAD;
    static constexpr int F_ALPHA = F_FEAT + 128;
    static constexpr int F_VIEWS = F_ALPHA + 16;
    static constexpr int F_RGB = F_VIEWS + 4 * (16 + KD);
    static constexpr int F_END = VD ? F_RGB + 8 : F_HEAD + 16;
    static constexpr int T_HEAD = 64;
    static constexpr int N_TILES = VD ? 64 + 8 + 1 + 4 + 1 : 64 + 1;
};

template <int LX, int LD, bool VD, class C>
__global__ __launch_bounds__(C::WAVES * 64, C::WAVES_PER_SIMD) void mlp_bf16_kernel(MlpArgs a) {
    constexpr int NP = C::NP;
    constexpr int WG_THREADS = C::WAVES * 64, WG_POINTS = C::WAVES * 32 * NP;
    using Lay = Layout<LX, LD, VD>;
    constexpr int KE = Lay::KE, KD = Lay::KD, NF = Lay::F_END, NB = (NF + C::BF - 1) / C::BF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *bias_lds = reinterpret_cast<float *>(smem + C::RING_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int h = lane >> 5;
    C c;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.gstream = reinterpret_cast<const char *>(a.stream_bf16) + lane * 16;
    c.ring_lane = smem + lane * 16;
    c.ring_u32 = (uint32_t)(uintptr_t)smem;
    c.bias_half = bias_lds + h * 16;

    pipeline_prologue<NB>(c);

    for (int i = tid; i < Lay::N_TILES * 32; i += WG_THREADS) bias_lds[i] = a.bias_bf16[i];

    // ---- this lane's NP points (both lane halves hold the same points)
    bf16x8 E[KE * NP];
    bf16x8 Dv[(VD ? KD : 1) * NP];
    int64_t pidx[NP];
    bool valid[NP];
    static_for<NP>([&](auto i_) {
        constexpr int i = i_;
        const int64_t p = (int64_t)blockIdx.x * WG_POINTS + (c.wave * NP + i) * 32 + (lane & 31);
        pidx[i] = p;
        valid[i] = p < a.P;
        const int64_t pc = valid[i] ? p : a.P - 1;
        const int64_t ray = (int64_t)((uint32_t)pc / (uint32_t)a.S);   // P < 2^31 (checked at launch)
        float x0, x1, x2;
        if (a.pts) {
            x0 = a.pts[3 * pc + 0]; x1 = a.pts[3 * pc + 1]; x2 = a.pts[3 * pc + 2];
        } else {
            const float *r = a.rays + ray * a.ray_stride;
            const float z = a.z_vals[pc];
            x0 = mul_then_add(r[3], z, r[0]);
            x1 = mul_then_add(r[4], z, r[1]);
            x2 = mul_then_add(r[5], z, r[2]);
        }
        if constexpr (C::ABL & 4) {
            static_for<KE>([&](auto k_) { constexpr int k = k_; for (int j = 0; j < 8; ++j) E[k * NP + i][j] = (__bf16)(x0 + j); });
            if constexpr (VD) static_for<KD>([&](auto k_) { constexpr int k = k_; for (int j = 0; j < 8; ++j) Dv[k * NP + i][j] = (__bf16)(x1 + j); });
        } else {
            encode<LX, KE, NP>(x0, x1, x2, h, E + i);
            if constexpr (VD) {
                const float *d = a.viewdirs + ray * a.vd_stride;
                encode<LD, KD, NP>(d[0], d[1], d[2], h, Dv + i);
            }
        }
    });

    if constexpr (C::PHASE > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // bias table stores, before the barrier publishes them
        block_sync<-1, NB>(c);                                 // publishes block 0
        static_for<C::LA>([&](auto i_) { constexpr int i = i_; c.q[i] = ring_frag<i>(c); });
    }
    bf16x8 A[16 * NP], B[16 * NP];
    layer<Lay::F_L0, 0, 8, KE, 0, true, NB, NF>(c, E, E, A);
    layer<Lay::F_L1 + 0 * 128, 8, 8, 16, 0, true, NB, NF>(c, A, A, B);
    layer<Lay::F_L1 + 1 * 128, 16, 8, 16, 0, true, NB, NF>(c, B, B, A);
    layer<Lay::F_L1 + 2 * 128, 24, 8, 16, 0, true, NB, NF>(c, A, A, B);
    layer<Lay::F_L1 + 3 * 128, 32, 8, 16, 0, true, NB, NF>(c, B, B, A);
    layer<Lay::F_L5, 40, 8, KE, 16, true, NB, NF>(c, E, A, B);          // skip: [input_pts | h]
    layer<Lay::F_L6, 48, 8, 16, 0, true, NB, NF>(c, B, B, A);
    layer<Lay::F_L6 + 128, 56, 8, 16, 0, true, NB, NF>(c, A, A, B);     // h7 in B

    if constexpr (VD) {
        layer<Lay::F_FEAT, 64, 8, 16, 0, false, NB, NF>(c, B, B, A);                       // feature (no activation)
        f32x16 alpha[NP], rgb[NP];
        tile<Lay::F_ALPHA, 72, 16, 0, NB, NF>(c, B, B, alpha);                             // row 0 = sigma
        layer<Lay::F_VIEWS, 73, 4, 16, KD, true, NB, NF>(c, A, Dv, B);                     // views_linears.0
        tile<Lay::F_RGB, 77, 8, 0, NB, NF>(c, B, B, rgb);                                  // rows 0..2
        static_for<NP>([&](auto i_) {
            constexpr int i = i_;
            if (valid[i] && h == 0) {
                f32x4 o = {rgb[i][0], rgb[i][1], rgb[i][2], alpha[i][0]};
                *reinterpret_cast<f32x4 *>(a.out + 4 * pidx[i]) = o;
            }
        });
    } else {
        f32x16 o[NP];
        tile<Lay::F_HEAD, 64, 16, 0, NB, NF>(c, B, B, o);
        static_for<NP>([&](auto i_) {
            constexpr int i = i_;
            if (valid[i]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = acc_row(r, h);
                    if (row < a.out_ch) a.out[(int64_t)a.out_ch * pidx[i] + row] = o[i][r];
                }
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the workgroup
}

// tuning knob (nerf_amd_set_tuning key 0): 0 = 16x16x32 kernel, 100+ = this file's shapes (launch_one).  Atomic: the
// launchers of concurrent host threads read it while nerf_amd_set_tuning may write it (nerf_amd.h threading contract).
std::atomic<int> g_variant{0};

template <int LX, int LD, bool VD, class C>
static int launch_wg(const MlpArgs &a, int n_frags_used, int n_tiles, hipStream_t s) {
    constexpr int WG_THREADS = C::WAVES * 64, WG_POINTS = C::WAVES * 32 * C::NP;
    using Lay = Layout<LX, LD, VD>;
    if (n_frags_used != Lay::F_END || n_tiles != Lay::N_TILES) return NERF_AMD_EINVAL;
    const size_t lds = C::RING_BYTES + (size_t)Lay::N_TILES * 32 * sizeof(float);
    static DynamicLdsOptIn opt_in;
    if (opt_in.ensure(reinterpret_cast<const void *>(mlp_bf16_kernel<LX, LD, VD, C>), lds) != hipSuccess) return NERF_AMD_EHIP;
    const int64_t groups = (a.P + WG_POINTS - 1) / WG_POINTS;
    if (groups <= 0) return NERF_AMD_OK;
    if (a.P >= (int64_t)1 << 31) return NERF_AMD_EINVAL;
    hipLaunchKernelGGL((mlp_bf16_kernel<LX, LD, VD, C>), dim3((unsigned)groups), dim3(WG_THREADS), lds, s, a);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

using CfgDefault = Ctx<8, 16, 4, 8, 2>;   // 64-KiB ring, mid-block sync, 2-deep read-ahead (fastest of tools/mlp_ab.py)

#ifdef NERF_AMD_EXPERIMENTS
#include "../../tools/experiments/field_variants_gen1.inc"
#endif

template <int LX, int LD, bool VD>
static int launch_one(const MlpArgs &a, int n_frags_used, int n_tiles, hipStream_t s) {
#ifdef NERF_AMD_EXPERIMENTS   // A/B builds only: the variant table lives in tools/experiments/field_variants_gen1.inc
    {
        int rc_x = NERF_AMD_EUNSUPPORTED;
        if (experiment_launch_gen1<LX, LD, VD>(a, n_frags_used, n_tiles, s, &rc_x)) return rc_x;
    }
#endif
    if constexpr (LX == 10 && LD == 4 && VD)
        if (g_variant == 101) return launch_wg<LX, LD, VD, Ctx<8, 16, 3, 0>>(a, n_frags_used, n_tiles, s);   // round-1 first shape (A/B reference)
    return launch_wg<LX, LD, VD, CfgDefault>(a, n_frags_used, n_tiles, s);
}

bool mlp_bf16_supported(int multires, int multires_views, int use_viewdirs) {
    if (use_viewdirs) return (multires == 10 && multires_views == 4) || (multires == 15 && multires_views == 6);
    return multires == 10 || multires == 15;
}

int launch_mlp_bf16(const MlpArgs &a, int multires, int multires_views, int use_viewdirs,
                    int n_frags_used, int n_tiles, hipStream_t s) {
    if (use_viewdirs) {
        if (multires == 10 && multires_views == 4) return launch_one<10, 4, true>(a, n_frags_used, n_tiles, s);
        if (multires == 15 && multires_views == 6) return launch_one<15, 6, true>(a, n_frags_used, n_tiles, s);
    } else {
        if (multires == 10) return launch_one<10, 0, false>(a, n_frags_used, n_tiles, s);
        if (multires == 15) return launch_one<15, 0, false>(a, n_frags_used, n_tiles, s);
    }
    return NERF_AMD_EUNSUPPORTED;
}

}  // namespace na
