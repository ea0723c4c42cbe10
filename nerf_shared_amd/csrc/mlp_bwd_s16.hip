// mlp_bwd_s16.hip -- backward of the fused 8x256 NeRF MLP with respect to every
// layer's pre-activation (the "dX chain"), for the view-branch models (multires 10/4 and 15/6) and the
// output_linear models (use_viewdirs=False, nerf.py:91-94,131-132; multires 10 or 15, output_ch <= 16).
//
// What torch.autograd computes for NeRF.MLP (/root/reference/nerf_shared/nerf.py:110-134)
// given dL/draw: walking the layers in reverse,
//     g_pre(l) = relu'(h_l) * ( W_{l+1}^T g_pre(l+1) )
// is the forward kernel's computation with transposed weights: the gradient tile of one
// layer, converted to bf16 in place, is the B operand of the next (earlier) layer, so the
// whole chain stays in registers exactly like the forward activations (mlp_bf16_s16.hip).
// ReLU masks come from the bit rows the training forward saved (one bit per activation, 1/16 of
// re-reading the bf16 activations); every g_pre(l) is
// written to HBM (slot-major bf16 rows, same layout as the saved activations) for the
// weight-gradient GEMMs  dW_l = g_pre(l)^T h_{l-1}  that follow (backward.hip).
// Ray gradients: the gradient with respect to the encodings (three more transposed products whose
// rows are encoding slots, FRAG_TE16) is pushed through d/dx sin(2^f x) = 2^f cos(2^f x) in
// registers and summed per point (pts mode) or per ray (rays mode: dL/do += g, dL/dd += z g).
#include <hip/hip_runtime.h>
#include <utility>

#include "kernels.h"
#include "launch_util.h"
#include "pipeline.h"
#include "program.h"

#ifdef NERF_AMD_EXPERIMENTS            // scratch builds only (tools/experiments/README.md); the shipping build uses 4 ring slots
#include "../../tools/experiments/save_variants.inc"
#endif
#ifndef NA_EXPERIMENT_BWD_NS
#define NA_EXPERIMENT_BWD_NS 4
#endif

namespace na {

#define MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// Pair of 16-row tiles of W^T over K1 k-steps of x1 and K2 of x2, no bias.
template <int F0, int K1, int K2, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tpair(C &c, const bf16x8 *x1, const bf16x8 *x2, f32x4 (&acc)[2][2]) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    acc[0][0] = zero; acc[0][1] = zero; acc[1][0] = zero; acc[1][1] = zero;
    static_for<K1>([&](auto k_) {
        constexpr int k = k_, n = F0 + 2 * k;
        const bf16x8 w0 = take<n, NB, NFRAGS>(c);
        const bf16x8 w1 = take<n + 1, NB, NFRAGS>(c);
        acc[0][0] = MFMA16(w0, x1[2 * k], acc[0][0]);
        acc[0][1] = MFMA16(w0, x1[2 * k + 1], acc[0][1]);
        acc[1][0] = MFMA16(w1, x1[2 * k], acc[1][0]);
        acc[1][1] = MFMA16(w1, x1[2 * k + 1], acc[1][1]);
    });
    static_for<K2>([&](auto k_) {
        constexpr int k = k_, n = F0 + 2 * K1 + 2 * k;
        const bf16x8 w0 = take<n, NB, NFRAGS>(c);
        const bf16x8 w1 = take<n + 1, NB, NFRAGS>(c);
        acc[0][0] = MFMA16(w0, x2[2 * k], acc[0][0]);
        acc[0][1] = MFMA16(w0, x2[2 * k + 1], acc[0][1]);
        acc[1][0] = MFMA16(w1, x2[2 * k], acc[1][0]);
        acc[1][1] = MFMA16(w1, x2[2 * k + 1], acc[1][1]);
    });
}

// bf16 gradient fragment from a tile pair, zeroed where the saved activation is zero (ReLU').
// The mask bits of this fragment sit at BIT0 + i (element 2i) and 16 + BIT0 + i (element 2i + 1) of
// `bits` (mlp_bf16_s16.hip save_bits); each bit is sign-extended to a half-word.
template <bool MASK, int BIT0>
__device__ __forceinline__ bf16x8 pack_grad(const f32x4 &even, const f32x4 &odd, unsigned bits) {
    bf16x8 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) { y[r] = (__bf16)even[r]; y[4 + r] = (__bf16)odd[r]; }
    if (MASK) {
        u32x4 v = __builtin_bit_cast(u32x4, y);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)bits, BIT0 + i, 1);
            const unsigned hi = (unsigned)__builtin_amdgcn_sbfe((int)bits, 16 + BIT0 + i, 1);
            v[i] &= __builtin_amdgcn_perm(hi, lo, 0x05040100u);       // {hi.half, lo.half}
        }
        y = __builtin_bit_cast(bf16x8, v);
    }
    return y;
}

// Mask bytes of one layer for this lane's two points (NPAIR bytes each): loaded one layer ahead of
// their use so the wait the compiler places before the first use never reaches back to recent DMA.
template <int NPAIR>
struct MaskBits {
    unsigned w[2][NPAIR / 4];
};
template <int NPAIR>
__device__ __forceinline__ MaskBits<NPAIR> load_bits(const uint8_t *base, const int64_t (&pidx)[2], int q) {
    static_assert(NPAIR == 4 || NPAIR == 8, "one or two dwords of mask bytes");
    MaskBits<NPAIR> m;
    static_for<2>([&](auto cc_) {
        constexpr int cc = cc_;
        const uint8_t *src = base + pidx[cc] * (4 * NPAIR) + q * NPAIR;
        if constexpr (NPAIR == 8) {
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const u32x2 v = *reinterpret_cast<const u32x2 *>(src);
            m.w[cc][0] = v[0]; m.w[cc][1] = v[1];
        } else {
            m.w[cc][0] = *reinterpret_cast<const unsigned *>(src);
        }
    });
    return m;
}

// One transposed layer: NPAIR pairs of input-feature tiles -> g[2 * NPAIR]; g is masked with the
// layer's ReLU bits and stored to `dst` (ROW elements per point, rows of padding points included:
// the store count per pair is a compile-time constant, see BwdLedger).
template <int F0, int NPAIR, int K1, int K2, bool MASK, int ROW, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tlayer(C &c, const bf16x8 *x1, const bf16x8 *x2, bf16x8 *g, const MaskBits<NPAIR> &mask,
                                       uint16_t *dst, const int64_t (&pidx)[2], int q) {
    static_for<NPAIR>([&](auto p_) {
        constexpr int p = p_;
        f32x4 acc[2][2];
        tpair<F0 + p * 2 * (K1 + K2), K1, K2, NB, NFRAGS>(c, x1, x2, acc);
        g[2 * p] = pack_grad<MASK, 4 * (p % 4)>(acc[0][0], acc[1][0], mask.w[0][p / 4]);
        g[2 * p + 1] = pack_grad<MASK, 4 * (p % 4)>(acc[0][1], acc[1][1], mask.w[1][p / 4]);
        static_for<2>([&](auto cc_) {
            constexpr int cc = cc_;
            *reinterpret_cast<bf16x8 *>(dst + pidx[cc] * ROW + p * 32 + q * 8) = g[2 * p + cc];
        });
    });
}

// Gradient of a generated encoding (encode16 of mlp_bf16_s16.hip) with respect to its 3 inputs:
// g[i] is dL/d(slot i) of this lane's slots (same order as encode16's vals[]).
template <int L, int K>
__device__ __forceinline__ void encode16_bwd(float x0, float x1, float x2, int h, int b, const float *g, float (&gx)[3]) {
    constexpr float INV2PI_HI = 0.15915494f;
    constexpr float INV2PI_LO = (float)(0.15915494309189535 - (double)INV2PI_HI);
    constexpr int NSTEP = (L + 1) / 2, CAP = 8 * K;
    constexpr int N_EVEN = gen16_ntrig(L, 0), N_ODD = gen16_ntrig(L, 1);
    const float x[3] = {x0, x1, x2};
    const float phase = (h ? 0.25f : 0.0f) + 0.25f;          // cos(u) = sin(u + 1/4 turn): derivative of the forward value
    const float s0 = b ? 2.0f : 1.0f;
    const int n_mine = b ? N_ODD : N_EVEN;
    // reduction on |x| as in the forward (encode16): d/dx sin(2^f x) = 2^f cos(.) is even in x, d/dx cos(2^f x)
    // = -2^f sin(.) is odd, so here it is the cos family (h == 1) that takes x's sign bit
    float ra[3], tl[3];
    unsigned sgn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ax = __builtin_fabsf(x[c]);
        sgn[c] = h ? (__builtin_bit_cast(unsigned, x[c]) & 0x80000000u) : 0u;
        const float th = ax * INV2PI_HI;
        tl[c] = (__builtin_fmaf(ax, INV2PI_HI, -th) + ax * INV2PI_LO) * s0;
        ra[c] = __builtin_amdgcn_fractf(th * s0);
    }
    float scale = s0;                                         // 2^f of the current step
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (3 * s + c < CAP) {
                const float dv = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, __builtin_amdgcn_sinf(ra[c] + (tl[c] + phase))) ^ sgn[c]) * scale;
                gx[c] += (3 * s + c < n_mine) ? g[3 * s + c] * dv : 0.0f;
            }
            ra[c] = __builtin_amdgcn_fractf(ra[c] * 4.0f);
            tl[c] *= 4.0f;
        }
        scale *= 4.0f;
    }
    // raw-coordinate slots: derivative 1
#pragma unroll
    for (int i = (N_ODD < N_EVEN ? N_ODD : N_EVEN); i < CAP; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool even_hit = i >= N_EVEN && (h ? gen16_misc(L, 1, 0, i - N_EVEN) : gen16_misc(L, 0, 0, i - N_EVEN)) == c;
            const bool odd_hit = i >= N_ODD && (h ? gen16_misc(L, 1, 1, i - N_ODD) : gen16_misc(L, 0, 1, i - N_ODD)) == c;
            gx[c] += (b ? odd_hit : even_hit) ? g[i] : 0.0f;
        }
    }
}

// Transposed products whose rows are the slots of a generated encoding (FRAG_TE16): NPAIR pairs,
// K1 k-steps of x1 each; g[cc][8*pair + 4*u + r] receives this lane's slot gradients.
template <int F0, int NPAIR, int K1, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tenc(C &c, const bf16x8 *x1, float (&g)[2][8 * NPAIR]) {
    static_for<NPAIR>([&](auto p_) {
        constexpr int p = p_;
        f32x4 acc[2][2];
        tpair<F0 + p * 2 * K1, K1, 0, NB, NFRAGS>(c, x1, x1, acc);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 4; ++r) g[cc][8 * p + 4 * u + r] = acc[u][cc][r];
    });
}

// Fragment offsets of the backward stream (program.cpp frags_bwd) for encodings of KE / KD k-steps.
// Without a view branch (VD = false) the chain starts at g_h8 = relu'(h8) * (W_output^T dL/draw): 8 pairs x 1 k-step.
template <int KE, int KD, bool VD = true>
struct LayoutB {
    static constexpr int F_HV = 0;                       // 4 pairs x 1 k-step
    static constexpr int F_FEAT = F_HV + 8;              // 8 pairs x 4 k-steps
    static constexpr int F_DIRS = F_FEAT + 64;           // view-direction encoding slots: KD pairs x 4 k-steps
    static constexpr int F_H8 = VD ? F_DIRS + 8 * KD : 0;   // 8 pairs x (8 + 1) k-steps (VD) / x 1 k-step
    static constexpr int H8_FRAGS_PER_PAIR = VD ? 18 : 2;
    static constexpr int F_L7 = F_H8 + 8 * H8_FRAGS_PER_PAIR;   // pts_linears.7, .6, .5: 8 pairs x 8 k-steps each
    static constexpr int F_E5 = F_L7 + 3 * 128;          // xyz encoding slots through pts_linears.5: KE pairs x 8 k-steps
    static constexpr int F_L4 = F_E5 + 16 * KE;          // pts_linears.4 .. .1
    static constexpr int F_E0 = F_L4 + 4 * 128;          // xyz encoding slots through pts_linears.0
    static constexpr int F_END = F_E0 + 16 * KE;
};

// Row stores issued before fragment n (pipeline.h LEDGER): two per finished tile pair of a tlayer
// (the encoding products store nothing).
template <int KE, int KD, bool VD = true>
struct BwdLedger {
    using L = LayoutB<KE, KD, VD>;
    static constexpr int pairs_done(int n, int f0, int frags_per_pair, int n_pairs) {
        const int d = n <= f0 ? 0 : (n - f0) / frags_per_pair;
        return d > n_pairs ? n_pairs : d;
    }
    static constexpr int stores_before(int n) {
        return 2 * ((VD ? pairs_done(n, L::F_HV, 2, 4) + pairs_done(n, L::F_FEAT, 8, 8) : 0) +
                    pairs_done(n, L::F_H8, L::H8_FRAGS_PER_PAIR, 8) +
                    pairs_done(n, L::F_L7, 16, 24) + pairs_done(n, L::F_L4, 16, 32));
    }
};

// RAYG: someone wants dL/dpts / dL/drays / dL/dviewdirs.  A training step does not (render_utils.py:145 cuts the only path
// from the loss to the rays' depths, and the rays are data): then the three encoding products and the encodings' derivatives
// -- 6-9 % of the chain's MFMAs, all of its trigonometry, and the registers that made the multires-15 instantiation spill
// 120 VGPRs -- are skipped; their fragments pass through the ring unread (pipeline.h skip_frags).
template <int LX, int LD, bool VD, class C, bool RAYG>
__global__ __launch_bounds__(C::WAVES * 64, 2) void mlp_bwd_s16_kernel(MlpArgs a) {
    constexpr int WG_POINTS = C::WAVES * 32;
    constexpr int KE = gen16_ksteps(LX), KD = VD ? gen16_ksteps(LD) : 1;
    using LayoutB = LayoutB<KE, KD, VD>;
    constexpr int NF = LayoutB::F_END, NB = (NF + C::BF - 1) / C::BF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = lane >> 4;
    C c;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.gstream = reinterpret_cast<const char *>(a.stream_bwd) + lane * 16;
    c.ring_lane = smem + lane * 16;
    c.ring_u32 = (uint32_t)(uintptr_t)smem;
    c.bias_half = nullptr;

    pipeline_prologue<NB>(c);

    int64_t pidx[2], rayi[2];
    bool valid[2];
    bf16x8 Grgb[2], Gsig[2];
    float xp[2][3], dv[2][3], zp[2];
    static_for<2>([&](auto cc_) {
        constexpr int cc = cc_;
        const int64_t p = (int64_t)blockIdx.x * WG_POINTS + c.wave * 32 + cc * 16 + (lane & 15);
        pidx[cc] = p;
        valid[cc] = p < a.P;
        if constexpr (RAYG) {
            const int64_t pc = valid[cc] ? p : a.P - 1;
            const int64_t ray = (int64_t)((uint32_t)pc / (uint32_t)a.S);
            rayi[cc] = ray;
            zp[cc] = 0.f;
            if (a.pts) {
                xp[cc][0] = a.pts[3 * pc]; xp[cc][1] = a.pts[3 * pc + 1]; xp[cc][2] = a.pts[3 * pc + 2];
            } else {
                const float *r = a.rays + ray * a.ray_stride;
                zp[cc] = a.z_vals[pc];
                xp[cc][0] = mul_then_add(r[3], zp[cc], r[0]);
                xp[cc][1] = mul_then_add(r[4], zp[cc], r[1]);
                xp[cc][2] = mul_then_add(r[5], zp[cc], r[2]);
            }
            if constexpr (VD) {
                const float *d = a.viewdirs + ray * a.vd_stride;
                dv[cc][0] = d[0]; dv[cc][1] = d[1]; dv[cc][2] = d[2];
            }
        }
        if constexpr (!VD) {
            // dL/draw [P, out_ch] as the FRAG_TG16 operand: k slot (q, j) = output_linear row 8 q + j (q < 2); the bf16
            // values also go to g_rawb as rows of 16 columns for the head's weight-gradient product (backward.hip)
            bf16x8 o = {};
            if (q < 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = 8 * q + j;
                    const float gv = (valid[cc] && col < a.out_ch) ? a.g_raw[(int64_t)a.out_ch * p + col] : 0.0f;
                    o[j] = (__bf16)gv;
                }
                *reinterpret_cast<bf16x8 *>(a.g_rawb + 16 * p + 8 * q) = o;      // rows exist for the padding points
            }
            Gsig[cc] = o; Grgb[cc] = o;
            return;
        }
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (valid[cc]) g = *reinterpret_cast<const f32x4 *>(a.g_raw + 4 * p);
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        const bf16x4 gb = {(__bf16)g[0], (__bf16)g[1], (__bf16)g[2], (__bf16)g[3]};
        bf16x8 r = {}, s = {};
        if (q == 0) {
            r[0] = gb[0]; r[1] = gb[1]; r[2] = gb[2];                       // k slot (q=0, j) = rgb_linear output j
            s[0] = gb[3];                                                   // k slot (0, 0)   = alpha_linear output
        }
        Grgb[cc] = r; Gsig[cc] = s;
        if (valid[cc] && q == 0) *reinterpret_cast<bf16x4 *>(a.g_rawb + 4 * p) = gb;
        // the same values as the operand of the head weight-gradient products: transposed inside the wave's 32-point chunk,
        // lane quarter q writes column q (zeros for the padding points, so their saved rows contribute nothing)
        {
            const int pl = cc * 16 + (lane & 15);                          // this point inside its chunk (a wave's 32 points)
            const __bf16 v = q == 0 ? gb[0] : q == 1 ? gb[1] : q == 2 ? gb[2] : gb[3];
            reinterpret_cast<__bf16 *>(a.g_rawt)[(p >> 5) * 128 + ((pl >> 3) * 4 + q) * 8 + (pl & 7)] = v;
        }
    });

    if constexpr (C::PHASE > 0) {
        block_sync<-1, NB>(c);
        static_for<C::LA>([&](auto i_) { constexpr int i = i_; c.q[i] = ring_frag<i>(c); });
    }
    const int64_t HS = pad_points(a.P) * 256, BS = pad_points(a.P) * 32;
    const int hh = q >> 1, bb = q & 1;
    bf16x8 A[16], B[16];
    float gx[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, gd[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const MaskBits<8> none = {};
    MaskBits<8> m_cur = load_bits<8>(a.sv_bits + 7 * BS, pidx, q), m_next;
    if constexpr (VD) {
        // g_hv = relu'(hv) * (W_rgb^T g_rgb)
        const MaskBits<4> m_hv = load_bits<4>(a.sv_bits + 8 * BS, pidx, q);
        tlayer<LayoutB::F_HV, 4, 1, 0, true, 128, NB, NF>(c, Grgb, Grgb, B, m_hv, a.g_hv, pidx, q);
        // g_feat = W_views[:, :256]^T g_hv          (feature_linear has no activation)
        tlayer<LayoutB::F_FEAT, 8, 4, 0, false, 256, NB, NF>(c, B, B, A, none, a.g_feat, pidx, q);
        if constexpr (RAYG) {   // view-direction encoding: g_dirs = W_views[:, 256:]^T g_hv, then through the encoding
            float g[2][8 * KD];
            tenc<LayoutB::F_DIRS, KD, 4, NB, NF>(c, B, g);
            static_for<2>([&](auto cc_) { constexpr int cc = cc_; encode16_bwd<LD, KD>(dv[cc][0], dv[cc][1], dv[cc][2], hh, bb, g[cc], gd[cc]); });
        } else {
            skip_frags<LayoutB::F_DIRS, 8 * KD, NB, NF>(c);
        }
        // g_h8 = relu'(h8) * (W_feature^T g_feat + W_alpha^T g_sigma)
        m_next = load_bits<8>(a.sv_bits + 6 * BS, pidx, q);
        tlayer<LayoutB::F_H8, 8, 8, 1, true, 256, NB, NF>(c, A, Gsig, B, m_cur, a.g_h + 7 * HS, pidx, q);
    } else {
        // g_h8 = relu'(h8) * (W_output^T dL/draw)
        m_next = load_bits<8>(a.sv_bits + 6 * BS, pidx, q);
        tlayer<LayoutB::F_H8, 8, 1, 0, true, 256, NB, NF>(c, Gsig, Gsig, B, m_cur, a.g_h + 7 * HS, pidx, q);
    }
    // g_h(l-1) = relu'(h(l-1)) * (W_l^T g_h(l)),  l = 7 .. 1   (layer 5 uses the h-columns of its [e | h] input)
    m_cur = m_next; m_next = load_bits<8>(a.sv_bits + 5 * BS, pidx, q);
    tlayer<LayoutB::F_L7 + 0 * 128, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, m_cur, a.g_h + 6 * HS, pidx, q);
    m_cur = m_next; m_next = load_bits<8>(a.sv_bits + 4 * BS, pidx, q);
    tlayer<LayoutB::F_L7 + 1 * 128, 8, 8, 0, true, 256, NB, NF>(c, A, A, B, m_cur, a.g_h + 5 * HS, pidx, q);
    m_cur = m_next; m_next = load_bits<8>(a.sv_bits + 3 * BS, pidx, q);
    tlayer<LayoutB::F_L7 + 2 * 128, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, m_cur, a.g_h + 4 * HS, pidx, q);
    if constexpr (RAYG) {   // xyz encoding through the skip layer's [input_pts] columns (its pre-activation gradient is still in B)
        float g[2][8 * KE];
        tenc<LayoutB::F_E5, KE, 8, NB, NF>(c, B, g);
        static_for<2>([&](auto cc_) { constexpr int cc = cc_; encode16_bwd<LX, KE>(xp[cc][0], xp[cc][1], xp[cc][2], hh, bb, g[cc], gx[cc]); });
    } else {
        skip_frags<LayoutB::F_E5, 16 * KE, NB, NF>(c);
    }
    m_cur = m_next; m_next = load_bits<8>(a.sv_bits + 2 * BS, pidx, q);
    tlayer<LayoutB::F_L4 + 0 * 128, 8, 8, 0, true, 256, NB, NF>(c, A, A, B, m_cur, a.g_h + 3 * HS, pidx, q);
    m_cur = m_next; m_next = load_bits<8>(a.sv_bits + 1 * BS, pidx, q);
    tlayer<LayoutB::F_L4 + 1 * 128, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, m_cur, a.g_h + 2 * HS, pidx, q);
    m_cur = m_next; m_next = load_bits<8>(a.sv_bits + 0 * BS, pidx, q);
    tlayer<LayoutB::F_L4 + 2 * 128, 8, 8, 0, true, 256, NB, NF>(c, A, A, B, m_cur, a.g_h + 1 * HS, pidx, q);
    m_cur = m_next;
    tlayer<LayoutB::F_L4 + 3 * 128, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, m_cur, a.g_h + 0 * HS, pidx, q);
    if constexpr (RAYG) {   // xyz encoding through pts_linears.0
        float g[2][8 * KE];
        tenc<LayoutB::F_E0, KE, 8, NB, NF>(c, A, g);
        static_for<2>([&](auto cc_) { constexpr int cc = cc_; encode16_bwd<LX, KE>(xp[cc][0], xp[cc][1], xp[cc][2], hh, bb, g[cc], gx[cc]); });
    } else {
        skip_frags<LayoutB::F_E0, 16 * KE, NB, NF>(c);
    }
    // ---- point / ray gradients: sum the four lane quarters of each point, then one lane per point writes
    if constexpr (RAYG) static_for<2>([&](auto cc_) {
        constexpr int cc = cc_;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gx[cc][k] += __shfl_xor(gx[cc][k], 16); gx[cc][k] += __shfl_xor(gx[cc][k], 32);
            gd[cc][k] += __shfl_xor(gd[cc][k], 16); gd[cc][k] += __shfl_xor(gd[cc][k], 32);
        }
        if (valid[cc] && q == 0) {
            if (a.g_pts) { a.g_pts[3 * pidx[cc]] = gx[cc][0]; a.g_pts[3 * pidx[cc] + 1] = gx[cc][1]; a.g_pts[3 * pidx[cc] + 2] = gx[cc][2]; }
            if (a.g_rays) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    atomicAdd(a.g_rays + rayi[cc] * 6 + k, gx[cc][k]);
                    atomicAdd(a.g_rays + rayi[cc] * 6 + 3 + k, gx[cc][k] * zp[cc]);
                }
            }
            if constexpr (VD) {
                if (a.g_vd)
#pragma unroll
                    for (int k = 0; k < 3; ++k) atomicAdd(a.g_vd + rayi[cc] * 3 + k, gd[cc][k]);
            }
        }
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the workgroup
}

template <int LX, int LD, bool VD, bool RAYG>
static int launch_bwd_as(const MlpArgs &a, int n_frags_used, hipStream_t s) {
    constexpr int KE = gen16_ksteps(LX), KD = VD ? gen16_ksteps(LD) : 1;
    using C = Ctx<8, 16, NA_EXPERIMENT_BWD_NS, 8, 2, 0, 1, 0, BwdLedger<KE, KD, VD>>;
    if (n_frags_used != LayoutB<KE, KD, VD>::F_END) return NERF_AMD_EINVAL;
    if (a.P <= 0) return NERF_AMD_OK;
    if (a.P >= (int64_t)1 << 31) return NERF_AMD_EINVAL;
    if (!VD && a.out_ch > 16) return NERF_AMD_EUNSUPPORTED;
    const size_t lds = C::RING_BYTES;
    static DynamicLdsOptIn opt_in;
    if (opt_in.ensure(reinterpret_cast<const void *>(mlp_bwd_s16_kernel<LX, LD, VD, C, RAYG>), lds) != hipSuccess) return NERF_AMD_EHIP;
    const int64_t groups = (a.P + 255) / 256;
    hipLaunchKernelGGL((mlp_bwd_s16_kernel<LX, LD, VD, C, RAYG>), dim3((unsigned)groups), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

template <int LX, int LD, bool VD>
static int launch_bwd(const MlpArgs &a, int n_frags_used, hipStream_t s) {
    // the encoding products only when a gradient with respect to points, rays or view directions has a place to go
    if (a.g_pts || a.g_rays || a.g_vd) return launch_bwd_as<LX, LD, VD, true>(a, n_frags_used, s);
    return launch_bwd_as<LX, LD, VD, false>(a, n_frags_used, s);
}

int launch_mlp_bwd_s16(const MlpArgs &a, int multires, int multires_views, int use_viewdirs, int n_frags_used, hipStream_t s) {
    if (use_viewdirs) {
        if (multires == 10 && multires_views == 4) return launch_bwd<10, 4, true>(a, n_frags_used, s);
        if (multires == 15 && multires_views == 6) return launch_bwd<15, 6, true>(a, n_frags_used, s);
    } else {
        if (multires == 10) return launch_bwd<10, 0, false>(a, n_frags_used, s);
        if (multires == 15) return launch_bwd<15, 0, false>(a, n_frags_used, s);
    }
    return NERF_AMD_EUNSUPPORTED;
}

}  // namespace na
