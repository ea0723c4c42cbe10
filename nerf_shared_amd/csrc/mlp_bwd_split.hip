// mlp_bwd_split.hip -- the dX chain of the fused 8x256 NeRF MLP at fp32-class accuracy (NERF_AMD_PREC_FP32_SPLIT):
// mlp_bwd_s16.hip's computation with every operand an fp16 (hi, lo) pair and every product three MFMAs (split.h).
//
// What torch.autograd computes for NeRF.MLP (/root/reference/nerf_shared/nerf.py:110-134) given dL/draw, in the fp32
// the reference trains and pose-optimises in (main.py:85-104, demo_est_rel_pose.py:87-98): walking the layers in reverse
//     g_pre(l) = relu'(h_l) * ( W_{l+1}^T g_pre(l+1) )
// with the transposed weights streamed as fp16 pairs (program.h frags_bwd_split) and the gradient tile of one layer,
// split into (hi, lo) in place, as the B operand of the next (earlier) layer.  Like mlp_split.hip a wave owns 16 points
// (its two column tiles are the hi and the lo half of one tile), a workgroup 128.
//   * dL/draw enters multiplied by a power-of-two loss scale S (split.h: fp16's range), derived from the launch's own
//     maximum by gmax_kernel; every g_pre(l) is stored scaled, as a plane of hi rows and a plane of lo rows, for the
//     weight-gradient products (backward.hip), whose reduction divides by S; ray gradients are divided here.
//   * ReLU masks come from the bit rows of the training forward (mlp_split.hip save_bits_split).
//   * Ray gradients: the encoding-slot products (FRAG_TE16) give every lane the gradients of the encoding values it
//     generated; the derivative d/dx sin(2^f x) = 2^f cos(2^f x) is the partner lane's saved encoding value times 2^f.
#include <hip/hip_runtime.h>
#include <utility>

#include "kernels.h"
#include "launch_util.h"
#include "pipeline.h"
#include "program.h"
#include "split.h"

namespace na {

// Partial maxima of |dL/draw| for the loss scale: GRAD_SCALE_PARTS blocks, one partial each (NaNs are skipped by fmaxf;
// an infinity yields S = 1 and propagates as itself).
__global__ __launch_bounds__(256) void gmax_kernel(const float *g, int64_t n, float *parts) {
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, __builtin_fabsf(g[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// Pair of 16-row tiles of W^T over K1 k-steps of x1 and K2 of x2 (x[2k] = hi, x[2k+1] = lo), no bias; per k-step the
// stream holds hi(tile 0), hi(tile 1), lo(tile 0), lo(tile 1).  acc[u][0] = hi x hi sums of tile u, acc[u][1] = the two
// cross terms (scaled by 2^11).
template <int F0, int K1, int K2, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tpair_split(C &c, const bf16x8 *x1, const bf16x8 *x2, f32x4 (&acc)[2][2]) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    acc[0][0] = zero; acc[0][1] = zero; acc[1][0] = zero; acc[1][1] = zero;
    auto kstep = [&](auto n_, const bf16x8 &xh, const bf16x8 &xl) {
        constexpr int n = n_;
        const bf16x8 w0h = take<n, NB, NFRAGS>(c);
        const bf16x8 w1h = take<n + 1, NB, NFRAGS>(c);
        const bf16x8 w0l = take<n + 2, NB, NFRAGS>(c);
        const bf16x8 w1l = take<n + 3, NB, NFRAGS>(c);
        acc[0][0] = MFMAH(w0h, xh, acc[0][0]);
        acc[1][0] = MFMAH(w1h, xh, acc[1][0]);
        acc[0][1] = MFMAH(w0h, xl, acc[0][1]);
        acc[1][1] = MFMAH(w1h, xl, acc[1][1]);
        acc[0][1] = MFMAH(w0l, xh, acc[0][1]);
        acc[1][1] = MFMAH(w1l, xh, acc[1][1]);
        sched_step_split<C, 4, 6>();
    };
    static_for<K1>([&](auto k_) { constexpr int k = k_; kstep(std::integral_constant<int, F0 + 4 * k>{}, x1[2 * k], x1[2 * k + 1]); });
    static_for<K2>([&](auto k_) { constexpr int k = k_; kstep(std::integral_constant<int, F0 + 4 * K1 + 4 * k>{}, x2[2 * k], x2[2 * k + 1]); });
}

// accumulators of a tile pair -> the (hi, lo) gradient fragments of one k-step of the next (earlier) layer, zeroed where
// the saved activation is zero (ReLU').  The mask bits of this fragment sit at BIT0 + i (element 2i) and 16 + BIT0 + i
// (element 2i + 1) of `bits` (mlp_split.hip save_bits_split).
template <bool MASK, int BIT0>
__device__ __forceinline__ void pack_grad_split(const f32x4 (&acc)[2][2], unsigned bits, bf16x8 &yh, bf16x8 &yl) {
    f16x8 h, l;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            float v0 = __builtin_fmaf(acc[u][1][r], SPLIT_INV, acc[u][0][r]);
            float v1 = __builtin_fmaf(acc[u][1][r + 1], SPLIT_INV, acc[u][0][r + 1]);
            if (MASK) {
                const int i = (4 * u + r) >> 1;
                v0 = ((bits >> (BIT0 + i)) & 1u) ? v0 : 0.0f;
                v1 = ((bits >> (16 + BIT0 + i)) & 1u) ? v1 : 0.0f;
            }
            f16x2 a, b;
            split_f16_pair(v0, v1, a, b);
            h[4 * u + r] = a[0]; h[4 * u + r + 1] = a[1];
            l[4 * u + r] = b[0]; l[4 * u + r + 1] = b[1];
        }
    yh = __builtin_bit_cast(bf16x8, h);
    yl = __builtin_bit_cast(bf16x8, l);
}

// Mask bytes of one layer for this lane's point (NPAIR bytes), loaded one layer ahead of their use.
template <int NPAIR>
struct MaskBitsS {
    unsigned w[NPAIR / 4];
};
template <int NPAIR>
__device__ __forceinline__ MaskBitsS<NPAIR> load_bits_split(const uint8_t *base, int64_t p, int q) {
    static_assert(NPAIR == 4 || NPAIR == 8, "one or two dwords of mask bytes");
    MaskBitsS<NPAIR> m;
    const uint8_t *src = base + p * (4 * NPAIR) + q * NPAIR;
    if constexpr (NPAIR == 8) {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        const u32x2 v = *reinterpret_cast<const u32x2 *>(src);
        m.w[0] = v[0]; m.w[1] = v[1];
    } else {
        m.w[0] = *reinterpret_cast<const unsigned *>(src);
    }
    return m;
}

// One transposed layer: NPAIR pairs of input-feature tiles -> g[2 * NPAIR] (k-step p: g[2p] = hi, g[2p+1] = lo), masked
// with the layer's ReLU bits and stored to the hi / lo planes (ROW values per point; rows of padding points included:
// the store count per pair is a compile-time constant, see BwdSplitLedger).
template <int F0, int NPAIR, int K1, int K2, bool MASK, int ROW, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tlayer_split(C &c, const bf16x8 *x1, const bf16x8 *x2, bf16x8 *g, const MaskBitsS<NPAIR> &mask,
                                             uint16_t *dst_hi, uint16_t *dst_lo, int64_t p, int q) {
    static_for<NPAIR>([&](auto pp_) {
        constexpr int pp = pp_;
        f32x4 acc[2][2];
        tpair_split<F0 + pp * 4 * (K1 + K2), K1, K2, NB, NFRAGS>(c, x1, x2, acc);
        pack_grad_split<MASK, 4 * (pp % 4)>(acc, mask.w[pp / 4], g[2 * pp], g[2 * pp + 1]);
        *reinterpret_cast<bf16x8 *>(dst_hi + p * ROW + pp * 32 + q * 8) = g[2 * pp];
        *reinterpret_cast<bf16x8 *>(dst_lo + p * ROW + pp * 32 + q * 8) = g[2 * pp + 1];
    });
}

// Transposed products whose rows are the slots of a generated encoding (FRAG_TE16): NPAIR pairs, K1 k-steps of x1 each;
// g[8*pair + 4*u + r] receives this lane's slot gradients (fp32).
template <int F0, int NPAIR, int K1, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tenc_split(C &c, const bf16x8 *x1, float (&g)[8 * NPAIR]) {
    static_for<NPAIR>([&](auto pp_) {
        constexpr int pp = pp_;
        f32x4 acc[2][2];
        tpair_split<F0 + pp * 4 * K1, K1, 0, NB, NFRAGS>(c, x1, x1, acc);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) g[8 * pp + 4 * u + r] = __builtin_fmaf(acc[u][1][r], SPLIT_INV, acc[u][0][r]);
    });
}

// Gradient of the reference embedding as mlp_split.hip encode_split generates it (slot i = 8 ks + j of lane quarter
// (h, b): sin (h = 0) / cos (h = 1) of x[i % 3] 2^(2 (i / 3) + b) while i < gen16_ntrig(L, b), then raw coordinates) with
// respect to its 3 inputs; g[i] = dL/d(slot i).  No trigonometry here: d/dx sin(2^f x) = 2^f cos(2^f x) and d/dx cos = -2^f sin,
// and those values are what the PARTNER lane quarter (1 - h, b) generated in the same slots of the forward -- they are read
// back from the saved encoding rows (hi + 2^-11 lo: 22 bits of the forward's accurate sincosf).
template <int L, int K, int ROW>
__device__ __forceinline__ void encode_split_bwd(const uint16_t *sv_hi, const uint16_t *sv_lo, int64_t p, int h, int b, const float *g,
                                                 float (&gx)[3]) {
    constexpr int N_EVEN = gen16_ntrig(L, 0), N_ODD = gen16_ntrig(L, 1), CAP = 8 * K;
    const float s0 = b ? 2.0f : 1.0f;
    const int n_mine = b ? N_ODD : N_EVEN;
    const int qp = 2 * (1 - h) + b;                                 // the partner quarter
    const float sgn = h ? -1.0f : 1.0f;
    static_for<K>([&](auto ks_) {
        constexpr int ks = ks_;
        const f16x8 ph = *reinterpret_cast<const f16x8 *>(sv_hi + p * ROW + ks * 32 + qp * 8);
        const f16x8 pl = *reinterpret_cast<const f16x8 *>(sv_lo + p * ROW + ks * 32 + qp * 8);
        static_for<8>([&](auto j_) {
            constexpr int j = j_, i = 8 * ks + j;
            if constexpr (i < N_EVEN || i < N_ODD) {
                const float scale = __builtin_ldexpf(1.0f, 2 * (i / 3)) * s0;
                const float other = __builtin_fmaf((float)pl[j], SPLIT_INV, (float)ph[j]);
                gx[i % 3] += (i < n_mine) ? g[i] * (other * (sgn * scale)) : 0.0f;
            }
            if constexpr (i >= N_EVEN || i >= N_ODD) {              // raw-coordinate slots: derivative 1
#pragma unroll
                for (int cidx = 0; cidx < 3; ++cidx) {
                    const bool even_hit = i >= N_EVEN && (h ? gen16_misc(L, 1, 0, i - N_EVEN) : gen16_misc(L, 0, 0, i - N_EVEN)) == cidx;
                    const bool odd_hit = i >= N_ODD && (h ? gen16_misc(L, 1, 1, i - N_ODD) : gen16_misc(L, 0, 1, i - N_ODD)) == cidx;
                    gx[cidx] += (b ? odd_hit : even_hit) ? g[i] : 0.0f;
                }
            }
        });
    });
}

// Fragment offsets of the split backward stream (program.cpp frags_bwd_split): mlp_bwd_s16.hip's LayoutB with four
// fragments per k-step instead of two.
template <int KE, int KD, bool VD = true>
struct LayoutBS {
    static constexpr int F_HV = 0;                       // 4 pairs x 1 k-step
    static constexpr int F_FEAT = F_HV + 16;             // 8 pairs x 4 k-steps
    static constexpr int F_DIRS = F_FEAT + 128;          // view-direction encoding slots: KD pairs x 4 k-steps
    static constexpr int F_H8 = VD ? F_DIRS + 16 * KD : 0;   // 8 pairs x (8 + 1) k-steps (VD) / x 1 k-step
    static constexpr int H8_FRAGS_PER_PAIR = VD ? 36 : 4;
    static constexpr int F_L7 = F_H8 + 8 * H8_FRAGS_PER_PAIR;   // pts_linears.7, .6, .5: 8 pairs x 8 k-steps each
    static constexpr int F_E5 = F_L7 + 3 * 256;          // xyz encoding slots through pts_linears.5: KE pairs x 8 k-steps
    static constexpr int F_L4 = F_E5 + 32 * KE;          // pts_linears.4 .. .1
    static constexpr int F_E0 = F_L4 + 4 * 256;          // xyz encoding slots through pts_linears.0
    static constexpr int F_END = F_E0 + 32 * KE;
};

// Row stores issued before fragment n (pipeline.h LEDGER): two (hi, lo) per finished tile pair of a tlayer_split.
template <int KE, int KD, bool VD = true>
struct BwdSplitLedger {
    using L = LayoutBS<KE, KD, VD>;
    static constexpr int pairs_done(int n, int f0, int frags_per_pair, int n_pairs) {
        const int d = n <= f0 ? 0 : (n - f0) / frags_per_pair;
        return d > n_pairs ? n_pairs : d;
    }
    static constexpr int stores_before(int n) {
        return 2 * ((VD ? pairs_done(n, L::F_HV, 4, 4) + pairs_done(n, L::F_FEAT, 16, 8) : 0) +
                    pairs_done(n, L::F_H8, L::H8_FRAGS_PER_PAIR, 8) +
                    pairs_done(n, L::F_L7, 32, 24) + pairs_done(n, L::F_L4, 32, 32));
    }
};

// RAYG as in mlp_bwd_s16.hip: without a taker for dL/dpts, dL/drays or dL/dviewdirs the encoding products are skipped.
template <int LX, int LD, bool VD, class C, bool RAYG>
__global__ __launch_bounds__(C::WAVES * 64, 2) void mlp_bwd_split_kernel(MlpArgs a) {
    constexpr int WG_POINTS = C::WAVES * 16;
    constexpr int KE = gen16_ksteps(LX), KD = VD ? gen16_ksteps(LD) : 1;
    using Lay = LayoutBS<KE, KD, VD>;
    constexpr int NF = Lay::F_END, NB = (NF + C::BF - 1) / C::BF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = lane >> 4;
    C c;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.lag = 0;
    c.phase = __builtin_amdgcn_readfirstlane(C::N_PHASES == 4 ? ((c.wave + 2 * (c.wave >> 2)) & 3) : (c.wave >= C::WAVES / 2 ? 2 : 0));
    c.gstream = reinterpret_cast<const char *>(a.stream_bwd_split) + lane * 16;
    c.ring_lane = smem + lane * 16;
    c.ring_u32 = (uint32_t)(uintptr_t)smem;
    c.bias_half = nullptr;
    if constexpr ((C::OPT & 64) != 0) {
        c.rsrc = make_rsrc(a.stream_bwd_split, (unsigned)(NB * C::BLOCK_BYTES));
        c.rsrc_next = c.rsrc;
        c.lane16 = lane * 16;
        c.wave_off = c.wave * C::PIECES * 1024;
    }

    pipeline_prologue<NB>(c);

    // ---- the loss scale (every wave derives the same value), this lane's point, dL/draw as the first B operands
    const float S = grad_scale_of(a.g_scale);
    const float S_inv = __builtin_bit_cast(float, grad_scale_inv_bits(__builtin_bit_cast(unsigned, S)));
    if (blockIdx.x == 0 && tid == 0) { a.g_scale[GRAD_SCALE_PARTS] = S; a.g_scale[GRAD_SCALE_PARTS + 1] = S_inv; }
    const int64_t p = (int64_t)blockIdx.x * WG_POINTS + c.wave * 16 + (lane & 15);
    const bool valid = p < a.P;
    const int64_t pc = valid ? p : a.P - 1;
    const int64_t ray = (int64_t)((uint32_t)pc / (uint32_t)a.S);
    const float zp = a.pts ? 0.f : a.z_vals[pc];                   // dL/dd = z dL/dpts (rays mode)
    bf16x8 Grgb[2], Gsig[2];                                       // (hi, lo) of the one k-step dL/draw fills
    if constexpr (VD) {
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (valid) g = *reinterpret_cast<const f32x4 *>(a.g_raw + 4 * p);
        _Float16 gh[4], gl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) split_f16(g[k] * S, gh[k], gl[k]);
        f16x8 rh = {}, rl = {}, sh = {}, sl = {};
        if (q == 0) {
            rh[0] = gh[0]; rh[1] = gh[1]; rh[2] = gh[2];           // k slot (q = 0, j) = rgb_linear output j
            rl[0] = gl[0]; rl[1] = gl[1]; rl[2] = gl[2];
            sh[0] = gh[3]; sl[0] = gl[3];                           // k slot (0, 0)   = alpha_linear output
        }
        Grgb[0] = __builtin_bit_cast(bf16x8, rh); Grgb[1] = __builtin_bit_cast(bf16x8, rl);
        Gsig[0] = __builtin_bit_cast(bf16x8, sh); Gsig[1] = __builtin_bit_cast(bf16x8, sl);
        // the same values as the operand of the head weight-gradient products: transposed inside 32-point chunks (kernels.h
        // g_rawt), lane quarter q writes column q; zeros for the padding points
        {
            const int pl = (int)(p & 31);
            const _Float16 vh = q == 0 ? gh[0] : q == 1 ? gh[1] : q == 2 ? gh[2] : gh[3];
            const _Float16 vl = q == 0 ? gl[0] : q == 1 ? gl[1] : q == 2 ? gl[2] : gl[3];
            const int64_t at = (p >> 5) * 128 + ((pl >> 3) * 4 + q) * 8 + (pl & 7);
            reinterpret_cast<_Float16 *>(a.g_rawt)[at] = vh;
            reinterpret_cast<_Float16 *>(a.g_rawt_lo)[at] = vl;
        }
    } else {
        // dL/draw [P, out_ch] as the FRAG_TG16 operand: k slot (q, j) = output_linear row 8 q + j (q < 2)
        f16x8 oh = {}, ol = {};
        if (q < 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = 8 * q + j;
                const float gv = (valid && col < a.out_ch) ? a.g_raw[(int64_t)a.out_ch * p + col] * S : 0.0f;
                _Float16 hh, ll;
                split_f16(gv, hh, ll);
                oh[j] = hh; ol[j] = ll;
            }
        }
        Gsig[0] = __builtin_bit_cast(bf16x8, oh); Gsig[1] = __builtin_bit_cast(bf16x8, ol);
        Grgb[0] = Gsig[0]; Grgb[1] = Gsig[1];
    }

    if constexpr (C::PHASE > 0) {
        block_sync<-1, NB>(c);
        static_for<C::LA>([&](auto i_) { constexpr int i = i_; c.q[i] = ring_frag<i>(c); });
    }
    const int64_t HS = pad_points(a.P) * 256, BS = pad_points(a.P) * 32;
    const int hh = q >> 1, bb = q & 1;
    bf16x8 A[16], B[16];
    float gx[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
    const MaskBitsS<8> none = {};
    MaskBitsS<8> m_cur = load_bits_split<8>(a.sv_bits + 7 * BS, p, q), m_next;
    if constexpr (VD) {
        // g_hv = relu'(hv) * (W_rgb^T g_rgb)
        const MaskBitsS<4> m_hv = load_bits_split<4>(a.sv_bits + 8 * BS, p, q);
        tlayer_split<Lay::F_HV, 4, 1, 0, true, 128, NB, NF>(c, Grgb, Grgb, B, m_hv, a.g_hv, a.g_hv_lo, p, q);
        // g_feat = W_views[:, :256]^T g_hv          (feature_linear has no activation)
        tlayer_split<Lay::F_FEAT, 8, 4, 0, false, 256, NB, NF>(c, B, B, A, none, a.g_feat, a.g_feat_lo, p, q);
        if constexpr (RAYG) {   // view-direction encoding: g_dirs = W_views[:, 256:]^T g_hv, then through the encoding
            float g[8 * KD];
            tenc_split<Lay::F_DIRS, KD, 4, NB, NF>(c, B, g);
            encode_split_bwd<LD, KD, 32 * KD>(a.sv_d, a.sv_d_lo, p, hh, bb, g, gd);
        } else {
            skip_frags<Lay::F_DIRS, 16 * KD, NB, NF>(c);
        }
        // g_h8 = relu'(h8) * (W_feature^T g_feat + W_alpha^T g_sigma)
        m_next = load_bits_split<8>(a.sv_bits + 6 * BS, p, q);
        tlayer_split<Lay::F_H8, 8, 8, 1, true, 256, NB, NF>(c, A, Gsig, B, m_cur, a.g_h + 7 * HS, a.g_h_lo + 7 * HS, p, q);
    } else {
        // g_h8 = relu'(h8) * (W_output^T dL/draw)
        m_next = load_bits_split<8>(a.sv_bits + 6 * BS, p, q);
        tlayer_split<Lay::F_H8, 8, 1, 0, true, 256, NB, NF>(c, Gsig, Gsig, B, m_cur, a.g_h + 7 * HS, a.g_h_lo + 7 * HS, p, q);
    }
    // g_h(l-1) = relu'(h(l-1)) * (W_l^T g_h(l)),  l = 7 .. 1   (layer 5 uses the h-columns of its [e | h] input)
    m_cur = m_next; m_next = load_bits_split<8>(a.sv_bits + 5 * BS, p, q);
    tlayer_split<Lay::F_L7 + 0 * 256, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, m_cur, a.g_h + 6 * HS, a.g_h_lo + 6 * HS, p, q);
    m_cur = m_next; m_next = load_bits_split<8>(a.sv_bits + 4 * BS, p, q);
    tlayer_split<Lay::F_L7 + 1 * 256, 8, 8, 0, true, 256, NB, NF>(c, A, A, B, m_cur, a.g_h + 5 * HS, a.g_h_lo + 5 * HS, p, q);
    m_cur = m_next; m_next = load_bits_split<8>(a.sv_bits + 3 * BS, p, q);
    tlayer_split<Lay::F_L7 + 2 * 256, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, m_cur, a.g_h + 4 * HS, a.g_h_lo + 4 * HS, p, q);
    if constexpr (RAYG) {   // xyz encoding through the skip layer's [input_pts] columns (its pre-activation gradient is still in B)
        float g[8 * KE];
        tenc_split<Lay::F_E5, KE, 8, NB, NF>(c, B, g);
        encode_split_bwd<LX, KE, (KE == 3 ? 128 : 32 * KE)>(a.sv_e, a.sv_e_lo, p, hh, bb, g, gx);
    } else {
        skip_frags<Lay::F_E5, 32 * KE, NB, NF>(c);
    }
    m_cur = m_next; m_next = load_bits_split<8>(a.sv_bits + 2 * BS, p, q);
    tlayer_split<Lay::F_L4 + 0 * 256, 8, 8, 0, true, 256, NB, NF>(c, A, A, B, m_cur, a.g_h + 3 * HS, a.g_h_lo + 3 * HS, p, q);
    m_cur = m_next; m_next = load_bits_split<8>(a.sv_bits + 1 * BS, p, q);
    tlayer_split<Lay::F_L4 + 1 * 256, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, m_cur, a.g_h + 2 * HS, a.g_h_lo + 2 * HS, p, q);
    m_cur = m_next; m_next = load_bits_split<8>(a.sv_bits + 0 * BS, p, q);
    tlayer_split<Lay::F_L4 + 2 * 256, 8, 8, 0, true, 256, NB, NF>(c, A, A, B, m_cur, a.g_h + 1 * HS, a.g_h_lo + 1 * HS, p, q);
    m_cur = m_next;
    tlayer_split<Lay::F_L4 + 3 * 256, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, m_cur, a.g_h + 0 * HS, a.g_h_lo + 0 * HS, p, q);
    if constexpr (RAYG) {   // xyz encoding through pts_linears.0
        float g[8 * KE];
        tenc_split<Lay::F_E0, KE, 8, NB, NF>(c, A, g);
        encode_split_bwd<LX, KE, (KE == 3 ? 128 : 32 * KE)>(a.sv_e, a.sv_e_lo, p, hh, bb, g, gx);
    } else {
        skip_frags<Lay::F_E0, 32 * KE, NB, NF>(c);
    }
    // ---- point / ray gradients: sum the four lane quarters of the point, take the loss scale off, one lane writes
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gx[k] += __shfl_xor(gx[k], 16); gx[k] += __shfl_xor(gx[k], 32);
        gd[k] += __shfl_xor(gd[k], 16); gd[k] += __shfl_xor(gd[k], 32);
        gx[k] *= S_inv; gd[k] *= S_inv;
    }
    if (RAYG && valid && q == 0) {
        if (a.g_pts) { a.g_pts[3 * p] = gx[0]; a.g_pts[3 * p + 1] = gx[1]; a.g_pts[3 * p + 2] = gx[2]; }
        if (a.g_rays) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                atomicAdd(a.g_rays + ray * 6 + k, gx[k]);
                atomicAdd(a.g_rays + ray * 6 + 3 + k, gx[k] * zp);
            }
        }
        if constexpr (VD) {
            if (a.g_vd)
#pragma unroll
                for (int k = 0; k < 3; ++k) atomicAdd(a.g_vd + ray * 3 + k, gd[k]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the workgroup
}

template <int LX, int LD, bool VD, bool RAYG>
static int launch_bwd_split_as(const MlpArgs &a, int n_frags_used, hipStream_t s) {
    constexpr int KE = gen16_ksteps(LX), KD = VD ? gen16_ksteps(LD) : 1;
    using C = Ctx<8, 16, 4, 8, 2, 0, 1, 0, BwdSplitLedger<KE, KD, VD>>;
    if (n_frags_used != LayoutBS<KE, KD, VD>::F_END) return NERF_AMD_EINVAL;
    if (a.P <= 0) return NERF_AMD_OK;
    if (a.P >= (int64_t)1 << 31) return NERF_AMD_EINVAL;
    if (!VD && a.out_ch > 16) return NERF_AMD_EUNSUPPORTED;
    hipLaunchKernelGGL(gmax_kernel, dim3(GRAD_SCALE_PARTS), dim3(256), 0, s, a.g_raw, a.P * (int64_t)a.out_ch, a.g_scale);
    const size_t lds = C::RING_BYTES;
    static DynamicLdsOptIn opt_in;
    if (opt_in.ensure(reinterpret_cast<const void *>(mlp_bwd_split_kernel<LX, LD, VD, C, RAYG>), lds) != hipSuccess) return NERF_AMD_EHIP;
    const int64_t groups = (a.P + 127) / 128;
    hipLaunchKernelGGL((mlp_bwd_split_kernel<LX, LD, VD, C, RAYG>), dim3((unsigned)groups), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

template <int LX, int LD, bool VD>
static int launch_bwd_split(const MlpArgs &a, int n_frags_used, hipStream_t s) {
    if (a.g_pts || a.g_rays || a.g_vd) return launch_bwd_split_as<LX, LD, VD, true>(a, n_frags_used, s);
    return launch_bwd_split_as<LX, LD, VD, false>(a, n_frags_used, s);
}

int launch_mlp_bwd_split(const MlpArgs &a, int multires, int multires_views, int use_viewdirs, int n_frags_used, hipStream_t s) {
    if (use_viewdirs) {
        if (multires == 10 && multires_views == 4) return launch_bwd_split<10, 4, true>(a, n_frags_used, s);
        if (multires == 15 && multires_views == 6) return launch_bwd_split<15, 6, true>(a, n_frags_used, s);
    } else {
        if (multires == 10) return launch_bwd_split<10, 0, false>(a, n_frags_used, s);
        if (multires == 15) return launch_bwd_split<15, 0, false>(a, n_frags_used, s);
    }
    return NERF_AMD_EUNSUPPORTED;
}

}  // namespace na
