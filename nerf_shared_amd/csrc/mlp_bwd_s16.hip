// mlp_bwd_s16.hip -- backward of the fused 8x256 NeRF MLP with respect to every
// layer's pre-activation (the "dX chain"), for the view-branch (10,4) model.
//
// What torch.autograd computes for NeRF.MLP (/root/reference/nerf_shared/nerf.py:110-134)
// given dL/draw: walking the layers in reverse,
//     g_pre(l) = relu'(h_l) * ( W_{l+1}^T g_pre(l+1) )
// is the forward kernel's computation with transposed weights: the gradient tile of one
// layer, converted to bf16 in place, is the B operand of the next (earlier) layer, so the
// whole chain stays in registers exactly like the forward activations (mlp_bf16_s16.hip).
// ReLU masks come from the activations the training forward saved; every g_pre(l) is
// written to HBM (slot-major bf16 rows, same layout as the saved activations) for the
// weight-gradient GEMMs  dW_l = g_pre(l)^T h_{l-1}  that follow (backward.hip).
// Encodings are treated as constants (no ray gradients yet).
#include <hip/hip_runtime.h>
#include <utility>

#include "kernels.h"
#include "pipeline.h"
#include "program.h"

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
template <bool MASK>
__device__ __forceinline__ bf16x8 pack_grad(const f32x4 &even, const f32x4 &odd, const bf16x8 &h) {
    bf16x8 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) { y[r] = (__bf16)even[r]; y[4 + r] = (__bf16)odd[r]; }
    if (MASK) {
        u32x4 v = __builtin_bit_cast(u32x4, y);
        const u32x4 hv = __builtin_bit_cast(u32x4, h);
        const unsigned ones = 0x00010001u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned m;
            asm("v_pk_min_u16 %0, %1, %2\n\tv_pk_sub_u16 %0, 0, %0" : "=&v"(m) : "v"(hv[i]), "v"(ones));   // 0xffff where h != 0
            v[i] &= m;
        }
        y = __builtin_bit_cast(bf16x8, v);
    }
    return y;
}

// One transposed layer: NPAIR pairs of input-feature tiles -> g[2 * NPAIR]; g is masked with the
// saved activation rows at `mask` (ROW elements per point) and stored to `dst`.
template <int F0, int NPAIR, int K1, int K2, bool MASK, int ROW, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tlayer(C &c, const bf16x8 *x1, const bf16x8 *x2, bf16x8 *g, const uint16_t *mask,
                                       uint16_t *dst, const int64_t (&pidx)[2], const bool (&valid)[2], int q) {
    static_for<NPAIR>([&](auto p_) {
        constexpr int p = p_;
        bf16x8 h[2];
        if constexpr (MASK) {
            static_for<2>([&](auto cc_) {
                constexpr int cc = cc_;
                const int64_t row = valid[cc] ? pidx[cc] : 0;
                h[cc] = *reinterpret_cast<const bf16x8 *>(mask + row * ROW + p * 32 + q * 8);
            });
        }
        f32x4 acc[2][2];
        tpair<F0 + p * 2 * (K1 + K2), K1, K2, NB, NFRAGS>(c, x1, x2, acc);
        g[2 * p] = pack_grad<MASK>(acc[0][0], acc[1][0], h[0]);
        g[2 * p + 1] = pack_grad<MASK>(acc[0][1], acc[1][1], h[1]);
        static_for<2>([&](auto cc_) {
            constexpr int cc = cc_;
            if (valid[cc]) *reinterpret_cast<bf16x8 *>(dst + pidx[cc] * ROW + p * 32 + q * 8) = g[2 * p + cc];
        });
    });
}

struct LayoutB {
    static constexpr int F_HV = 0;                 // 4 pairs x 1 k-step
    static constexpr int F_FEAT = F_HV + 8;        // 8 pairs x 4 k-steps
    static constexpr int F_H8 = F_FEAT + 64;       // 8 pairs x (8 + 1) k-steps
    static constexpr int F_H7 = F_H8 + 144;        // then 7 layers of 8 pairs x 8 k-steps
    static constexpr int F_END = F_H7 + 7 * 128;
};

template <class C>
__global__ __launch_bounds__(C::WAVES * 64, 2) void mlp_bwd_s16_kernel(MlpArgs a) {
    constexpr int WG_POINTS = C::WAVES * 32;
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

    int64_t pidx[2];
    bool valid[2];
    bf16x8 Grgb[2], Gsig[2];
    static_for<2>([&](auto cc_) {
        constexpr int cc = cc_;
        const int64_t p = (int64_t)blockIdx.x * WG_POINTS + c.wave * 32 + cc * 16 + (lane & 15);
        pidx[cc] = p;
        valid[cc] = p < a.P;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (valid[cc] && q == 0) g = *reinterpret_cast<const f32x4 *>(a.g_raw + 4 * p);
        bf16x8 r = {}, s = {};
        r[0] = (__bf16)g[0]; r[1] = (__bf16)g[1]; r[2] = (__bf16)g[2];     // k slot (q=0, j) = rgb_linear output j
        s[0] = (__bf16)g[3];                                                // k slot (0, 0)   = alpha_linear output
        Grgb[cc] = r; Gsig[cc] = s;
        if (valid[cc] && q == 0) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4 o = {r[0], r[1], r[2], s[0]};
            *reinterpret_cast<bf16x4 *>(a.g_rawb + 4 * p) = o;
        }
    });

    if constexpr (C::PHASE > 0) {
        block_sync<-1, NB>(c);
        static_for<C::LA>([&](auto i_) { constexpr int i = i_; c.q[i] = ring_frag<i>(c); });
    }
    const int64_t HS = a.P * 256;
    bf16x8 A[16], B[16];
    // g_hv = relu'(hv) * (W_rgb^T g_rgb)
    tlayer<LayoutB::F_HV, 4, 1, 0, true, 128, NB, NF>(c, Grgb, Grgb, B, a.sv_hv, a.g_hv, pidx, valid, q);
    // g_feat = W_views[:, :256]^T g_hv          (feature_linear has no activation)
    tlayer<LayoutB::F_FEAT, 8, 4, 0, false, 256, NB, NF>(c, B, B, A, nullptr, a.g_feat, pidx, valid, q);
    // g_h8 = relu'(h8) * (W_feature^T g_feat + W_alpha^T g_sigma)
    tlayer<LayoutB::F_H8, 8, 8, 1, true, 256, NB, NF>(c, A, Gsig, B, a.sv_h + 7 * HS, a.g_h + 7 * HS, pidx, valid, q);
    // g_h(l-1) = relu'(h(l-1)) * (W_l^T g_h(l)),  l = 7 .. 1   (layer 5 uses the h-columns of its [e | h] input)
    tlayer<LayoutB::F_H7 + 0 * 128, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, a.sv_h + 6 * HS, a.g_h + 6 * HS, pidx, valid, q);
    tlayer<LayoutB::F_H7 + 1 * 128, 8, 8, 0, true, 256, NB, NF>(c, A, A, B, a.sv_h + 5 * HS, a.g_h + 5 * HS, pidx, valid, q);
    tlayer<LayoutB::F_H7 + 2 * 128, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, a.sv_h + 4 * HS, a.g_h + 4 * HS, pidx, valid, q);
    tlayer<LayoutB::F_H7 + 3 * 128, 8, 8, 0, true, 256, NB, NF>(c, A, A, B, a.sv_h + 3 * HS, a.g_h + 3 * HS, pidx, valid, q);
    tlayer<LayoutB::F_H7 + 4 * 128, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, a.sv_h + 2 * HS, a.g_h + 2 * HS, pidx, valid, q);
    tlayer<LayoutB::F_H7 + 5 * 128, 8, 8, 0, true, 256, NB, NF>(c, A, A, B, a.sv_h + 1 * HS, a.g_h + 1 * HS, pidx, valid, q);
    tlayer<LayoutB::F_H7 + 6 * 128, 8, 8, 0, true, 256, NB, NF>(c, B, B, A, a.sv_h + 0 * HS, a.g_h + 0 * HS, pidx, valid, q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the workgroup
}

int launch_mlp_bwd_s16(const MlpArgs &a, int n_frags_used, hipStream_t s) {
    using C = Ctx<8, 16, 4, 8, 2>;
    if (n_frags_used != LayoutB::F_END) return NERF_AMD_EINVAL;
    if (a.P <= 0) return NERF_AMD_OK;
    if (a.P >= (int64_t)1 << 31) return NERF_AMD_EINVAL;
    const size_t lds = C::RING_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_bwd_s16_kernel<C>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return NERF_AMD_EHIP;
        attr_set = true;
    }
    const int64_t groups = (a.P + 255) / 256;
    hipLaunchKernelGGL((mlp_bwd_s16_kernel<C>), dim3((unsigned)groups), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

}  // namespace na
