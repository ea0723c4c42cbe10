// mlp_split.hip -- the fused positional-encoding + 8x256 NeRF MLP at fp32-class accuracy on the 16-bit matrix
// pipe: every operand is an unevaluated pair of fp16 values and every product is three MFMAs
// (v_mfma_f32_16x16x32_f16, fp32 accumulate).  NERF_AMD_PREC_FP32_SPLIT.
//
// Replaces NeRF.forward + NeRF.MLP (/root/reference/nerf_shared/nerf.py:96-134) like mlp_bf16_s16.hip does,
// for callers that need the reference's fp32 results ("fp32 (or split) variant
// for tight-tolerance parity") at more than the fp32 MFMA rate of mlp_fp32.hip.
//
// Arithmetic.  x = x_hi + 2^-11 x_lo with x_hi = fp16(x) and x_lo = fp16((x - x_hi) 2^11): 22 significant bits,
// the residual is exact in fp32, the 2^11 keeps x_lo a normal fp16 number.  Weights are split the same way
// when they are packed.  A product W x becomes
//     acc_h += W_hi x_hi                    (one MFMA)
//     acc_l += W_hi x_lo + W_lo x_hi        (two MFMAs)          y = acc_h + 2^-11 acc_l + bias
// and the dropped W_lo x_lo term is 2^-22 relative.  (bf16 pairs carry 16 bits: measured on the x3 weight sets
// against an fp64 evaluation they miss the fp32 gates, 1e-4 + 1e-4 |y|, by 4-6x; fp16 pairs are inside them
// by 4x, i.e. as far from fp64 as the fp32 reference itself.)  Range: |activation| must stay below 65504.
//
// Layout.  The machinery of mlp_bf16_s16.hip (program.h "s16" layout, pipeline.h ring) with the two column
// tiles of a wave re-used as (hi, lo) of ONE 16-point tile: a wave owns 16 points, a workgroup 128.
// Activation fragment of k-step k: x[2k] = hi, x[2k+1] = lo.  The weight stream holds, per pair of 16-row
// output tiles and k-step, four 1-KiB fragments: hi(tile 0), hi(tile 1), lo(tile 0), lo(tile 1); six MFMAs
// consume them.  Accumulators convert in place into the next layer's (hi, lo) k-step.
// The positional encoding is the reference's own expression, sin / cos of x 2^f in fp32 (accurate libm
// routines, like mlp_fp32.hip), generated in the B-operand slot order (program.h gen16_col).
#include <hip/hip_runtime.h>
#include <utility>

#include "kernels.h"
#include "launch_util.h"
#include "pipeline.h"
#include "program.h"
#include "split.h"

namespace na {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;

// Pair of 16-row output tiles T, T+1 over K1 k-steps of x1 and K2 of x2; F0 = first fragment (split numbering).
// acc[u][0] = hi x hi sums of tile u (starts at the bias), acc[u][1] = the two cross terms (scaled by 2^11).
template <int F0, int T, int K1, int K2, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tile_pair_split(C &c, const bf16x8 *x1, const bf16x8 *x2, f32x4 (&acc)[2][2]) {
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(c.bias_half + T * 16);
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(c.bias_half + (T + 1) * 16);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        acc[0][0] = b0; acc[0][1] = z; acc[1][0] = b1; acc[1][1] = z;
    }
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

// A single 16-row tile (sigma head, rgb head, output_linear): fragments hi, lo per k-step.
template <int F0, int T, int K1, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tile_single_split(C &c, const bf16x8 *x1, f32x4 &out) {
    f32x4 ah = *reinterpret_cast<const f32x4 *>(c.bias_half + T * 16);
    f32x4 al = {0.f, 0.f, 0.f, 0.f};
    static_for<K1>([&](auto k_) {
        constexpr int k = k_, n = F0 + 2 * k;
        const bf16x8 wh = take<n, NB, NFRAGS>(c);
        const bf16x8 wl = take<n + 1, NB, NFRAGS>(c);
        ah = MFMAH(wh, x1[2 * k], ah);
        al = MFMAH(wh, x1[2 * k + 1], al);
        al = MFMAH(wl, x1[2 * k], al);
        sched_step_split<C, 2, 3>();
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = __builtin_fmaf(al[r], SPLIT_INV, ah[r]);
}

// accumulators of a tile pair -> the (hi, lo) fragments of one k-step of the next layer: elements 0-3 from the
// even tile, 4-7 from the odd tile (program.h acc16_col).
template <bool RELU>
__device__ __forceinline__ void pack_pair_split(const f32x4 (&acc)[2][2], bf16x8 &yh, bf16x8 &yl) {
    f16x8 h, l;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            float v0 = __builtin_fmaf(acc[u][1][r], SPLIT_INV, acc[u][0][r]);
            float v1 = __builtin_fmaf(acc[u][1][r + 1], SPLIT_INV, acc[u][0][r + 1]);
            if (RELU) { v0 = relu_bits(v0); v1 = relu_bits(v1); }
            f16x2 a, b;
            split_f16_pair(v0, v1, a, b);
            h[4 * u + r] = a[0]; h[4 * u + r + 1] = a[1];
            l[4 * u + r] = b[0]; l[4 * u + r + 1] = b[1];
        }
    yh = __builtin_bit_cast(bf16x8, h);
    yl = __builtin_bit_cast(bf16x8, l);
}

// A hidden layer of NPAIR tile pairs -> y[2 * NPAIR] (k-step p: y[2p] = hi, y[2p+1] = lo).
template <int F0, int T0, int NPAIR, int K1, int K2, bool RELU, int NB, int NFRAGS, class C>
__device__ __forceinline__ void layer_split(C &c, const bf16x8 *x1, const bf16x8 *x2, bf16x8 *y) {
    static_for<NPAIR>([&](auto p_) {
        constexpr int p = p_;
        f32x4 acc[2][2];
        tile_pair_split<F0 + p * 4 * (K1 + K2), T0 + 2 * p, K1, K2, NB, NFRAGS>(c, x1, x2, acc);
        pack_pair_split<RELU>(acc, y[2 * p], y[2 * p + 1]);
    });
}

// The reference embedding (nerf.py:32-41) in the slot order of program.h gen16_col: lane quarter q = 2h + b holds, in
// slot i = 8 ks + j, sin (h = 0) or cos (h = 1) of x[i % 3] * 2^(2 (i / 3) + b) while i < gen16_ntrig(L, b), then raw
// coordinates.  Accurate sinf / cosf of the fp32 product (2^f is exact), as Embedder.embed evaluates it.
template <int L, int K>
__device__ __forceinline__ void encode_split(const float (&x)[3], int h, int b, bf16x8 *out /* [2 * K]: hi, lo per k-step */) {
    constexpr int N_EVEN = gen16_ntrig(L, 0), N_ODD = gen16_ntrig(L, 1);
    auto coord = [&](int mc) { return mc == 0 ? x[0] : mc == 1 ? x[1] : mc == 2 ? x[2] : 0.0f; };
    const float s0 = b ? 2.0f : 1.0f;
    static_for<K>([&](auto ks_) {
        constexpr int ks = ks_;
        f16x8 hi, lo;
        static_for<8>([&](auto j_) {
            constexpr int j = j_, i = 8 * ks + j;
            float v;
            constexpr bool trig_even = i < N_EVEN, trig_odd = i < N_ODD;
            float t = 0.0f;
            if constexpr (trig_even || trig_odd) {
                const float arg = x[i % 3] * (__builtin_ldexpf(1.0f, 2 * (i / 3)) * s0);
                float sv, cv;
                sincosf(arg, &sv, &cv);
                t = h ? cv : sv;
            }
            const float m_even = trig_even ? t : (h ? coord(gen16_misc(L, 1, 0, i - N_EVEN)) : coord(gen16_misc(L, 0, 0, i - N_EVEN)));
            const float m_odd = trig_odd ? t : (h ? coord(gen16_misc(L, 1, 1, i - N_ODD)) : coord(gen16_misc(L, 0, 1, i - N_ODD)));
            v = b ? m_odd : m_even;
            _Float16 a, c2;
            split_f16(v, a, c2);
            hi[j] = a;
            lo[j] = c2;
        });
        out[2 * ks] = __builtin_bit_cast(bf16x8, hi);
        out[2 * ks + 1] = __builtin_bit_cast(bf16x8, lo);
    });
}

// Training forward: the (hi, lo) fragments of NK k-steps as they are, into a plane of hi rows and a plane of lo rows of
// ROW fp16 values per point, slot-major like the bf16 training arrays (kernels.h).  Unconditional: rows exist for a
// workgroup's padding points (pad_points).
template <int NK, int ROW>
__device__ __forceinline__ void save_frags_split(uint16_t *hi, uint16_t *lo, const bf16x8 *y, int64_t p, int q) {
    static_for<NK>([&](auto k_) {
        constexpr int k = k_;
        *reinterpret_cast<bf16x8 *>(hi + p * ROW + k * 32 + q * 8) = y[2 * k];
        *reinterpret_cast<bf16x8 *>(lo + p * ROW + k * 32 + q * 8) = y[2 * k + 1];
    });
}

// "activation > 0" bits of a post-ReLU layer in the bit-row layout of mlp_bf16_s16.hip save_bits (dword k/4 of the lane's
// NK/4 dwords: bit 4 (k%4) + i = element 2i, bit 16 + 4 (k%4) + i = element 2i + 1 of k-step k).  An activation is
// positive when either half of its pair is non-zero (a value below fp16's smallest denormal lives in lo alone).
template <int NK>
__device__ __forceinline__ void save_bits_split(uint8_t *base, const bf16x8 *y, int64_t p, int q) {
    static_assert(NK == 4 || NK == 8, "one or two dwords of mask bits");
    unsigned w[NK / 4];
#pragma unroll
    for (int i = 0; i < NK / 4; ++i) w[i] = 0;
    static_for<NK>([&](auto k_) {
        constexpr int k = k_;
        const u32x4s vh = __builtin_bit_cast(u32x4s, y[2 * k]), vl = __builtin_bit_cast(u32x4s, y[2 * k + 1]);
        unsigned t = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned m;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(vh[i] | vl[i]), "v"(0x00010001u));   // 1 per non-zero half
            t |= m << i;
        }
        w[k / 4] |= t << (4 * (k % 4));
    });
    uint8_t *dst = base + p * (4 * NK) + q * NK;
    if constexpr (NK == 8) {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        u32x2 o = {w[0], w[1]};
        *reinterpret_cast<u32x2 *>(dst) = o;
    } else {
        *reinterpret_cast<unsigned *>(dst) = w[0];
    }
}

// Fragment offsets of the split stream: every region of the s16 layout (mlp_bf16_s16.hip Layout16) twice as long.
template <int LX, int LD, bool VD>
struct LayoutSplit {
    static constexpr int KE = gen16_ksteps(LX);
    static constexpr int KD = VD ? gen16_ksteps(LD) : 0;
    static constexpr int F_L0 = 0;
    static constexpr int F_L1 = F_L0 + 32 * KE;                 // 8 pairs x KE k-steps x 4 fragments
    static constexpr int F_L5 = F_L1 + 4 * 256;
    static constexpr int F_L6 = F_L5 + 32 * (KE + 8);
    static constexpr int F_HEAD = F_L6 + 2 * 256;
    static constexpr int F_FEAT = F_HEAD;
    static constexpr int F_ALPHA = F_FEAT + 256;
    static constexpr int F_VIEWS = F_ALPHA + 16;
    static constexpr int F_RGB = F_VIEWS + 16 * (8 + KD);
    static constexpr int F_END = VD ? F_RGB + 8 : F_HEAD + 16;
    static constexpr int N_TILES = VD ? 128 + 16 + 1 + 8 + 1 : 128 + 1;
};

template <int LX, int LD, bool VD, class C, bool SAVE = false>
__global__ __launch_bounds__(C::WAVES * 64, 2) void mlp_split_kernel(MlpArgs a) {
    constexpr int WG_THREADS = C::WAVES * 64, WG_POINTS = C::WAVES * 16;
    using Lay = LayoutSplit<LX, LD, VD>;
    constexpr int KE = Lay::KE, KD = Lay::KD, NF = Lay::F_END, NB = (NF + C::BF - 1) / C::BF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *bias_lds = reinterpret_cast<float *>(smem + C::RING_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = lane >> 4;
    C c;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.lag = __builtin_amdgcn_readfirstlane(c.wave >= C::WAVES / 2 ? 1 : 0);
    c.phase = __builtin_amdgcn_readfirstlane(C::N_PHASES == 4 ? ((c.wave + 2 * (c.wave >> 2)) & 3) : (c.wave >= C::WAVES / 2 ? 2 : 0));
    c.gstream = reinterpret_cast<const char *>(a.stream_split) + lane * 16;
    c.ring_lane = smem + lane * 16;
    c.ring_u32 = (uint32_t)(uintptr_t)smem;
    c.bias_half = bias_lds + q * 4;          // this lane's 4 rows of every 16-row tile
    if constexpr ((C::OPT & 64) != 0) {
        c.rsrc = make_rsrc(a.stream_split, (unsigned)(NB * C::BLOCK_BYTES));
        c.rsrc_next = c.rsrc;
        c.lane16 = lane * 16;
        c.wave_off = c.wave * C::PIECES * 1024;
    }
    for (int i = tid; i < Lay::N_TILES * 16; i += WG_THREADS) bias_lds[i] = a.bias_s16[i];

    // one workgroup per CU walks the 128-point tiles blockIdx.x, blockIdx.x + gridDim.x, ...
    const int64_t n_point_tiles = (a.P + WG_POINTS - 1) / WG_POINTS;
    // (a.tile_ctr != NULL: every tile after the first is a ticket from the launch's counter -- the dies do not hold the same
    // clock at the power cap, mlp_bf16_s16.hip mlp_bf16_s16p_kernel; the atomic is taken at the top of a tile and consumed
    // behind the tile's own end-of-tile drain, so it costs no wait)
    lds_u32_t *ticket_lds = (lds_u32_t *)(uintptr_t)(uint32_t)(uintptr_t)(bias_lds + Lay::N_TILES * 16);
    const bool dynamic = a.tile_ctr != nullptr;
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < n_point_tiles;) {
    unsigned ticket = 0;
    if (dynamic && tid == 0) ticket = atomicAdd(a.tile_ctr, 1u);
    // opaque per-iteration copy of the stream pointer (the DMA source addresses must not be hoisted out of the loop)
    asm volatile("" : "+v"(c.gstream));
    pipeline_prologue<NB>(c);

    // ---- this lane's point: column lane & 15 of the wave's 16-point tile (all four lane quarters hold the same point)
    bf16x8 E[KE * 2];
    bf16x8 Dv[(VD ? KD : 1) * 2];
    const int64_t p = tile * WG_POINTS + c.wave * 16 + (lane & 15);
    const bool valid = p < a.P;
    {
        const int64_t pc = valid ? p : a.P - 1;
        const int64_t ray = (int64_t)((uint32_t)pc / (uint32_t)a.S);   // P < 2^31 (checked at launch)
        float xs[3], dv[3] = {0.f, 0.f, 0.f};
        if (a.pts) {
            xs[0] = a.pts[3 * pc + 0]; xs[1] = a.pts[3 * pc + 1]; xs[2] = a.pts[3 * pc + 2];
        } else {
            const float *r = a.rays + ray * a.ray_stride;
            const float z = a.z_vals[pc];
            xs[0] = mul_then_add(r[3], z, r[0]);           // pts = o + d z, rounded like the reference's two ops
            xs[1] = mul_then_add(r[4], z, r[1]);
            xs[2] = mul_then_add(r[5], z, r[2]);
        }
        if constexpr (VD) {
            const float *d = a.viewdirs + ray * a.vd_stride;
            dv[0] = d[0]; dv[1] = d[1]; dv[2] = d[2];
        }
        encode_split<LX, KE>(xs, q >> 1, q & 1, E);
        if constexpr (VD) encode_split<LD, KD>(dv, q >> 1, q & 1, Dv);
    }

    if constexpr (C::PHASE > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // bias table stores, before the barrier publishes them
        block_sync<-1, NB>(c);                                 // publishes block 0
        static_for<C::LA>([&](auto i_) { constexpr int i = i_; c.q[i] = ring_frag<i>(c); });
    }
    // training forward: every layer's (hi, lo) output also goes to HBM for the backward pass (mlp_bwd_split.hip, backward.hip)
    const int64_t HS = pad_points(a.P) * 256;                   // one saved hidden layer (one plane)
    const int64_t BS = pad_points(a.P) * 32;                    // one layer of mask-bit rows
    if constexpr (SAVE) {
        // a three-k-step encoding (multires 15) is saved in rows of 128 slots, the fourth k-step zero: its weight-gradient
        // product then has power-of-two rows (backward.hip enc_row_slots)
        constexpr int ROW_E = KE == 3 ? 128 : 32 * KE;
        save_frags_split<KE, ROW_E>(a.sv_e, a.sv_e_lo, E, p, q);
        if constexpr (KE == 3) {
            const bf16x8 zero2[2] = {};
            save_frags_split<1, ROW_E>(a.sv_e + 96, a.sv_e_lo + 96, zero2, p, q);
        }
        if constexpr (VD) save_frags_split<KD, 32 * KD>(a.sv_d, a.sv_d_lo, Dv, p, q);
    }
    auto save_h = [&](auto l_, const bf16x8 *y) {
        constexpr int l = l_;
        if constexpr (SAVE) {
            save_frags_split<8, 256>(a.sv_h + l * HS, a.sv_h_lo + l * HS, y, p, q);
            save_bits_split<8>(a.sv_bits + l * BS, y, p, q);
        }
    };
    using std::integral_constant;
    bf16x8 A[16], B[16];
    layer_split<Lay::F_L0, 0, 8, KE, 0, true, NB, NF>(c, E, E, A);
    save_h(integral_constant<int, 0>{}, A);
    layer_split<Lay::F_L1 + 0 * 256, 16, 8, 8, 0, true, NB, NF>(c, A, A, B);
    save_h(integral_constant<int, 1>{}, B);
    layer_split<Lay::F_L1 + 1 * 256, 32, 8, 8, 0, true, NB, NF>(c, B, B, A);
    save_h(integral_constant<int, 2>{}, A);
    layer_split<Lay::F_L1 + 2 * 256, 48, 8, 8, 0, true, NB, NF>(c, A, A, B);
    save_h(integral_constant<int, 3>{}, B);
    layer_split<Lay::F_L1 + 3 * 256, 64, 8, 8, 0, true, NB, NF>(c, B, B, A);
    save_h(integral_constant<int, 4>{}, A);
    layer_split<Lay::F_L5, 80, 8, KE, 8, true, NB, NF>(c, E, A, B);            // skip: [input_pts | h]
    save_h(integral_constant<int, 5>{}, B);
    layer_split<Lay::F_L6, 96, 8, 8, 0, true, NB, NF>(c, B, B, A);
    save_h(integral_constant<int, 6>{}, A);
    layer_split<Lay::F_L6 + 256, 112, 8, 8, 0, true, NB, NF>(c, A, A, B);      // h7 in B
    save_h(integral_constant<int, 7>{}, B);

    if constexpr (VD) {
        layer_split<Lay::F_FEAT, 128, 8, 8, 0, false, NB, NF>(c, B, B, A);     // feature (no activation)
        if constexpr (SAVE) save_frags_split<8, 256>(a.sv_feat, a.sv_feat_lo, A, p, q);
        f32x4 alpha, rgb;
        tile_single_split<Lay::F_ALPHA, 144, 8, NB, NF>(c, B, alpha);          // row 0 = sigma
        layer_split<Lay::F_VIEWS, 145, 4, 8, KD, true, NB, NF>(c, A, Dv, B);   // views_linears.0 (128 rows)
        if constexpr (SAVE) { save_frags_split<4, 128>(a.sv_hv, a.sv_hv_lo, B, p, q); save_bits_split<4>(a.sv_bits + 8 * BS, B, p, q); }
        tile_single_split<Lay::F_RGB, 153, 4, NB, NF>(c, B, rgb);              // rows 0..2
        if (valid && q == 0) {
            f32x4 o = {rgb[0], rgb[1], rgb[2], alpha[0]};
            *reinterpret_cast<f32x4 *>(a.out + 4 * p) = o;
        }
    } else {
        f32x4 o;
        tile_single_split<Lay::F_HEAD, 128, 8, NB, NF>(c, B, o);
        if (valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q + r;
                if (row < a.out_ch) a.out[(int64_t)a.out_ch * p + row] = o[r];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the tile (or the workgroup)
    if (dynamic && tid == 0) *ticket_lds = gridDim.x + ticket;
    __syncthreads();                                           // every wave is done with the ring before it is refilled
    tile = dynamic ? (int64_t)*ticket_lds : tile + gridDim.x;
    }
    if (dynamic && tid == 0) {                                 // the last workgroup out leaves the pair zero for its next launch
        if (atomicAdd(a.tile_ctr + 1, 1u) == gridDim.x - 1) { a.tile_ctr[0] = 0u; a.tile_ctr[1] = 0u; }
    }
}

template <int LX, int LD, bool VD, class C, bool SAVE = false>
static int launch_split(const MlpArgs &a, int n_frags_used, int n_tiles, hipStream_t s) {
    constexpr int WG_THREADS = C::WAVES * 64, WG_POINTS = C::WAVES * 16;
    using Lay = LayoutSplit<LX, LD, VD>;
    if (n_frags_used != Lay::F_END || n_tiles != Lay::N_TILES) return NERF_AMD_EINVAL;
    const size_t lds = C::RING_BYTES + (size_t)Lay::N_TILES * 16 * sizeof(float) + 16;      // + the ticket word
    static DynamicLdsOptIn opt_in;         // per kernel instantiation, tracks every device (launch_util.h)
    if (opt_in.ensure(reinterpret_cast<const void *>(mlp_split_kernel<LX, LD, VD, C, SAVE>), lds) != hipSuccess) return NERF_AMD_EHIP;
    int64_t groups = (a.P + WG_POINTS - 1) / WG_POINTS;
    if (groups <= 0) return NERF_AMD_OK;
    if (a.P >= (int64_t)1 << 31) return NERF_AMD_EINVAL;
    const int n_wg = device_cu_count();      // one workgroup per CU walks the tiles
    const bool deal = groups > 2 * (int64_t)n_wg && g_variant != 42;      // dealt by ticket (A/B 42: blockIdx + k gridDim)
    if (groups > n_wg) groups = n_wg;
    MlpArgs a2 = a;
    a2.tile_ctr = tile_counter_for(deal, s);
    hipLaunchKernelGGL((mlp_split_kernel<LX, LD, VD, C, SAVE>), dim3((unsigned)groups), dim3(WG_THREADS), lds, s, a2);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// the pipeline shape of the bf16 kernel (mlp_bf16_s16.hip Cfg16): 64-KiB ring of 16-fragment blocks, mid-block sync,
// 4-deep read-ahead pinned in front of its MFMAs, split DMA issue, buffer-addressed ring DMA
using CfgSplit = Ctx<8, 16, 4, 8, 4, 0, 1, 4 + 8 + 64>;

bool mlp_split_supported(int multires, int multires_views, int use_viewdirs, int out_ch) {
    if (use_viewdirs) return (multires == 10 && multires_views == 4) || (multires == 15 && multires_views == 6);
    return out_ch <= 16 && (multires == 10 || multires == 15);
}

int launch_mlp_split(const MlpArgs &a, int multires, int multires_views, int use_viewdirs, int n_frags_used, int n_tiles,
                     hipStream_t s) {
    if (use_viewdirs) {
        if (multires == 10 && multires_views == 4) return launch_split<10, 4, true, CfgSplit>(a, n_frags_used, n_tiles, s);
        if (multires == 15 && multires_views == 6) return launch_split<15, 6, true, CfgSplit>(a, n_frags_used, n_tiles, s);
    } else if (a.out_ch <= 16) {
        if (multires == 10) return launch_split<10, 0, false, CfgSplit>(a, n_frags_used, n_tiles, s);
        if (multires == 15) return launch_split<15, 0, false, CfgSplit>(a, n_frags_used, n_tiles, s);
    }
    return NERF_AMD_EUNSUPPORTED;
}

// The training forward of the split-precision mode: the same kernel, every layer's (hi, lo) output saved (kernels.h).
using CfgSplitSave = Ctx<8, 16, 4, 8, 2>;     // the plain pipeline shape, like the bf16 training forward

int launch_mlp_split_save(const MlpArgs &a, int multires, int multires_views, int use_viewdirs, int n_frags_used, int n_tiles,
                          hipStream_t s) {
    if (use_viewdirs) {
        if (multires == 10 && multires_views == 4) return launch_split<10, 4, true, CfgSplitSave, true>(a, n_frags_used, n_tiles, s);
        if (multires == 15 && multires_views == 6) return launch_split<15, 6, true, CfgSplitSave, true>(a, n_frags_used, n_tiles, s);
    } else if (a.out_ch <= 16) {
        if (multires == 10) return launch_split<10, 0, false, CfgSplitSave, true>(a, n_frags_used, n_tiles, s);
        if (multires == 15) return launch_split<15, 0, false, CfgSplitSave, true>(a, n_frags_used, n_tiles, s);
    }
    return NERF_AMD_EUNSUPPORTED;
}

}  // namespace na
