// mlp_bf16_s16.hip -- the fused positional-encoding + 8x256 NeRF MLP on
// v_mfma_f32_16x16x32_bf16 (second generation of mlp_bf16.hip; same algorithm,
// same pipeline, different MFMA shape).
//
// Why a second shape: this kernel is power-bound, not issue-bound.  On MI355X the
// 16x16x32 form sustains ~15 % more FLOP/s than 32x32x16 at equal cycles per FLOP
// (tools/micro/mfma_shape.hip: 2.1 vs 1.84 PFLOP/s from registers, 1.70 vs 1.50 with
// one LDS fragment read per two MFMAs), so the same work finishes sooner.
//
// Layout (program.h, "s16"): a wave owns 32 points as two 16-column tiles c = 0,1.
// Output features come in 16-row tiles; two consecutive tiles (a pair) share the B
// operands, and their accumulators -- lane (col, q) holds rows 4q..4q+3 -- convert in
// place into the next layer's 32-deep k-step: elements 0-3 from the even tile, 4-7 from
// the odd tile.  Each 1-KiB A fragment (16 rows x 32 k) feeds two MFMAs (c = 0,1).
#include <hip/hip_runtime.h>
#include <utility>

#include "kernels.h"
#include "launch_util.h"
#include "pipeline.h"
#include "program.h"

// Experiment scaffolding is not part of this translation unit: A/B and scratch builds (make EXTRA=-DNERF_AMD_EXPERIMENTS ...)
// pull their variant tables and hooks from tools/experiments/ (README.md there); the shipping build sees empty hooks.
#ifdef NERF_AMD_EXPERIMENTS
#include "../../tools/experiments/save_variants.inc"
#endif
#ifndef NA_EXPERIMENT_SAVE_FRAGS_HOOK
#define NA_EXPERIMENT_SAVE_FRAGS_HOOK
#endif

namespace na {

#ifdef NERF_AMD_STAMPS
// Diagnostic build only (cdna_hip_programming.md section 7, in-kernel stamps): per wave, the shader cycles spent
// in the three segments of a tile -- [loop top .. first weight block published], [.. last MFMA issued],
// [.. end-of-tile drain] -- summed over the wave's tiles into a buffer nothing else reads.
unsigned long long *g_stamp_buf = nullptr;
#define STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define STAMP(var)
#endif

#define MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0)

// Scheduling pipeline of one k-step (OPT & 4): NR LDS reads (the read-ahead of the fragments LA ahead), then NM
// MFMAs.  Without it the machine scheduler sinks half of the fragment reads to just in front of their first MFMA
// (tools/isa_readahead.py), which exposes the full LDS latency every two MFMAs.
template <class C, int NR, int NM>
__device__ __forceinline__ void sched_step() {
    if constexpr ((C::OPT & 4) != 0) {
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);   // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);   // MFMA
    }
}

// Pair of 16-row output tiles T, T+1 over K1 k-steps of x1 and K2 of x2.
// Activation fragment of k-step k, column tile c is x[2*k + c].  acc[u][c]: tile u, column tile c.
struct NoHook {
    template <class P, class K> __device__ __forceinline__ void operator()(P, K) const {}
};

// hook(p, k): called after the MFMAs of k-step k (0 .. K1+K2-1) of the pair; the pipelined kernel hangs the next
// tile's coordinate loads and positional encoding on it, a piece at a time.
template <int F0, int T, int K1, int K2, int NB, int NFRAGS, class C, class P = std::integral_constant<int, 0>, class H = NoHook>
__device__ __forceinline__ void tile_pair(C &c, const bf16x8 *x1, const bf16x8 *x2, f32x4 (&acc)[2][2], P p_ = P{}, H &&hook = H{}) {
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(c.bias_half + T * 16);
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(c.bias_half + (T + 1) * 16);
        acc[0][0] = b0; acc[0][1] = b0; acc[1][0] = b1; acc[1][1] = b1;
    }
    static_for<K1>([&](auto k_) {
        constexpr int k = k_, n = F0 + 2 * k;
        const bf16x8 w0 = take<n, NB, NFRAGS>(c);
        const bf16x8 w1 = take<n + 1, NB, NFRAGS>(c);
        acc[0][0] = MFMA16(w0, x1[2 * k], acc[0][0]);
        acc[0][1] = MFMA16(w0, x1[2 * k + 1], acc[0][1]);
        acc[1][0] = MFMA16(w1, x1[2 * k], acc[1][0]);
        acc[1][1] = MFMA16(w1, x1[2 * k + 1], acc[1][1]);
        sched_step<C, 2, 4>();
        hook(p_, std::integral_constant<int, k>{});
    });
    static_for<K2>([&](auto k_) {
        constexpr int k = k_, n = F0 + 2 * K1 + 2 * k;
        const bf16x8 w0 = take<n, NB, NFRAGS>(c);
        const bf16x8 w1 = take<n + 1, NB, NFRAGS>(c);
        acc[0][0] = MFMA16(w0, x2[2 * k], acc[0][0]);
        acc[0][1] = MFMA16(w0, x2[2 * k + 1], acc[0][1]);
        acc[1][0] = MFMA16(w1, x2[2 * k], acc[1][0]);
        acc[1][1] = MFMA16(w1, x2[2 * k + 1], acc[1][1]);
        sched_step<C, 2, 4>();
        hook(p_, std::integral_constant<int, K1 + k>{});
    });
}

// A single 16-row tile (the 1-row sigma head, the 3-row rgb head, output_linear).
template <int F0, int T, int K1, int NB, int NFRAGS, class C>
__device__ __forceinline__ void tile_single(C &c, const bf16x8 *x1, f32x4 (&acc)[2]) {
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(c.bias_half + T * 16);
    acc[0] = b0; acc[1] = b0;
    static_for<K1>([&](auto k_) {
        constexpr int k = k_, n = F0 + k;
        const bf16x8 w0 = take<n, NB, NFRAGS>(c);
        acc[0] = MFMA16(w0, x1[2 * k], acc[0]);
        acc[1] = MFMA16(w0, x1[2 * k + 1], acc[1]);
        sched_step<C, 1, 2>();
    });
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// bias/MFMA result -> next layer's B fragment.  ReLU is applied after the bf16 rounding, on the
// packed halves: one packed signed-16-bit max per register (negative bf16 = negative int16);
// round-to-nearest is monotonic and sign-symmetric, so relu(round(x)) == round(relu(x)).
template <bool RELU, bool PKRELU = true>
__device__ __forceinline__ bf16x8 pack_pair(const f32x4 &even, const f32x4 &odd) {
    bf16x8 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = even[r], b = odd[r];
        if (RELU && !PKRELU) { a = relu_bits(a); b = relu_bits(b); }
        y[r] = (__bf16)a;
        y[4 + r] = (__bf16)b;
    }
    if (RELU && PKRELU) {
        u32x4 v = __builtin_bit_cast(u32x4, y);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned d = v[i];
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(d) : "v"(d));
            v[i] = d;
        }
        y = __builtin_bit_cast(bf16x8, v);
    }
    return y;
}

// A hidden layer of NPAIR tile pairs -> y[2 * NPAIR] (k-step p, column tile c at y[2p + c]).
template <int F0, int T0, int NPAIR, int K1, int K2, bool RELU, int NB, int NFRAGS, class C, class H = NoHook>
__device__ __forceinline__ void layer16(C &c, const bf16x8 *x1, const bf16x8 *x2, bf16x8 *y, H &&hook = H{}) {
    static_for<NPAIR>([&](auto p_) {
        constexpr int p = p_;
        f32x4 acc[2][2];
        tile_pair<F0 + p * 2 * (K1 + K2), T0 + 2 * p, K1, K2, NB, NFRAGS>(c, x1, x2, acc, p_, hook);
        y[2 * p] = pack_pair<RELU, !(C::OPT & 1)>(acc[0][0], acc[1][0]);
        y[2 * p + 1] = pack_pair<RELU, !(C::OPT & 1)>(acc[0][1], acc[1][1]);
    });
}

// Positional encoding into FRAG_GEN16 layout (program.h: gen16_col).  h = sin/cos family,
// b = frequency parity of this lane quarter.  th + tl = x / (2 pi) as an exact fp32 pair;
// multiplying by 4 and v_fract are exact, v_sin_f32 takes revolutions.
// The work comes in pieces -- begin, NSTEP x step, finish -- so the pipelined kernel can spread the encoding of its
// NEXT tile over the k-steps of the current one; encode16 runs them back to back.
template <int L, int K>
struct Enc16 {
    static constexpr int NSTEP = (L + 1) / 2, CAP = 8 * K;
    static constexpr int N_EVEN = gen16_ntrig(L, 0), N_ODD = gen16_ntrig(L, 1);
    static constexpr int N_COMMON = N_ODD < N_EVEN ? N_ODD : N_EVEN;     // slots that hold a trig value in every lane quarter
    static constexpr int N_TAIL = (N_EVEN > N_ODD ? N_EVEN : N_ODD) - N_COMMON;
    float x[3], ra[3], tl[3], phase;
    unsigned sgn[3];
    float tail[N_TAIL > 0 ? N_TAIL : 1];      // trig values of the slots that are raw coordinates in the other parity

    __device__ __forceinline__ void begin(float x0, float x1, float x2, int h, int b) {
        constexpr float INV2PI_HI = 0.15915494f;
        constexpr float INV2PI_LO = (float)(0.15915494309189535 - (double)INV2PI_HI);
        x[0] = x0; x[1] = x1; x[2] = x2;
        phase = h ? 0.25f : 0.0f;
        const float s0 = b ? 2.0f : 1.0f;
        // The reduction runs on |x|: v_fract of a negative number is x - floor(x) *rounded*, and that ulp would be
        // quadrupled with every step (3e-3 at 2^14, a bf16 quantum).  sin is odd and cos even, so the sin family
        // (h == 0) gets x's sign bit back at the end.
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ax = __builtin_fabsf(x[c]);
            sgn[c] = h ? 0u : (__builtin_bit_cast(unsigned, x[c]) & 0x80000000u);
            const float th = ax * INV2PI_HI;
            tl[c] = (__builtin_fmaf(ax, INV2PI_HI, -th) + ax * INV2PI_LO) * s0;
            ra[c] = __builtin_amdgcn_fractf(th * s0);
        }
    }
    // step S: the three trig values of slots 3S .. 3S+2, converted to bf16 and placed at once (a slot i lives in
    // out[(i / 8) * STRIDE][i % 8]) unless the slot's content depends on the lane's frequency parity (finish)
    template <int S, int STRIDE>
    __device__ __forceinline__ void step(bf16x8 *out) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            constexpr int dummy = 0; (void)dummy;
            const int i = 3 * S + c;
            if (i < CAP) {
                const float v = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, __builtin_amdgcn_sinf(ra[c] + (tl[c] + phase))) ^ sgn[c]);
                if (i < N_COMMON) out[(i / 8) * STRIDE][i % 8] = (__bf16)v;
                else if (i - N_COMMON < N_TAIL) tail[i - N_COMMON] = v;
            }
            ra[c] = __builtin_amdgcn_fractf(ra[c] * 4.0f);
            tl[c] *= 4.0f;
        }
    }
    template <int STRIDE>
    __device__ __forceinline__ void finish(int h, int b, bf16x8 *out) {
        // slots past a group's trig count hold raw coordinates (or nothing)
        auto coord = [&](int mc) { return mc == 0 ? x[0] : mc == 1 ? x[1] : mc == 2 ? x[2] : 0.0f; };
#pragma unroll
        for (int i = N_COMMON; i < CAP; ++i) {
            const float t = (i - N_COMMON < N_TAIL) ? tail[i - N_COMMON < N_TAIL ? i - N_COMMON : 0] : 0.0f;
            const float v_even = i < N_EVEN ? t : (h ? coord(gen16_misc(L, 1, 0, i - N_EVEN)) : coord(gen16_misc(L, 0, 0, i - N_EVEN)));
            const float v_odd = i < N_ODD ? t : (h ? coord(gen16_misc(L, 1, 1, i - N_ODD)) : coord(gen16_misc(L, 0, 1, i - N_ODD)));
            out[(i / 8) * STRIDE][i % 8] = (__bf16)(b ? v_odd : v_even);
        }
    }
};

template <int L, int K, int STRIDE>
__device__ __forceinline__ void encode16(float x0, float x1, float x2, int h, int b, bf16x8 *out) {
    Enc16<L, K> e;
    e.begin(x0, x1, x2, h, b);
    static_for<Enc16<L, K>::NSTEP>([&](auto s_) { e.template step<s_, STRIDE>(out); });
    e.template finish<STRIDE>(h, b, out);
}

// Save NK k-steps of fragments (y[2*ks + cc]) as slot-major bf16 rows of ROW elements.  Unconditional:
// the training arrays have rows for a workgroup's padding points (kernels.h pad_points).
template <int NK, int ROW>
__device__ __forceinline__ void save_frags(uint16_t *base, const bf16x8 *y, const int64_t (&pidx)[2], int q) {
    NA_EXPERIMENT_SAVE_FRAGS_HOOK              // empty in the shipping build (tools/experiments/README.md)
    static_for<NK>([&](auto k_) {
        constexpr int k = k_;
        static_for<2>([&](auto cc_) {
            constexpr int cc = cc_;
            *reinterpret_cast<bf16x8 *>(base + pidx[cc] * ROW + k * 32 + q * 8) = y[2 * k + cc];
        });
    });
}

// "activation > 0" bits of a post-ReLU layer for the backward's ReLU masks, 8 bits per fragment and
// lane: dword k/4 of the lane's NK/4 dwords holds, for k-step k, bit 4*(k%4) + i = element 2i and bit
// 16 + 4*(k%4) + i = element 2i + 1 (i = 0..3).  A lane quarter's dwords are contiguous at
// row*4*NK + q*NK, so the backward fetches a whole layer's masks with one load per point.
template <int NK>
__device__ __forceinline__ void save_bits(uint8_t *base, const bf16x8 *y, const int64_t (&pidx)[2], int q) {
    static_assert(NK == 4 || NK == 8, "one or two dwords of mask bits");
    static_for<2>([&](auto cc_) {
        constexpr int cc = cc_;
        unsigned w[NK / 4];
#pragma unroll
        for (int i = 0; i < NK / 4; ++i) w[i] = 0;
        static_for<NK>([&](auto k_) {
            constexpr int k = k_;
            const u32x4 v = __builtin_bit_cast(u32x4, y[2 * k + cc]);
            unsigned t = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned m;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(v[i]), "v"(0x00010001u));   // 1 per non-zero half
                t |= m << i;
            }
            w[k / 4] |= t << (4 * (k % 4));
        });
        uint8_t *dst = base + pidx[cc] * (4 * NK) + q * NK;
        if constexpr (NK == 8) {
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            u32x2 o = {w[0], w[1]};
            *reinterpret_cast<u32x2 *>(dst) = o;
        } else {
            *reinterpret_cast<unsigned *>(dst) = w[0];
        }
    });
}

template <int LX, int LD, bool VD>
struct Layout16 {
    static constexpr int KE = gen16_ksteps(LX);
    static constexpr int KD = VD ? gen16_ksteps(LD) : 0;
    static constexpr int F_L0 = 0;
    static constexpr int F_L1 = F_L0 + 16 * KE;
    static constexpr int F_L5 = F_L1 + 4 * 128;
    static constexpr int F_L6 = F_L5 + 16 * (KE + 8);
    static constexpr int F_HEAD = F_L6 + 2 * 128;
    static constexpr int F_FEAT = F_HEAD;
    static constexpr int F_ALPHA = F_FEAT + 128;
    static constexpr int F_VIEWS = F_ALPHA + 8;
    static constexpr int F_RGB = F_VIEWS + 8 * (8 + KD);
    static constexpr int F_END = VD ? F_RGB + 4 : F_HEAD + 8;
    static constexpr int N_TILES = VD ? 128 + 16 + 1 + 8 + 1 : 128 + 1;
};

template <int LX, int LD, bool VD, class C, bool SAVE = false>
__global__ __launch_bounds__(C::WAVES * 64, 2) void mlp_bf16_s16_kernel(MlpArgs a) {
    constexpr int WG_THREADS = C::WAVES * 64, WG_POINTS = C::WAVES * 32;
    using Lay = Layout16<LX, LD, VD>;
    constexpr int KE = Lay::KE, KD = Lay::KD, NF = Lay::F_END, NB = (NF + C::BF - 1) / C::BF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *bias_lds = reinterpret_cast<float *>(smem + C::RING_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = lane >> 4;
    C c;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.lag = __builtin_amdgcn_readfirstlane(c.wave >= C::WAVES / 2 ? 1 : 0);
    // SPLIT_DMA issue phase (an SGPR: the DMA asm branches on it): two phases 0 / 2 by wave half; four phases
    // 0 1 2 3 2 3 0 1 over the waves, so that SIMD partners (w, w + 4) are always half a block apart
    c.phase = __builtin_amdgcn_readfirstlane(C::N_PHASES == 4 ? ((c.wave + 2 * (c.wave >> 2)) & 3) : (c.wave >= C::WAVES / 2 ? 2 : 0));
    if constexpr ((C::OPT & 16) != 0) {      // static priority for the second-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
        if (c.lag) __builtin_amdgcn_s_setprio(1);
    }
    c.gstream = reinterpret_cast<const char *>(a.stream_s16) + lane * 16;
    c.ring_lane = smem + lane * 16;
    c.ring_u32 = (uint32_t)(uintptr_t)smem;
    c.bias_half = bias_lds + q * 4;          // this lane's 4 rows of every 16-row tile
    if constexpr ((C::OPT & 64) != 0) {
        c.rsrc = make_rsrc(a.stream_s16, (unsigned)(NB * C::BLOCK_BYTES));
        c.rsrc_next = c.rsrc;
        c.lane16 = lane * 16;
        c.wave_off = c.wave * C::PIECES * 1024;
    }

    for (int i = tid; i < Lay::N_TILES * 16; i += WG_THREADS) bias_lds[i] = a.bias_s16[i];

    // A workgroup walks 256-point tiles blockIdx.x, blockIdx.x + gridDim.x, ...: with one workgroup per CU
    // (the launcher's choice for large P) the dispatch of a fresh workgroup per tile disappears.
    const int64_t n_point_tiles = (a.P + WG_POINTS - 1) / WG_POINTS;
    // (a.tile_ctr != NULL: every tile after the first is a ticket from the launch's counter, see mlp_bf16_s16p_kernel; here
    // the atomic is taken at the top of a tile and consumed behind the tile's own end-of-tile drain)
    lds_u32_t *ticket_lds = (lds_u32_t *)(uintptr_t)(uint32_t)(uintptr_t)(bias_lds + Lay::N_TILES * 16);
    const bool dynamic = a.tile_ctr != nullptr;
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < n_point_tiles;) {
    unsigned ticket = 0;
    if (dynamic && tid == 0) ticket = atomicAdd(a.tile_ctr, 1u);
    STAMP(t0);
    // opaque per-iteration copy of the stream pointer: otherwise the 148 DMA source addresses of the body are
    // loop-invariant, get hoisted in front of the loop and cost ~300 VGPRs
    asm volatile("" : "+v"(c.gstream));
    pipeline_prologue<NB>(c);

    // ---- this lane's two points: column tile cc, column lane&15 (all four lane quarters hold the same points).
    // All loads of both points are issued before anything is computed from them (one branch on the input mode,
    // one round trip to memory per tile instead of four: the encodings below are long enough that the compiler
    // does not move the second point's loads above them by itself).
    bf16x8 E[KE * 2];
    bf16x8 Dv[(VD ? KD : 1) * 2];
    int64_t pidx[2];
    bool valid[2];
    float xs[2][3], dv[2][3];
    {
        int64_t pc[2], ray[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int64_t p = tile * WG_POINTS + c.wave * 32 + cc * 16 + (lane & 15);
            pidx[cc] = p;
            valid[cc] = p < a.P;
            pc[cc] = valid[cc] ? p : a.P - 1;
            ray[cc] = (int64_t)((uint32_t)pc[cc] / (uint32_t)a.S);   // P < 2^31 (checked at launch)
        }
        auto load_dirs = [&]() {
            if constexpr (VD) {
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const float *d = a.viewdirs + ray[cc] * a.vd_stride;
                    dv[cc][0] = d[0]; dv[cc][1] = d[1]; dv[cc][2] = d[2];
                }
            }
        };
        if (a.pts) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                xs[cc][0] = a.pts[3 * pc[cc] + 0]; xs[cc][1] = a.pts[3 * pc[cc] + 1]; xs[cc][2] = a.pts[3 * pc[cc] + 2];
            }
            load_dirs();
        } else {
            float o[2][6], z[2];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const float *r = a.rays + ray[cc] * a.ray_stride;
#pragma unroll
                for (int k = 0; k < 6; ++k) o[cc][k] = r[k];
                z[cc] = a.z_vals[pc[cc]];
            }
            load_dirs();
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {                       // pts = o + d z, rounded like the reference's two ops
                xs[cc][0] = mul_then_add(o[cc][3], z[cc], o[cc][0]);
                xs[cc][1] = mul_then_add(o[cc][4], z[cc], o[cc][1]);
                xs[cc][2] = mul_then_add(o[cc][5], z[cc], o[cc][2]);
            }
        }
    }
    static_for<2>([&](auto cc_) {
        constexpr int cc = cc_;
        encode16<LX, KE, 2>(xs[cc][0], xs[cc][1], xs[cc][2], q >> 1, q & 1, E + cc);
        if constexpr (VD) encode16<LD, KD, 2>(dv[cc][0], dv[cc][1], dv[cc][2], q >> 1, q & 1, Dv + cc);
    });

    if constexpr (C::PHASE > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // bias table stores, before the barrier publishes them
        block_sync<-1, NB>(c);                                 // publishes block 0
        static_for<C::LA>([&](auto i_) { constexpr int i = i_; c.q[i] = ring_frag<i>(c); });
    }
    STAMP(t1);
    // training forward: every layer's output also goes to HBM for the backward pass.  (These stores
    // sit in the same vmcnt queue as the ring DMA, so the counted waits are conservative right after a
    // layer's burst; measured, a store ledger like the backward kernel's gains nothing here -- the
    // kernel is bound by its 5.3 KB of writes per point.)
    const int64_t HS = pad_points(a.P) * 256;                   // one saved hidden layer
    const int64_t BS = pad_points(a.P) * 32;                    // one layer of mask-bit rows
    if constexpr (SAVE) {
        // a three-k-step encoding (multires 15) is saved in rows of 128 slots, the fourth k-step zero: its weight-gradient
        // product then has power-of-two rows (backward.hip enc_row_slots)
        constexpr int ROW_E = KE == 3 ? 128 : 32 * KE;
        save_frags<KE, ROW_E>(a.sv_e, E, pidx, q);
        if constexpr (KE == 3) {
            const bf16x8 zero2[2] = {};
            save_frags<1, ROW_E>(a.sv_e + 96, zero2, pidx, q);
        }
        if constexpr (VD) save_frags<KD, 32 * KD>(a.sv_d, Dv, pidx, q);
    }
    bf16x8 A[16], B[16];
    layer16<Lay::F_L0, 0, 8, KE, 0, true, NB, NF>(c, E, E, A);
    if constexpr (SAVE) { save_frags<8, 256>(a.sv_h + 0 * HS, A, pidx, q); save_bits<8>(a.sv_bits + 0 * BS, A, pidx, q); }
    layer16<Lay::F_L1 + 0 * 128, 16, 8, 8, 0, true, NB, NF>(c, A, A, B);
    if constexpr (SAVE) { save_frags<8, 256>(a.sv_h + 1 * HS, B, pidx, q); save_bits<8>(a.sv_bits + 1 * BS, B, pidx, q); }
    layer16<Lay::F_L1 + 1 * 128, 32, 8, 8, 0, true, NB, NF>(c, B, B, A);
    if constexpr (SAVE) { save_frags<8, 256>(a.sv_h + 2 * HS, A, pidx, q); save_bits<8>(a.sv_bits + 2 * BS, A, pidx, q); }
    layer16<Lay::F_L1 + 2 * 128, 48, 8, 8, 0, true, NB, NF>(c, A, A, B);
    if constexpr (SAVE) { save_frags<8, 256>(a.sv_h + 3 * HS, B, pidx, q); save_bits<8>(a.sv_bits + 3 * BS, B, pidx, q); }
    layer16<Lay::F_L1 + 3 * 128, 64, 8, 8, 0, true, NB, NF>(c, B, B, A);
    if constexpr (SAVE) { save_frags<8, 256>(a.sv_h + 4 * HS, A, pidx, q); save_bits<8>(a.sv_bits + 4 * BS, A, pidx, q); }
    layer16<Lay::F_L5, 80, 8, KE, 8, true, NB, NF>(c, E, A, B);            // skip: [input_pts | h]
    if constexpr (SAVE) { save_frags<8, 256>(a.sv_h + 5 * HS, B, pidx, q); save_bits<8>(a.sv_bits + 5 * BS, B, pidx, q); }
    layer16<Lay::F_L6, 96, 8, 8, 0, true, NB, NF>(c, B, B, A);
    if constexpr (SAVE) { save_frags<8, 256>(a.sv_h + 6 * HS, A, pidx, q); save_bits<8>(a.sv_bits + 6 * BS, A, pidx, q); }
    layer16<Lay::F_L6 + 128, 112, 8, 8, 0, true, NB, NF>(c, A, A, B);      // h7 in B
    if constexpr (SAVE) { save_frags<8, 256>(a.sv_h + 7 * HS, B, pidx, q); save_bits<8>(a.sv_bits + 7 * BS, B, pidx, q); }

    if constexpr (VD) {
        layer16<Lay::F_FEAT, 128, 8, 8, 0, false, NB, NF>(c, B, B, A);     // feature (no activation)
        if constexpr (SAVE) save_frags<8, 256>(a.sv_feat, A, pidx, q);
        f32x4 alpha[2], rgb[2];
        tile_single<Lay::F_ALPHA, 144, 8, NB, NF>(c, B, alpha);            // row 0 = sigma
        layer16<Lay::F_VIEWS, 145, 4, 8, KD, true, NB, NF>(c, A, Dv, B);   // views_linears.0 (128 rows)
        if constexpr (SAVE) { save_frags<4, 128>(a.sv_hv, B, pidx, q); save_bits<4>(a.sv_bits + 8 * BS, B, pidx, q); }
        tile_single<Lay::F_RGB, 153, 4, NB, NF>(c, B, rgb);                // rows 0..2
        static_for<2>([&](auto cc_) {
            constexpr int cc = cc_;
            if (valid[cc] && q == 0) {
                f32x4 o = {rgb[cc][0], rgb[cc][1], rgb[cc][2], alpha[cc][0]};
                *reinterpret_cast<f32x4 *>(a.out + 4 * pidx[cc]) = o;
            }
        });
    } else {
        f32x4 o[2];
        tile_single<Lay::F_HEAD, 128, 8, NB, NF>(c, B, o);
        static_for<2>([&](auto cc_) {
            constexpr int cc = cc_;
            if (valid[cc]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * q + r;
                    if (row < a.out_ch) a.out[(int64_t)a.out_ch * pidx[cc] + row] = o[cc][r];
                }
            }
        });
    }
    STAMP(t2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the tile (or the workgroup)
    if (dynamic && tid == 0) *ticket_lds = gridDim.x + ticket;
    __syncthreads();                                           // every wave is done with the ring before it is refilled
    tile = dynamic ? (int64_t)*ticket_lds : tile + gridDim.x;
#ifdef NERF_AMD_STAMPS
    if (a.stamps && lane == 0) {
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        unsigned long long *o = a.stamps + ((size_t)blockIdx.x * C::WAVES + c.wave) * 4;
        o[0] += t1 - t0; o[1] += t2 - t1; o[2] += t3 - t2; o[3] += 1;
    }
#endif
    }
    if (dynamic && tid == 0) {                                 // the last workgroup out leaves the pair zero for its next launch
        if (atomicAdd(a.tile_ctr + 1, 1u) == gridDim.x - 1) { a.tile_ctr[0] = 0u; a.tile_ctr[1] = 0u; }
    }
}


// ---------------------------------------------------------------------------------------------------------
// The pipelined form of the kernel above (inference; the training forward keeps the simple one).  Same
// arithmetic, same fragment order, bit-identical outputs; what changes is what happens BETWEEN two tiles of
// a workgroup.  tools/tile_stamps.py measured the simple kernel's tile at 105 k cycles of which 4.9 k are the
// tile start (coordinate loads, positional encoding, the first weight block's DMA latency) and 0.4 k the
// end-of-tile drain, with the matrix pipe idle throughout.  Here
//   * the weight stream is CONTINUOUS (pipeline.h, OPT 32): the next tile's first two blocks are fetched
//     under the current tile's last two and no DMA is drained at a tile boundary;
//   * the next tile's coordinates are loaded and its encodings generated a piece at a time on the k-steps
//     of layers 6 and 7 (the xyz encoding is dead after the skip layer, so it is overwritten in place; the
//     view-direction encoding is double-buffered), i.e. in the VALU slots the MFMAs leave free.
// ---------------------------------------------------------------------------------------------------------
template <int LX, int LD, bool VD, class C>
__global__ __launch_bounds__(C::WAVES * 64, 2) void mlp_bf16_s16p_kernel(MlpArgs a) {
    static_assert((C::OPT & 32) != 0 && C::PHASE > 0, "the pipelined kernel needs the continuous ring");
    constexpr int WG_THREADS = C::WAVES * 64, WG_POINTS = C::WAVES * 32;
    using Lay = Layout16<LX, LD, VD>;
    constexpr int KE = Lay::KE, KD = Lay::KD, NF = Lay::F_END, NB = (NF + C::BF - 1) / C::BF;
    constexpr int SHIFT = NB % C::NS;                      // ring slots the block numbering advances per tile
    using EncX = Enc16<LX, KE>;
    using EncD = Enc16<(VD ? LD : 1), (VD ? KD : 1)>;
    // Pieces of the next tile's preparation, one per hook slot.  A 256-wide layer's pair is one ring block (16 fragments,
    // k-steps 0..7) with its block sync in front of k-step 4, and has two slots: 2p after k-step 0 (half a block behind
    // the previous sync) and 2p+1 after k-step 4 (right behind its own sync).  Global loads go right behind a sync: they
    // sit in the same in-order vmcnt queue as the ring DMA, so the NEXT sync's counted wait also waits for them (it is
    // only stricter than it needs to be), a whole block later; their first use comes after that sync.
    //   layers 6 + 7 (32 slots): the NEXT tile's two points one after the other -- loads, point, NSX encoding steps --
    //     so only one point's raw coordinates and encoder state are live at a time (registers are what limits this);
    //   feature layer (16 slots, view-branch models): THIS tile's view directions -- loads, then 2 x NSD steps -- which
    //     nothing needs before the views layer that follows.
    constexpr int NSX = EncX::NSTEP, NSD = VD ? EncD::NSTEP : 0;
    constexpr int PT_SLOTS = 4 + NSX + ((4 + NSX) & 1);           // load (odd slot), two slots later the point, then the steps; even length
    constexpr int SLOT_LOAD = 1, SLOT_PTS = 3, SLOT_ENC = 4;       // relative to a point's first slot
    static_assert(C::BF == 16 && C::PHASE == 8, "hook slots are laid out for a sync in front of k-step 4");
    static_assert(2 * PT_SLOTS <= 32 && 3 + 2 * NSD <= 16, "the encodings must fit on the hooks they are given");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *bias_lds = reinterpret_cast<float *>(smem + C::RING_BYTES);
    lds_u32_t *ticket_lds = (lds_u32_t *)(uintptr_t)(uint32_t)(uintptr_t)(bias_lds + Lay::N_TILES * 16);     // the tile after next (dynamic deal)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = lane >> 4;
    const int64_t n_point_tiles = (a.P + WG_POINTS - 1) / WG_POINTS;
    int64_t tile = blockIdx.x;
    if (tile >= n_point_tiles) return;                     // whole workgroups only (the launcher never over-provisions)
    // The tiles a workgroup walks.  Static deal (a.tile_ctr == NULL): blockIdx, blockIdx + gridDim, ...  Dynamic deal: the
    // first two like that, every later one a TICKET from the launch's counter -- the eight dies do not hold the same clock
    // at the power cap (tools/micro/field_wg_ends.py: identical cycles per tile on every XCD, workgroups of the fastest die
    // out 6-8 % of the launch before those of the slowest), and a tile's result does not depend on who computes it.  The
    // ticket for the tile AFTER next is taken by wave 0 while the next tile's coordinates are loaded (the index of the next
    // tile is needed that early), handed to the other waves through one LDS word, and read at the end of the iteration.
    int64_t tile_n = tile + gridDim.x;
    const bool dynamic = a.tile_ctr != nullptr;
    C c;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.lag = __builtin_amdgcn_readfirstlane(c.wave >= C::WAVES / 2 ? 1 : 0);
    // SPLIT_DMA issue phase (an SGPR: the DMA asm branches on it): two phases 0 / 2 by wave half; four phases
    // 0 1 2 3 2 3 0 1 over the waves, so that SIMD partners (w, w + 4) are always half a block apart
    c.phase = __builtin_amdgcn_readfirstlane(C::N_PHASES == 4 ? ((c.wave + 2 * (c.wave >> 2)) & 3) : (c.wave >= C::WAVES / 2 ? 2 : 0));
    c.gstream = reinterpret_cast<const char *>(a.stream_s16) + lane * 16;
    c.gstream_next = c.gstream;
    c.ring_lane = smem + lane * 16;
    c.ring_u32 = (uint32_t)(uintptr_t)smem;
#pragma unroll
    for (int i = 0; i < C::NS; ++i) {
        c.slot_u32[i] = c.ring_u32 + i * C::BLOCK_BYTES;
        c.slot_lane[i] = c.slot_u32[i] + lane * 16;
    }
    if constexpr ((C::OPT & 64) != 0) {
        c.rsrc = make_rsrc(a.stream_s16, (unsigned)(NB * C::BLOCK_BYTES));
        c.rsrc_next = c.rsrc;                  // one model per launch: the next tile streams the same weights
        c.lane16 = lane * 16;
        c.wave_off = c.wave * C::PIECES * 1024;
    }
    c.bias_half = bias_lds + q * 4;          // this lane's 4 rows of every 16-row tile
    for (int i = tid; i < Lay::N_TILES * 16; i += WG_THREADS) bias_lds[i] = a.bias_s16[i];

    // ---- coordinates of a tile: this lane's two points (column tile cc, column lane&15)
    float ld[7], dvn[2][3];                  // raw loads of ONE point of the next tile: o(3) d(3) z | explicit point(3); this tile's view directions
    float xsn[3];
    const int32_t P32 = (int32_t)a.P;        // P < 2^31 (checked at launch)
    const int32_t lane_pt = c.wave * 32 + (lane & 15);
    auto point_index = [&](int64_t t, int cc) { return (int32_t)t * WG_POINTS + lane_pt + cc * 16; };
    auto issue_loads = [&](int64_t t, int cc) {
        const int32_t p = point_index(t, cc);
        const int64_t pc = p < P32 ? p : P32 - 1;
        if (a.pts) {
            ld[0] = a.pts[3 * pc + 0]; ld[1] = a.pts[3 * pc + 1]; ld[2] = a.pts[3 * pc + 2];
        } else {
            const float *r = a.rays + (int64_t)((uint32_t)pc / (uint32_t)a.S) * a.ray_stride;
#pragma unroll
            for (int k = 0; k < 6; ++k) ld[k] = r[k];
            ld[6] = a.z_vals[pc];
        }
    };
    // the view directions of the CURRENT tile's points: loaded and encoded on the hooks of its feature layer (Dv is not
    // needed before the views layer), so nothing of it is live while the next tile is prepared
    auto issue_dir_loads = [&]() {
        if constexpr (VD) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int32_t p = point_index(tile, cc);
                const int64_t pc = p < P32 ? p : P32 - 1;
                const float *d = a.viewdirs + (int64_t)((uint32_t)pc / (uint32_t)a.S) * a.vd_stride;
                dvn[cc][0] = d[0]; dvn[cc][1] = d[1]; dvn[cc][2] = d[2];
            }
        }
    };
    auto make_point = [&]() {
        if (a.pts) {
            xsn[0] = ld[0]; xsn[1] = ld[1]; xsn[2] = ld[2];
        } else {                                             // pts = o + d z, rounded like the reference's two ops
            xsn[0] = mul_then_add(ld[3], ld[6], ld[0]);
            xsn[1] = mul_then_add(ld[4], ld[6], ld[1]);
            xsn[2] = mul_then_add(ld[5], ld[6], ld[2]);
        }
    };

    bf16x8 E[KE * 2];
    bf16x8 Dv[(VD ? KD : 1) * 2];
    static_for<2>([&](auto cc_) {            // the first tile's points: nothing to hide the loads and the encoding under
        constexpr int cc = cc_;
        issue_loads(tile, cc);
        make_point();
        encode16<LX, KE, 2>(xsn[0], xsn[1], xsn[2], q >> 1, q & 1, E + cc);
    });
    pipeline_prologue<NB>(c);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // bias table stores, before the first barrier publishes them

    // ---- the dynamic deal's two steps, both on hook slots of layer 6 (wave 0 acts, the branch is inside the asm: to the
    // compiler the hooks stay straight-line code, see issue_block).  take_ticket: one lane's atomic add with return, issued
    // right in front of the next tile's first coordinate load -- the vmcnt wait the compiler places for THAT load, two hook
    // slots later, covers the older atomic too (returns are in order), so the ticket costs no wait of its own.
    unsigned ticket = 0;
    const unsigned ticket_lds_addr = (unsigned)(uintptr_t)ticket_lds;
    const int tk_sel = __builtin_amdgcn_readfirstlane(dynamic ? c.wave : 1);      // 0: this wave takes and publishes the tickets
    auto take_ticket = [&]() {
        unsigned long long save;
        const unsigned one = 1u;
        asm volatile("s_cmp_lg_u32 %[w], 0\n\t"
                     "s_cbranch_scc1 .Lskip_tk_%=\n\t"
                     "s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, 1\n\t"
                     "global_atomic_add %[r], %[p], %[o], off sc0\n\t"
                     "s_mov_b64 exec, %[sv]\n"
                     ".Lskip_tk_%=:"
                     : [r] "+v"(ticket), [sv] "=&s"(save)
                     : [w] "s"(tk_sel), [p] "v"(a.tile_ctr), [o] "v"(one)
                     : "memory", "scc");
    };
    // publish_ticket: behind the point the loaded coordinates were consumed at (the asm's input ties it to them, so the
    // compiler's wait for the load precedes it): tile index = 2 gridDim + ticket into the LDS word; the other waves read it at
    // the end of the iteration, many block syncs later.
    auto publish_ticket = [&]() {
        asm volatile("" : "+v"(ticket) : "v"(ld[0]));
        unsigned ts, tv;
        asm volatile("s_cmp_lg_u32 %[w], 0\n\t"
                     "s_cbranch_scc1 .Lskip_tw_%=\n\t"
                     "v_readfirstlane_b32 %[t], %[tk]\n\t"
                     "s_add_u32 %[t], %[t], %[b]\n\t"
                     "v_mov_b32 %[tv], %[t]\n\t"
                     "ds_write_b32 %[ad], %[tv]\n\t"
                     "s_waitcnt lgkmcnt(0)\n"
                     ".Lskip_tw_%=:"
                     : [t] "=&s"(ts), [tv] "=&v"(tv)
                     : [w] "s"(tk_sel), [tk] "v"(ticket), [b] "s"(2u * gridDim.x), [ad] "v"(ticket_lds_addr)
                     : "memory", "scc");
    };

    EncX ex;
    EncD ed;
    auto prepare = [&](auto slot_) {         // slot 0..31 of layers 6 + 7
        constexpr int slot = slot_, cc = slot / PT_SLOTS, r = slot % PT_SLOTS;
        if constexpr (cc < 2) {
            if constexpr (r == SLOT_LOAD) {
                if constexpr (cc == 0) take_ticket();
                issue_loads(c.has_next ? tile_n : tile, cc);                // nothing follows: reload this tile (never used)
            } else if constexpr (r == SLOT_PTS) {
                make_point();
                if constexpr (cc == 0) publish_ticket();
            } else if constexpr (r >= SLOT_ENC && r < SLOT_ENC + NSX) {
                constexpr int st = r - SLOT_ENC;
                if constexpr (st == 0) ex.begin(xsn[0], xsn[1], xsn[2], q >> 1, q & 1);
                ex.template step<st, 2>(E + cc);                                               // E is dead after the skip layer
                if constexpr (st == NSX - 1) ex.template finish<2>(q >> 1, q & 1, E + cc);
            }
        }
    };
    auto hook_feat = [&](auto p_, auto k_) {
        constexpr int p = p_, k = k_;
        if constexpr (VD && (k == 0 || k == 4)) {
            constexpr int slot = 2 * p + (k == 4);
            if constexpr (slot == 1) issue_dir_loads();
            if constexpr (slot >= 3 && slot < 3 + 2 * NSD) {
                constexpr int cc = (slot - 3) / NSD, st = (slot - 3) % NSD;
                if constexpr (st == 0) ed.begin(dvn[cc][0], dvn[cc][1], dvn[cc][2], q >> 1, q & 1);
                ed.template step<st, 2>(Dv + cc);
                if constexpr (st == NSD - 1) ed.template finish<2>(q >> 1, q & 1, Dv + cc);
            }
        }
    };
    auto hook6 = [&](auto p_, auto k_) {
        constexpr int p = p_, k = k_;
        if constexpr (k == 0 || k == 4) prepare(std::integral_constant<int, 2 * p + (k == 4)>{});
    };
    auto hook7 = [&](auto p_, auto k_) {
        constexpr int p = p_, k = k_;
        if constexpr (k == 0 || k == 4) prepare(std::integral_constant<int, 16 + 2 * p + (k == 4)>{});
    };

#ifdef NERF_AMD_STAMPS
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (;;) {
#ifdef NERF_AMD_STAMPS
    if (a.stamps && lane == 0) {             // diagnostic build: cycles per loop iteration (= per tile) of this wave
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();
        unsigned long long *o = a.stamps + ((size_t)blockIdx.x * C::WAVES + c.wave) * 4;
        o[1] += t_now - t_prev; o[3] += 1;
        t_prev = t_now;
    }
#endif
    c.has_next = tile_n < n_point_tiles;
    // opaque per-iteration copies of the stream pointers: otherwise the DMA source addresses of the body are
    // loop-invariant, get hoisted in front of the loop and cost ~300 VGPRs
    asm volatile("" : "+v"(c.gstream));
    c.gstream_next = c.gstream;              // one model per launch: the next tile streams the same weights

    block_sync<-1, NB>(c);                                 // publishes block 0
    static_for<C::LA>([&](auto i_) { constexpr int i = i_; c.q[i] = ring_frag<i>(c); });

    bf16x8 A[16], B[16];
    layer16<Lay::F_L0, 0, 8, KE, 0, true, NB, NF>(c, E, E, A);
    layer16<Lay::F_L1 + 0 * 128, 16, 8, 8, 0, true, NB, NF>(c, A, A, B);
    layer16<Lay::F_L1 + 1 * 128, 32, 8, 8, 0, true, NB, NF>(c, B, B, A);
    layer16<Lay::F_L1 + 2 * 128, 48, 8, 8, 0, true, NB, NF>(c, A, A, B);
    layer16<Lay::F_L1 + 3 * 128, 64, 8, 8, 0, true, NB, NF>(c, B, B, A);
    layer16<Lay::F_L5, 80, 8, KE, 8, true, NB, NF>(c, E, A, B);            // skip: [input_pts | h]
    layer16<Lay::F_L6, 96, 8, 8, 0, true, NB, NF>(c, B, B, A, hook6);      // + next tile: loads, points, encoding ...
    layer16<Lay::F_L6 + 128, 112, 8, 8, 0, true, NB, NF>(c, A, A, B, hook7);   // ... h7 in B

    if constexpr (VD) {
        layer16<Lay::F_FEAT, 128, 8, 8, 0, false, NB, NF>(c, B, B, A, hook_feat);   // feature (no activation) + this tile's view directions
        f32x4 alpha[2], rgb[2];
        tile_single<Lay::F_ALPHA, 144, 8, NB, NF>(c, B, alpha);            // row 0 = sigma
        layer16<Lay::F_VIEWS, 145, 4, 8, KD, true, NB, NF>(c, A, Dv, B);   // views_linears.0 (128 rows)
        tile_single<Lay::F_RGB, 153, 4, NB, NF>(c, B, rgb);                // rows 0..2
        static_for<2>([&](auto cc_) {
            constexpr int cc = cc_;
            const int32_t p = point_index(tile, cc);
            if (p < P32 && q == 0) {
                f32x4 o = {rgb[cc][0], rgb[cc][1], rgb[cc][2], alpha[cc][0]};
                *reinterpret_cast<f32x4 *>(a.out + 4 * (int64_t)p) = o;
            }
        });
    } else {
        f32x4 o[2];
        tile_single<Lay::F_HEAD, 128, 8, NB, NF>(c, B, o);
        static_for<2>([&](auto cc_) {
            constexpr int cc = cc_;
            const int32_t p = point_index(tile, cc);
            if (p < P32) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * q + r;
                    if (row < a.out_ch) a.out[(int64_t)a.out_ch * p + row] = o[cc][r];
                }
            }
        });
    }
    if (!c.has_next) break;
    tile = tile_n;
    tile_n = dynamic ? (int64_t)*ticket_lds : tile + gridDim.x;
    if constexpr (SHIFT != 0) {              // the next tile's block b lives in the slot this tile's block b + NB had
        uint32_t sl[C::NS], su[C::NS];
#pragma unroll
        for (int i = 0; i < C::NS; ++i) { sl[i] = c.slot_lane[(i + SHIFT) % C::NS]; su[i] = c.slot_u32[(i + SHIFT) % C::NS]; }
#pragma unroll
        for (int i = 0; i < C::NS; ++i) { c.slot_lane[i] = sl[i]; c.slot_u32[i] = su[i]; }
    }
    }
#ifdef NERF_AMD_STAMPS
    if (a.stamps && lane == 0) {             // diagnostic build: when this wave left the kernel (100 MHz clock) and on which XCD
        unsigned long long *o = a.stamps + ((size_t)blockIdx.x * C::WAVES + c.wave) * 4;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = wall_clock64();
        o[2] = xcc & 15;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the workgroup
    if (dynamic && tid == 0) {                                 // the last workgroup out leaves the pair zero for its next launch
        if (atomicAdd(a.tile_ctr + 1, 1u) == gridDim.x - 1) { a.tile_ctr[0] = 0u; a.tile_ctr[1] = 0u; }
    }
}

// ---- {ticket, done} pairs of the dynamic deal
namespace {
constexpr int TILE_CTR_SLOTS = 1024;
unsigned *g_tile_ctr[16] = {};
std::atomic<unsigned> g_tile_ctr_next{0};
}  // namespace

int tile_counters_init(int device) {
    if (device < 0 || device >= 16) return NERF_AMD_EINVAL;
    if (g_tile_ctr[device]) return NERF_AMD_OK;
    unsigned *p = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p), TILE_CTR_SLOTS * 2 * sizeof(unsigned)) != hipSuccess) return NERF_AMD_EHIP;
    if (hipMemset(p, 0, TILE_CTR_SLOTS * 2 * sizeof(unsigned)) != hipSuccess) { (void)hipFree(p); return NERF_AMD_EHIP; }
    g_tile_ctr[device] = p;
    return NERF_AMD_OK;
}

unsigned *tile_counter_slot(int device) {
    if (device < 0 || device >= 16 || !g_tile_ctr[device]) return nullptr;
    return g_tile_ctr[device] + 2 * (g_tile_ctr_next.fetch_add(1, std::memory_order_relaxed) % TILE_CTR_SLOTS);
}

unsigned *tile_counter_for(bool deal, hipStream_t s) {
    if (!deal) return nullptr;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? tile_counter_slot(dev) : nullptr;
}

template <int LX, int LD, bool VD, class C>
static int launch_wg16p(const MlpArgs &a, int n_frags_used, int n_tiles, hipStream_t s) {
    constexpr int WG_THREADS = C::WAVES * 64, WG_POINTS = C::WAVES * 32;
    using Lay = Layout16<LX, LD, VD>;
    if (n_frags_used != Lay::F_END || n_tiles != Lay::N_TILES) return NERF_AMD_EINVAL;
    const size_t lds = C::RING_BYTES + (size_t)Lay::N_TILES * 16 * sizeof(float) + 16;      // + the ticket word
    static DynamicLdsOptIn opt_in;
    if (opt_in.ensure(reinterpret_cast<const void *>(mlp_bf16_s16p_kernel<LX, LD, VD, C>), lds) != hipSuccess) return NERF_AMD_EHIP;
    int64_t groups = (a.P + WG_POINTS - 1) / WG_POINTS;
    if (groups <= 0) return NERF_AMD_OK;
    if (a.P >= (int64_t)1 << 31) return NERF_AMD_EINVAL;
    const int n_cu = device_cu_count();      // one workgroup per CU walks the tiles
    const bool deal = groups > 2 * (int64_t)n_cu && g_variant != 42;      // more than two tiles per workgroup: dealt by ticket (A/B 42: static)
    if (groups > n_cu) groups = n_cu;
    MlpArgs a2 = a;
    a2.tile_ctr = tile_counter_for(deal, s);
#ifdef NERF_AMD_STAMPS
    a2.stamps = g_stamp_buf;
#endif
    hipLaunchKernelGGL((mlp_bf16_s16p_kernel<LX, LD, VD, C>), dim3((unsigned)groups), dim3(WG_THREADS), lds, s, a2);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

template <int LX, int LD, bool VD, class C, bool SAVE = false>
static int launch_wg16(const MlpArgs &a, int n_frags_used, int n_tiles, hipStream_t s) {
    constexpr int WG_THREADS = C::WAVES * 64, WG_POINTS = C::WAVES * 32;
    using Lay = Layout16<LX, LD, VD>;
    if (n_frags_used != Lay::F_END || n_tiles != Lay::N_TILES) return NERF_AMD_EINVAL;
    const size_t lds = C::RING_BYTES + (size_t)Lay::N_TILES * 16 * sizeof(float) + 16;      // + the ticket word
    static DynamicLdsOptIn opt_in;         // per kernel instantiation, tracks every device (launch_util.h)
    if (opt_in.ensure(reinterpret_cast<const void *>(mlp_bf16_s16_kernel<LX, LD, VD, C, SAVE>), lds) != hipSuccess)
        return NERF_AMD_EHIP;
    int64_t groups = (a.P + WG_POINTS - 1) / WG_POINTS;
    if (groups <= 0) return NERF_AMD_OK;
    if (a.P >= (int64_t)1 << 31) return NERF_AMD_EINVAL;
    bool deal = false;
    if (g_variant != 31) {                 // one workgroup per CU walks the tiles (+1 %: no per-tile dispatch); 31 = A/B off
        const int n_wg = device_cu_count() * (8 / C::WAVES);     // 4-wave workgroups: two per CU
        deal = groups > 2 * (int64_t)n_wg && g_variant != 42;    // ... and takes them by ticket (A/B 42: blockIdx + k gridDim)
        if (groups > n_wg) groups = n_wg;
    }
    MlpArgs a2 = a;
    a2.tile_ctr = tile_counter_for(deal, s);
#ifdef NERF_AMD_STAMPS
    a2.stamps = g_stamp_buf;
#endif
    hipLaunchKernelGGL((mlp_bf16_s16_kernel<LX, LD, VD, C, SAVE>), dim3((unsigned)groups), dim3(WG_THREADS), lds, s, a2);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

// 64-KiB ring of 16-fragment blocks, mid-block sync, 4-deep read-ahead pinned in front of the MFMAs it runs ahead of
// (OPT 4), DMA issue of the two SIMD partners half a block apart (OPT 8).  tools/mlp_ab.py, 4096 x 192 points:
// 0.606 ms against 0.630 for the round-1 shape Ctx<8, 16, 4, 8, 2> on the same device.
// + ring DMA as buffer_load ... lds (OPT 64: one asm statement per wave and block, no per-piece address arithmetic):
// another -2.3 ... -3.6 % (fine) / -2.1 ... -4.5 % (coarse) depending on the device.
using Cfg16 = Ctx<8, 16, 4, 8, 4, 0, 1, 4 + 8 + 64>;
using Cfg16R1 = Ctx<8, 16, 4, 8, 2>;          // round-1 shape (A/B: nerf_amd_set_tuning(0, 40))
using Cfg16P = Ctx<8, 16, 4, 8, 4, 0, 1, 4 + 8 + 32 + 64>;   // the pipelined kernel: Cfg16 + continuous ring
#ifndef NA_EXPERIMENT_SAVE_CFG
template <int LX, int LD> using CfgSaveT = Ctx<8, 16, 4, 8, 2>;   // the training forward (Cfg16's read-ahead spills beside the saved rows)
#endif

#ifdef NERF_AMD_EXPERIMENTS
#include "../../tools/experiments/field_variants_s16.inc"
#endif

int launch_mlp_bf16_s16(const MlpArgs &a, int multires, int multires_views, int use_viewdirs,
                        int n_frags_used, int n_tiles, hipStream_t s) {
#ifdef NERF_AMD_EXPERIMENTS      // A/B builds only: the variant table lives in tools/experiments/field_variants_s16.inc
    {
        int rc_x = NERF_AMD_EUNSUPPORTED;
        if (experiment_launch_s16(a, multires, multires_views, use_viewdirs, n_frags_used, n_tiles, s, &rc_x)) return rc_x;
    }
#endif
    if (use_viewdirs && multires == 10 && multires_views == 4 && g_variant == 40) return launch_wg16<10, 4, true, Cfg16R1>(a, n_frags_used, n_tiles, s);
    if (g_variant != 41) {                  // 41 = A/B: the simple per-tile kernel
        if (use_viewdirs) {
            if (multires == 10 && multires_views == 4) return launch_wg16p<10, 4, true, Cfg16P>(a, n_frags_used, n_tiles, s);
            if (multires == 15 && multires_views == 6) return launch_wg16p<15, 6, true, Cfg16P>(a, n_frags_used, n_tiles, s);
        } else if (a.out_ch <= 16) {
            if (multires == 10) return launch_wg16p<10, 0, false, Cfg16P>(a, n_frags_used, n_tiles, s);
            if (multires == 15) return launch_wg16p<15, 0, false, Cfg16P>(a, n_frags_used, n_tiles, s);
        }
        return NERF_AMD_EUNSUPPORTED;
    }
    if (use_viewdirs) {
        if (multires == 10 && multires_views == 4) return launch_wg16<10, 4, true, Cfg16>(a, n_frags_used, n_tiles, s);
        if (multires == 15 && multires_views == 6) return launch_wg16<15, 6, true, Cfg16>(a, n_frags_used, n_tiles, s);
    } else if (a.out_ch <= 16) {
        if (multires == 10) return launch_wg16<10, 0, false, Cfg16>(a, n_frags_used, n_tiles, s);
        if (multires == 15) return launch_wg16<15, 0, false, Cfg16>(a, n_frags_used, n_tiles, s);
    }
    return NERF_AMD_EUNSUPPORTED;
}

#ifdef NERF_AMD_STAMPS
extern "C" void nerf_amd_debug_set_stamp_buffer(void *p) { g_stamp_buf = static_cast<unsigned long long *>(p); }
#endif

int launch_mlp_bf16_s16_save(const MlpArgs &a, int multires, int multires_views, int use_viewdirs, int n_frags_used, int n_tiles,
                             hipStream_t s) {
    // the training forward keeps the round-1 pipeline shape: with its activation stores the pinned / split shape spills
    if (use_viewdirs) {
        if (multires == 10 && multires_views == 4) return launch_wg16<10, 4, true, CfgSaveT<10, 4>, true>(a, n_frags_used, n_tiles, s);
        if (multires == 15 && multires_views == 6) return launch_wg16<15, 6, true, CfgSaveT<15, 6>, true>(a, n_frags_used, n_tiles, s);
    } else if (a.out_ch <= 16) {             // output_linear models (nerf.py:91-94): hidden layers saved the same way, no view branch
        using CfgSaveNV = Ctx<8, 16, 4, 8, 2>;
        if (multires == 10) return launch_wg16<10, 0, false, CfgSaveNV, true>(a, n_frags_used, n_tiles, s);
        if (multires == 15) return launch_wg16<15, 0, false, CfgSaveNV, true>(a, n_frags_used, n_tiles, s);
    }
    return NERF_AMD_EUNSUPPORTED;
}

}  // namespace na
