// split.h -- the arithmetic of NERF_AMD_PREC_FP32_SPLIT shared by the forward (mlp_split.hip), the dX chain
// (mlp_bwd_split.hip) and the weight-gradient products (backward.hip): a value is an unevaluated pair of fp16 numbers
//     x = x_hi + 2^-11 x_lo,   x_hi = fp16(x),   x_lo = fp16((x - x_hi) 2^11)          (22 significant bits)
// and a product of two such operands is three v_mfma_f32_16x16x32_f16 with fp32 accumulation (the lo x lo term,
// 2^-22 relative, is dropped).  Saved activations and gradients are stored exactly as the kernels hold them:
// a plane of hi rows and a plane of lo rows, both in the slot order of the bf16 training arrays (kernels.h).
#pragma once
#include <hip/hip_runtime.h>

#include "pipeline.h"

namespace na {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

#define MFMAH(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a_), __builtin_bit_cast(f16x8, b_), c_, 0, 0, 0)

constexpr float SPLIT_SCALE = 2048.0f, SPLIT_INV = 1.0f / 2048.0f;

// x -> (hi, lo): hi = fp16(x) rounded to nearest, lo = fp16 of the (exact) residual scaled by 2^11.  A denormal hi is
// fine: v_mfma_f32_16x16x32_f16 multiplies fp16 denormals exactly (tools/micro/mfma_f16_denorm.hip), and the residual
// carries whatever hi could not.
__device__ __forceinline__ void split_f16(float x, _Float16 &hi, _Float16 &lo) {
    const _Float16 h = (_Float16)x;
    hi = h;
    lo = (_Float16)((x - (float)h) * SPLIT_SCALE);
}

// Two values at once, on packed registers: hi = (fp16(x0), fp16(x1)) is one v_cvt_pk_f16_f32; each lo is ONE mixed-precision
// fma that reads its hi straight out of the packed register -- fp16(-2^11 hi + 2^11 x), both products exact, one rounding: the
// same value as split_f16's, without converting hi back to fp32 (the compiler does not form v_fma_mix from the C expression).
__device__ __forceinline__ void split_f16_pair(float x0, float x1, f16x2 &hi, f16x2 &lo) {
    f16x2 h;
    h[0] = (_Float16)x0; h[1] = (_Float16)x1;
    typedef __attribute__((ext_vector_type(2))) float f32x2s;
    const f32x2s sc = f32x2s{x0, x1} * SPLIT_SCALE;          // one v_pk_mul_f32
    const float kneg = -SPLIT_SCALE;
    unsigned l;
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hb), "v"(kneg), "v"(sc[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hb), "v"(kneg), "v"(sc[1]));
    hi = h;
    lo = __builtin_bit_cast(f16x2, l);
}

template <class C, int NR, int NM>
__device__ __forceinline__ void sched_step_split() {
    if constexpr ((C::OPT & 4) != 0) {
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);   // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);   // MFMA
    }
}

// ---- loss scale of the split-precision backward pass
// dL/draw reaches the dX chain as fp16 pairs, so it is multiplied by a power of two S first: S puts the largest
// |dL/draw| of the launch into [2^GRAD_SCALE_LOG2, 2^(GRAD_SCALE_LOG2 + 1)) -- 2^11 of headroom below fp16's 65504 for the
// growth of the gradient through the layers, and the bulk of the values inside fp16's normal range.  Every product of the
// chain is linear in the gradient, so the scale comes off exactly at the end (weight gradients: the slab reduction; ray
// gradients: before the atomics).  gmax_kernel leaves GRAD_SCALE_PARTS partial maxima; every consumer derives the same
// S from them (grad_scale_bits), no atomics and nothing to reset.
constexpr int GRAD_SCALE_LOG2 = 4;
constexpr int GRAD_SCALE_PARTS = 256;

// S as fp32 bits from the maximum of |dL/draw| (0, inf and NaN -> 1.0)
__host__ __device__ inline unsigned grad_scale_bits(float gmax) {
    unsigned u;
    __builtin_memcpy(&u, &gmax, 4);
    const int e = (int)((u >> 23) & 0xffu);
    if (e == 0 || e == 255) return 0x3f800000u;
    int se = 127 + GRAD_SCALE_LOG2 - (e - 127);
    se = se < 1 ? 1 : (se > 253 ? 253 : se);                  // 1 / S must be a normal number too
    return (unsigned)se << 23;
}
__host__ __device__ inline unsigned grad_scale_inv_bits(unsigned scale_bits) { return (254u - (scale_bits >> 23)) << 23; }

// One wave reads the partial maxima and returns S (all lanes).
__device__ __forceinline__ float grad_scale_of(const float *parts) {
    const int lane = threadIdx.x & 63;
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < GRAD_SCALE_PARTS / 64; ++i) m = fmaxf(m, parts[lane + 64 * i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    return __builtin_bit_cast(float, grad_scale_bits(m));
}

}  // namespace na
