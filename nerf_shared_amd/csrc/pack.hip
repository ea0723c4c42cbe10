// pack.hip -- re-pack live nn.Linear parameters ([out,in] fp32, state_dict
// layout of /root/reference/nerf_shared/nerf.py:62-94) into the fragment
// streams of program.h, on the device (no host sync), plus the host twin used
// by the CPU layout tests.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "program.h"

namespace na {

NA_HD inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// fp32 -> IEEE fp16 bits, round to nearest even (overflow -> inf), by hand: shared by the host twin, whose toolchain
// may lack the _Float16 conversion routines.
NA_HD inline uint16_t f32_to_f16_rne(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                 // NaN
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                // rounds to >= 65520: inf
    if (u < 0x38800000u) {                                                  // |f| < 2^-14: fp16 denormal or zero
        if (u < 0x33000000u) return (uint16_t)sign;                         // < 2^-25: zero
        const int e = (int)(u >> 23);                                       // biased fp32 exponent, 102..112
        uint32_t m = (u & 0x7fffffu) | 0x800000u;                           // 24-bit significand
        const int shift = 126 - e;                                          // 14..24: significand >> shift = multiples of 2^-24
        const uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1u);
        m >>= shift;
        if (rest > half || (rest == half && (m & 1u))) ++m;
        return (uint16_t)(sign | m);
    }
    u += 0xc8000000u;                                                       // re-bias the exponent: -(127 - 15) << 23
    u += 0x0fffu + ((u >> 13) & 1u);
    return (uint16_t)(sign | (u >> 13));
}
NA_HD inline float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {                                                              // denormal: normalise
            int k = 0;
            uint32_t mm = m;
            while (!(mm & 0x400u)) { mm <<= 1; ++k; }
            u = sign | ((uint32_t)(113 - k) << 23) | ((mm & 0x3ffu) << 13);
        }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
// Element of a fragment of `part` (program.h FragDesc): bf16(w), or the fp16 pair of the split-precision stream --
// hi = fp16(w) (0 where that would be a denormal, so no operand ever is one), lo = fp16((w - hi) 2^11).
NA_HD inline uint16_t pack_value(float w, int part) {
    if (part == 0) return f32_to_bf16_rne(w);
    uint16_t hi = f32_to_f16_rne(w);
    if ((hi & 0x7c00u) == 0) hi = 0;
    if (part == 1) return hi;
    return f32_to_f16_rne((w - f16_bits_to_f32(hi)) * 2048.0f);
}

// Weight column (inside the tensor) held by element j of lane `lane` of fragment d, or -1.
NA_HD inline int frag_source(const FragDesc &d, int lane, int j, int n_out, int *row) {
    if (d.kind == FRAG_T16 || d.kind == FRAG_TG16 || d.kind == FRAG_TE16) {   // transposed: *row = output feature, column = input feature
        const int i = lane & 15, q = lane >> 4;
        const int o = d.kind == FRAG_TG16 ? 32 * d.ks + 8 * q + j : acc16_col(d.ks, q, j);
        *row = o;
        if (o >= d.seg_len) return -1;
        if (d.kind == FRAG_TE16) {       // tile row i = encoding slot (ks' = t>>1, q' = i>>2, j' = 4(t&1) + (i&3))
            const int t = d.row0 >> 4;
            const int c = gen16_col(t >> 1, i >> 2, 4 * (t & 1) + (i & 3), d.L);
            return c < 0 ? -1 : d.col_base + c;
        }
        if (d.row0 + i >= d.L) return -1;
        return d.col_base + d.row0 + i;
    }
    const bool s16 = d.kind == FRAG_ACC16 || d.kind == FRAG_GEN16;
    const int o = s16 ? (lane & 15) : (lane & 31), h = s16 ? (lane >> 4) : (lane >> 5);
    *row = d.row0 + o;
    if (d.kind == FRAG_ZERO || *row >= n_out) return -1;
    const int c = d.kind == FRAG_GEN ? gen_col(d.ks, h, j, d.L)
                : d.kind == FRAG_ACC ? acc_col(d.ks, h, j)
                : d.kind == FRAG_GEN16 ? gen16_col(d.ks, h, j, d.L) : acc16_col(d.ks, h, j);
    if (c < 0 || c >= d.seg_len) return -1;
    return d.col_base + c;
}

// Every packed copy of a model from ONE launch (a re-pack follows every optimizer step of a training loop, where six
// small launches cost more on the host than on the GPU): block ranges select the job.
struct PackJobs {
    const FragDesc *frags_bwd, *frags16, *frags, *frags_split, *frags_bwd_split;
    const TileDesc *tiles16, *tiles;
    const LayerF32 *layers;
    const TrainLayerF32 *tlayers;
    const TensorDesc *tensors;
    uint16_t *stream_bwd, *stream_s16, *stream_bf16, *stream_split, *stream_bwd_split;
    float *bias_s16, *bias_bf16, *stream_f32, *bias_f32, *stream_f32_t;
    int n_bwd, n16, n32, n_split, n_bwd_split, n_tiles16, n_tiles, n_layers, n_layers_t;
    int b_bias16, b_bias32;          // blocks of the two bias tables
};
constexpr int PACK_F32_BLOCKS = 32;  // per layer

__device__ __forceinline__ void pack_frag(const FragDesc *frags, int n, const TensorDesc *tensors, const float *const *weights,
                                          uint16_t *stream) {
    const FragDesc d = frags[n];
    const int lane = threadIdx.x >> 3, j = threadIdx.x & 7;
    int row;
    const TensorDesc t = tensors[d.tensor];
    const int col = frag_source(d, lane, j, t.n_out, &row);
    const float v = col < 0 ? 0.0f : weights[d.tensor][(int64_t)row * t.n_in + col];
    stream[(int64_t)n * 512 + threadIdx.x] = pack_value(v, d.part);
}

__global__ __launch_bounds__(512) void pack_all_kernel(PackJobs J, PtrTable weights_tab, PtrTable biases_tab) {
    const float *const *weights = weights_tab.p;
    const float *const *biases = biases_tab.p;
    int b = blockIdx.x;
    if (b < J.n_bwd) { pack_frag(J.frags_bwd, b, J.tensors, weights, J.stream_bwd); return; }
    b -= J.n_bwd;
    if (b < J.n16) { pack_frag(J.frags16, b, J.tensors, weights, J.stream_s16); return; }
    b -= J.n16;
    if (b < J.n32) { pack_frag(J.frags, b, J.tensors, weights, J.stream_bf16); return; }
    b -= J.n32;
    if (b < J.n_split) { pack_frag(J.frags_split, b, J.tensors, weights, J.stream_split); return; }
    b -= J.n_split;
    if (b < J.n_bwd_split) { pack_frag(J.frags_bwd_split, b, J.tensors, weights, J.stream_bwd_split); return; }
    b -= J.n_bwd_split;
    if (b < J.b_bias16) {            // [tile][16 rows], natural row order
        const int e = b * 512 + threadIdx.x;
        if (e < J.n_tiles16 * 16) {
            const TileDesc t = J.tiles16[e >> 4];
            const int row = t.row0 + (e & 15);
            J.bias_s16[e] = row < J.tensors[t.tensor].n_out ? biases[t.tensor][row] : 0.0f;
        }
        return;
    }
    b -= J.b_bias16;
    if (b < J.b_bias32) {            // [tile][h][r] of the 32x32x16 kernel
        const int e = b * 512 + threadIdx.x;
        if (e < J.n_tiles * 32) {
            const TileDesc t = J.tiles[e >> 5];
            const int row = t.row0 + acc_row(e & 15, (e >> 4) & 1);
            J.bias_bf16[e] = row < J.tensors[t.tensor].n_out ? biases[t.tensor][row] : 0.0f;
        }
        return;
    }
    b -= J.b_bias32;
    // fp32 stream: blocks of 256 values = one (tile, group) of a layer; a 512-thread block packs two at a time
    if (b >= J.n_layers * PACK_F32_BLOCKS) {
        // the transposed fp32 stream of train_f32.hip: tiles over the layer's INPUT features, k over its outputs
        b -= J.n_layers * PACK_F32_BLOCKS;
        const int layer = b / PACK_F32_BLOCKS, bx = b - layer * PACK_F32_BLOCKS;
        if (layer >= J.n_layers_t) return;
        const LayerF32 L = J.layers[layer];
        const int tiles = (L.n_in + 31) >> 5, groups = (L.n_out + 7) >> 3;
        const int half = threadIdx.x >> 8, tid = threadIdx.x & 255;
        for (int blk = 2 * bx + half; blk < tiles * groups; blk += 2 * PACK_F32_BLOCKS) {
            const int t = blk / groups, g = blk - t * groups;
            const int lane = tid >> 2, i = tid & 3;
            const int row = 32 * t + (lane & 31), col = 8 * g + 2 * i + (lane >> 5);      // row: input feature, col: output feature
            const float v = (row < L.n_in && col < L.n_out) ? weights[L.tensor][(int64_t)col * L.n_in + row] : 0.0f;
            J.stream_f32_t[J.tlayers[layer].frag_off_t + (int64_t)blk * 256 + tid] = v;
        }
        return;
    }
    const int layer = b / PACK_F32_BLOCKS, bx = b - layer * PACK_F32_BLOCKS;
    if (layer >= J.n_layers) return;
    const LayerF32 L = J.layers[layer];
    const int tiles = (L.n_out + 31) >> 5, groups = (L.n_in + 7) >> 3;
    const int half = threadIdx.x >> 8, tid = threadIdx.x & 255;
    for (int blk = 2 * bx + half; blk < tiles * groups; blk += 2 * PACK_F32_BLOCKS) {
        const int t = blk / groups, g = blk - t * groups;
        const int lane = tid >> 2, i = tid & 3;
        const int row = 32 * t + (lane & 31), col = 8 * g + 2 * i + (lane >> 5);
        const float v = (row < L.n_out && col < L.n_in) ? weights[L.tensor][(int64_t)row * L.n_in + col] : 0.0f;
        J.stream_f32[L.frag_off + (int64_t)blk * 256 + tid] = v;
    }
    for (int i = bx * 512 + threadIdx.x; i < tiles * 32; i += PACK_F32_BLOCKS * 512)
        J.bias_f32[L.bias_off + i] = i < L.n_out ? biases[L.tensor][i] : 0.0f;
}

int launch_pack(const Program &p, const FragDesc *d_frags, const TileDesc *d_tiles, const LayerF32 *d_layers,
                const TensorDesc *d_tensors, const PtrTable &d_w, const PtrTable &d_b,
                uint16_t *stream_bf16, float *bias_bf16, float *stream_f32, float *bias_f32,
                const FragDesc *d_frags16, const TileDesc *d_tiles16, uint16_t *stream_s16, float *bias_s16,
                const FragDesc *d_frags_bwd, uint16_t *stream_bwd, const FragDesc *d_frags_split, uint16_t *stream_split,
                const FragDesc *d_frags_bwd_split, uint16_t *stream_bwd_split, const TrainLayerF32 *d_tlayers, float *stream_f32_t,
                int copies, hipStream_t s) {
    PackJobs J;
    J.frags_bwd_split = d_frags_bwd_split; J.stream_bwd_split = stream_bwd_split;
    J.frags_bwd = d_frags_bwd; J.frags16 = d_frags16; J.frags = d_frags; J.frags_split = d_frags_split;
    J.stream_split = stream_split;
    J.tiles16 = d_tiles16; J.tiles = d_tiles; J.layers = d_layers; J.tensors = d_tensors;
    J.stream_bwd = stream_bwd; J.stream_s16 = stream_s16; J.stream_bf16 = stream_bf16;
    J.bias_s16 = bias_s16; J.bias_bf16 = bias_bf16; J.stream_f32 = stream_f32; J.bias_f32 = bias_f32;
    // copies: NERF_AMD_COPY_* (include/nerf_amd.h); a job that is not asked for gets no blocks
    const bool c_bf16 = copies & NERF_AMD_COPY_BF16, c_split = copies & NERF_AMD_COPY_SPLIT;
    J.n_bwd = p.bf16_ok && (copies & NERF_AMD_COPY_BWD) ? (int)p.frags_bwd.size() : 0;
    J.n16 = p.bf16_ok && c_bf16 ? (int)p.frags16.size() : 0;
    J.n32 = p.bf16_ok && c_bf16 ? (int)p.frags.size() : 0;
    J.n_split = p.bf16_ok && c_split ? (int)p.frags_split.size() : 0;
    J.n_bwd_split = p.bf16_ok && (copies & NERF_AMD_COPY_BWD_SPLIT) ? (int)p.frags_bwd_split.size() : 0;
    J.n_tiles16 = p.bf16_ok && (c_bf16 || c_split) ? (int)p.tiles16.size() : 0;       // the 16-row bias table serves both
    J.n_tiles = p.bf16_ok && c_bf16 ? (int)p.tiles.size() : 0;
    J.n_layers = (copies & NERF_AMD_COPY_FP32) ? (int)p.layers.size() : 0;
    J.n_layers_t = (copies & NERF_AMD_COPY_FP32_BWD) && stream_f32_t ? (int)p.layers.size() : 0;
    J.tlayers = d_tlayers; J.stream_f32_t = stream_f32_t;
    J.b_bias16 = (J.n_tiles16 * 16 + 511) / 512;
    J.b_bias32 = (J.n_tiles * 32 + 511) / 512;
    const unsigned grid = (unsigned)(J.n_bwd + J.n16 + J.n32 + J.n_split + J.n_bwd_split + J.b_bias16 + J.b_bias32 + (J.n_layers + J.n_layers_t) * PACK_F32_BLOCKS);
    if (grid == 0) return NERF_AMD_OK;
    hipLaunchKernelGGL(pack_all_kernel, dim3(grid), dim3(512), 0, s, J, d_w, d_b);
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

void pack_bf16_host(const Program &p, int shape, const float *const *w, const float *const *b,
                    uint16_t *stream, float *bias) {
    if (shape == 17) {       // backward (transposed) stream; no bias table
        if (stream)
            for (size_t n = 0; n < p.frags_bwd.size(); ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        int row;
                        const TensorDesc &t = p.tensors[p.frags_bwd[n].tensor];
                        const int col = frag_source(p.frags_bwd[n], lane, j, t.n_out, &row);
                        stream[n * 512 + lane * 8 + j] =
                            f32_to_bf16_rne(col < 0 ? 0.0f : w[p.frags_bwd[n].tensor][(int64_t)row * t.n_in + col]);
                    }
        return;
    }
    if (shape == 18 || shape == 19) {   // split-precision streams (fp16 hi / lo fragments): 18 forward (bias table = shape 16's), 19 transposed
        const std::vector<FragDesc> &fr = shape == 18 ? p.frags_split : p.frags_bwd_split;
        if (stream)
            for (size_t n = 0; n < fr.size(); ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        int row;
                        const FragDesc &d = fr[n];
                        const TensorDesc &t = p.tensors[d.tensor];
                        const int col = frag_source(d, lane, j, t.n_out, &row);
                        stream[n * 512 + lane * 8 + j] = pack_value(col < 0 ? 0.0f : w[d.tensor][(int64_t)row * t.n_in + col], d.part);
                    }
        return;
    }
    if (shape == 16) {
        if (stream)
            for (size_t n = 0; n < p.frags16.size(); ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        int row;
                        const TensorDesc &t = p.tensors[p.frags16[n].tensor];
                        const int col = frag_source(p.frags16[n], lane, j, t.n_out, &row);
                        stream[n * 512 + lane * 8 + j] =
                            f32_to_bf16_rne(col < 0 ? 0.0f : w[p.frags16[n].tensor][(int64_t)row * t.n_in + col]);
                    }
        if (bias)
            for (size_t ti = 0; ti < p.tiles16.size(); ++ti)
                for (int r = 0; r < 16; ++r) {
                    const int row = p.tiles16[ti].row0 + r;
                    bias[ti * 16 + r] = row < p.tensors[p.tiles16[ti].tensor].n_out ? b[p.tiles16[ti].tensor][row] : 0.0f;
                }
        return;
    }
    if (stream)
        for (size_t n = 0; n < p.frags.size(); ++n)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    int row;
                    const TensorDesc &t = p.tensors[p.frags[n].tensor];
                    const int col = frag_source(p.frags[n], lane, j, t.n_out, &row);
                    stream[n * 512 + lane * 8 + j] =
                        f32_to_bf16_rne(col < 0 ? 0.0f : w[p.frags[n].tensor][(int64_t)row * t.n_in + col]);
                }
    if (bias)
        for (size_t ti = 0; ti < p.tiles.size(); ++ti)
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) {
                    const int row = p.tiles[ti].row0 + acc_row(r, h);
                    bias[ti * 32 + h * 16 + r] = row < p.tensors[p.tiles[ti].tensor].n_out ? b[p.tiles[ti].tensor][row] : 0.0f;
                }
}

}  // namespace na
