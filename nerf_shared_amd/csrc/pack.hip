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

__global__ __launch_bounds__(512) void pack_bf16_kernel(const FragDesc *frags, const TensorDesc *tensors,
                                                        PtrTable weights_tab, uint16_t *stream) {
    const float *const *weights = weights_tab.p;
    const FragDesc d = frags[blockIdx.x];
    const int lane = threadIdx.x >> 3, j = threadIdx.x & 7;
    int row;
    const TensorDesc t = tensors[d.tensor];
    const int col = frag_source(d, lane, j, t.n_out, &row);
    const float v = col < 0 ? 0.0f : weights[d.tensor][(int64_t)row * t.n_in + col];
    stream[(int64_t)blockIdx.x * 512 + threadIdx.x] = f32_to_bf16_rne(v);
}

__global__ __launch_bounds__(32) void pack_bias_bf16_kernel(const TileDesc *tiles, const TensorDesc *tensors,
                                                            PtrTable biases_tab, float *table) {
    const float *const *biases = biases_tab.p;
    const TileDesc t = tiles[blockIdx.x];
    const int h = threadIdx.x >> 4, r = threadIdx.x & 15;
    const int row = t.row0 + acc_row(r, h);
    table[blockIdx.x * 32 + threadIdx.x] = row < tensors[t.tensor].n_out ? biases[t.tensor][row] : 0.0f;
}

__global__ __launch_bounds__(16) void pack_bias_s16_kernel(const TileDesc *tiles, const TensorDesc *tensors,
                                                           PtrTable biases_tab, float *table) {
    const float *const *biases = biases_tab.p;
    const TileDesc t = tiles[blockIdx.x];
    const int row = t.row0 + threadIdx.x;           // [q][r] with row = 4q + r is just the natural order
    table[blockIdx.x * 16 + threadIdx.x] = row < tensors[t.tensor].n_out ? biases[t.tensor][row] : 0.0f;
}

// fp32 stream: one block per (layer, tile, group); 256 threads = 64 lanes x 4 k-pairs.
__global__ __launch_bounds__(256) void pack_f32_kernel(const LayerF32 *layers, int n_layers,
                                                       PtrTable weights_tab, PtrTable biases_tab,
                                                       float *stream, float *bias_out) {
    const float *const *weights = weights_tab.p;
    const float *const *biases = biases_tab.p;
    const LayerF32 L = layers[blockIdx.y];
    const int tiles = (L.n_out + 31) >> 5, groups = (L.n_in + 7) >> 3;
    for (int blk = blockIdx.x; blk < tiles * groups; blk += gridDim.x) {
        const int t = blk / groups, g = blk - t * groups;
        const int lane = threadIdx.x >> 2, i = threadIdx.x & 3;
        const int row = 32 * t + (lane & 31), col = 8 * g + 2 * i + (lane >> 5);
        const float v = (row < L.n_out && col < L.n_in) ? weights[L.tensor][(int64_t)row * L.n_in + col] : 0.0f;
        stream[L.frag_off + (int64_t)blk * 256 + threadIdx.x] = v;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < tiles * 32; i += gridDim.x * 256)
        bias_out[L.bias_off + i] = i < L.n_out ? biases[L.tensor][i] : 0.0f;
}

int launch_pack(const Program &p, const FragDesc *d_frags, const TileDesc *d_tiles, const LayerF32 *d_layers,
                const TensorDesc *d_tensors, const PtrTable &d_w, const PtrTable &d_b,
                uint16_t *stream_bf16, float *bias_bf16, float *stream_f32, float *bias_f32,
                const FragDesc *d_frags16, const TileDesc *d_tiles16, uint16_t *stream_s16, float *bias_s16,
                const FragDesc *d_frags_bwd, uint16_t *stream_bwd, hipStream_t s) {
    if (p.bf16_ok && !p.frags_bwd.empty())
        hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)p.frags_bwd.size()), dim3(512), 0, s,
                           d_frags_bwd, d_tensors, d_w, stream_bwd);
    if (p.bf16_ok) {
        hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)p.frags16.size()), dim3(512), 0, s,
                           d_frags16, d_tensors, d_w, stream_s16);
        hipLaunchKernelGGL(pack_bias_s16_kernel, dim3((unsigned)p.tiles16.size()), dim3(16), 0, s,
                           d_tiles16, d_tensors, d_b, bias_s16);
        hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)p.frags.size()), dim3(512), 0, s,
                           d_frags, d_tensors, d_w, stream_bf16);
        hipLaunchKernelGGL(pack_bias_bf16_kernel, dim3((unsigned)p.tiles.size()), dim3(32), 0, s,
                           d_tiles, d_tensors, d_b, bias_bf16);
    }
    hipLaunchKernelGGL(pack_f32_kernel, dim3(64, (unsigned)p.layers.size()), dim3(256), 0, s,
                       d_layers, (int)p.layers.size(), d_w, d_b, stream_f32, bias_f32);
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
