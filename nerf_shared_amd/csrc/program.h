// program.h -- how one reference NeRF (nerf.py:62-134) is laid out as MFMA
// fragment streams.  Shared by the host packer, the device pack kernel and
// the kernels that consume the streams.
//
// bf16 stream (fused kernel, mlp_bf16.hip)
//   The network is evaluated transposed: H_out^T[out, pts] = W[out, in] . H_in^T[in, pts],
//   so weights are the MFMA *A* operand and 32 points sit on the lanes of a wave
//   as the *B* operand.  The stream is the exact sequence of 1-KiB A fragments
//   (v_mfma_f32_32x32x16_bf16: lane l holds A[row = l&31][k = 8*(l>>5) + j], j<8)
//   in consumption order: layer -> 32-row output tile -> k-step.
//   Two k orders exist inside a k-step:
//    * FRAG_ACC: the B operand is the previous layer's accumulator tile converted
//      to bf16 in place (no lane movement).  Accumulator register r of lane
//      (col, h) holds feature row (r&3) + 8*(r>>2) + 4*h of its 32-row tile, so
//      k slot (h, j) of k-step ks means input column
//          32*(ks>>1) + 16*(ks&1) + 8*(j>>2) + 4*h + (j&3).
//    * FRAG_GEN: the B operand is a positional encoding generated in registers.
//      Lane half h = 0 evaluates the sin features, h = 1 the cos features, so
//      with e = 8*ks + j:  e < 3L  -> column 3 + 6*(e/3) + 3*h + e%3
//                          e = 3L  -> column 0 (h=0) / 2 (h=1)      (raw x / z)
//                          e = 3L+1-> column 1 (h=0) / none (h=1)   (raw y)
//      (column = index into the reference embedding, nerf.py:16-41).
//
// bf16 "s16" stream (mlp_bf16_s16.hip): same idea on v_mfma_f32_16x16x32_bf16
//   (lane l holds A[row = l&15][k = 8*(l>>4) + j]; C/D: col = l&15, row = 4*(l>>4) + r, r<4).
//   A wave still owns 32 points, as two 16-column tiles.  Fragments are 16 out rows x 32 k
//   (1 KiB), ordered layer -> pair of output tiles -> k-step -> tile of the pair.
//    * FRAG_ACC16: k-step ks is fed by the accumulators of output tiles 2ks and 2ks+1 of the
//      previous layer, so k slot (q = l>>4, j) means input column 32*ks + 16*(j>>2) + 4*q + (j&3).
//    * FRAG_GEN16: lane quarter q = 2h + b evaluates sin (h=0) or cos (h=1) of the frequencies
//      of parity b; see gen16_col below.
//
// backward stream (mlp_bwd_s16.hip): the same s16 machinery applied to g_in^T = W^T . g_out^T.
//   Fragments are 16 rows of W^T (= 16 input features of the Linear) x 32 k (= output features):
//    * FRAG_T16 : B operand is a gradient tile chain, k slot (q, j) of k-step ks is output
//                 feature acc16_col(ks, q, j);  element = W[that output][col_base + row0 + (l&15)].
//    * FRAG_TG16: B operand is built from dL/draw, k slot (q, j) is output feature 8q + j.
//    * FRAG_TE16: like FRAG_T16, but the 16 rows of the tile are slots of a generated encoding
//                 (row 16t + 4q + r = slot (ks = t>>1, q, j = 4(t&1) + r) of gen16_col), so every lane
//                 receives the gradients of exactly the encoding values it generated in the forward.
//   FragDesc: row0 = first input feature (or encoding slot) of the tile (relative to col_base),
//   seg_len = number of output features, L = number of input features of this segment
//   (FRAG_TE16: the multires of the encoding).
//
// fp32 stream (generic kernel, mlp_fp32.hip)
//   v_mfma_f32_32x32x2_f32: lane l holds A[row = l&31][k = l>>5].  Fragments are
//   grouped four k-pairs at a time so a lane loads 16 bytes: group g of tile t
//   holds W[32t + (l&31)][8g + 2i + (l>>5)], i < 4.
#pragma once
#include <stdint.h>
#include <vector>

#include "../../include/nerf_amd.h"

#if defined(__HIPCC__)
#define NA_HD __host__ __device__
#else
#define NA_HD
#endif

namespace na {

enum { FRAG_ACC = 0, FRAG_GEN = 1, FRAG_ZERO = 2, FRAG_ACC16 = 3, FRAG_GEN16 = 4, FRAG_T16 = 5, FRAG_TG16 = 6, FRAG_TE16 = 7 };

struct FragDesc {        // one bf16 A fragment: 32 out rows x 16 k
    int32_t tensor;      // index into the parameter list (nerf_amd.h order)
    int32_t kind;        // FRAG_*
    int32_t row0;        // first output row of the tile
    int32_t col_base;    // first weight column of this input segment
    int32_t ks;          // k-step inside the segment
    int32_t seg_len;     // valid columns in the segment
    int32_t L;           // multires of the generated encoding (FRAG_GEN)
    int32_t part;        // 0: bf16(w).  Split stream (mlp_split.hip): 1 = fp16 hi part of w, 2 = fp16 lo part, (w - hi) 2^11
};

struct TileDesc {        // one 32-row output tile (bias table entry)
    int32_t tensor;
    int32_t row0;
};

struct TensorDesc {      // one nn.Linear
    int32_t n_out, n_in;
};

// fp32 generic program: one entry per layer, executed in order by mlp_fp32.hip
struct LayerF32 {
    int32_t tensor;
    int32_t n_out, n_in;     // true sizes
    int32_t in_row;          // first LDS row of the input  (rows are features)
    int32_t out_row;         // first LDS row of the output, or -1: write to global
    int32_t out_col;         // channel offset in the global output row
    int32_t relu;
    int32_t in_buf;          // 0/1 ping-pong buffer the input is read from
    int64_t frag_off;        // float offset of this layer's fragments in the fp32 stream
    int64_t bias_off;        // float offset of the bias (padded to 32*tiles)
};

// The exact-fp32 training path (train_f32.hip: any architecture the constructor accepts): what the backward needs to know
// about layer l of the program beside its LayerF32.  Workspace arrays are [rows][pad64(P)] fp32, rows counted over the layers.
struct TrainLayerF32 {
    int64_t frag_off_t;      // float offset of the layer's TRANSPOSED fragments (tiles over n_in, k over n_out) in stream_f32_t
    int32_t x_row;           // first workspace row of the saved input  X_l   [n_in][Pp]
    int32_t g_row;           // first workspace row of dL/d(pre-activation)   [n_out][Pp]
    int32_t y_row;           // first workspace row of the saved post-ReLU output [n_out][Pp], -1 without ReLU
    int32_t lo, hi;          // rows [lo, hi) of the INPUT (relative to in_row) are hidden features: their gradient goes on
    int32_t accumulate;      // 1: another consumer of the same input was processed earlier in the backward order: +=
    int32_t lds_g_row;       // LDS row of this layer's output gradient in the backward kernel (heads: behind the forward's rows)
};

// Column of the reference embedding that slot (ks,h,j) of a FRAG_GEN segment holds, or -1.
NA_HD constexpr inline int gen_col(int ks, int h, int j, int L) {
    int e = 8 * ks + j;
    if (e < 3 * L) return 3 + 6 * (e / 3) + 3 * h + (e % 3);
    if (e == 3 * L) return h ? 2 : 0;
    if (e == 3 * L + 1) return h ? -1 : 1;
    return -1;
}
// Column (inside its segment) that slot (ks,h,j) of a FRAG_ACC segment holds.
NA_HD constexpr inline int acc_col(int ks, int h, int j) {
    return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
}
// Feature row that accumulator register r of lane half h holds (32x32 C/D layout).
NA_HD constexpr inline int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

NA_HD constexpr inline int gen_ksteps(int L) { return (3 * L + 2 + 7) / 8; }   // k-steps of a generated encoding

// --- 16x16x32 ("s16") layout
// Generated encodings: lane quarter q = 2h + b.  h picks sin (0) / cos (1); b picks the
// frequency parity, so that every lane walks f = b, b+2, b+4, ... with one exact x4 step.
// Slot i = 8*ks + j of group (h,b) holds trig feature (f = 2*(i/3) + b, coordinate i%3) while
// i < gen16_ntrig(L,b); the free slots after that hold the raw coordinates x,y,z, handed
// out in the fixed group order (h,b) = (0,1), (1,1), (0,0), (1,0).
NA_HD constexpr inline int gen16_ksteps(int L) { return (3 * L + 2 + 15) / 16; }
NA_HD constexpr inline int gen16_ntrig(int L, int b) { return 3 * ((L - b + 1) / 2); }
NA_HD constexpr inline int gen16_misc(int L, int h, int b, int m) {
    const int cap = 8 * gen16_ksteps(L);
    const int g = b ? h : 2 + h;                      // position in the hand-out order
    int offset = 0;
    for (int gg = 0; gg < g; ++gg) offset += cap - gen16_ntrig(L, gg < 2 ? 1 : 0);
    const int idx = offset + m;
    return idx < 3 ? idx : -1;
}
NA_HD constexpr inline int gen16_col(int ks, int q, int j, int L) {
    const int h = q >> 1, b = q & 1, i = 8 * ks + j, n = gen16_ntrig(L, b);
    if (i < n) return 3 + 6 * (2 * (i / 3) + b) + 3 * h + (i % 3);
    return gen16_misc(L, h, b, i - n);
}
NA_HD constexpr inline int acc16_col(int ks, int q, int j) { return 32 * ks + 16 * (j >> 2) + 4 * q + (j & 3); }

inline int embed_dim(int L, int i_embed) { return i_embed == -1 ? 3 : 3 + 6 * L; }

struct Program {
    nerf_amd_arch arch;
    int input_ch = 0, input_ch_views = 0, out_ch = 0;
    std::vector<TensorDesc> tensors;
    // bf16 fused program (empty when the architecture is not the canonical one)
    bool bf16_ok = false;
    int KE = 0, KD = 0;
    std::vector<FragDesc> frags;     // padded to a whole number of ring turns
    std::vector<TileDesc> tiles;
    int n_frags_used = 0;
    // bf16 s16 program (16x16x32 MFMA); tiles16 are 16-row tiles
    int KE16 = 0, KD16 = 0;
    std::vector<FragDesc> frags16;
    std::vector<TileDesc> tiles16;
    int n_frags16_used = 0;
    // split-precision stream (mlp_split.hip): the s16 program with every fragment as an fp16 (hi, lo) pair, ordered
    // layer -> pair of output tiles -> k-step -> part -> tile of the pair; the bias table is tiles16's
    std::vector<FragDesc> frags_split;
    int n_frags_split_used = 0;
    // backward (transposed-weight) stream for the s16 kernel; view-branch (10,4) model only
    std::vector<FragDesc> frags_bwd;
    int n_frags_bwd_used = 0;
    // the same transposed stream as fp16 (hi, lo) pairs for the split-precision dX chain (mlp_bwd_split.hip): per pair of
    // tiles and k-step four fragments hi(t0), hi(t1), lo(t0), lo(t1), like frags_split
    std::vector<FragDesc> frags_bwd_split;
    int n_frags_bwd_split_used = 0;
    // fp32 generic program
    std::vector<LayerF32> layers;
    int64_t f32_stream_floats = 0, f32_bias_floats = 0;
    int lds_rows = 0;                // rows of one ping-pong buffer
    // exact-fp32 training (train_f32.hip)
    std::vector<TrainLayerF32> tlayers;
    int64_t f32_stream_t_floats = 0; // the transposed fragment stream
    int train_f32_rows = 0;          // workspace rows in all
    int lds_rows_bwd = 0;            // LDS rows of the backward kernel: lds_rows + 32 for the heads' output gradients
};

constexpr int STREAM_PAD_FRAGS = 192;   // the bf16 stream is zero-padded to a multiple of this (any block size <= 64)

int build_program(const nerf_amd_arch &arch, Program &p, const char **err);

}  // namespace na
