// kernels.h -- internal launch interfaces between the C ABI (capi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <vector>

#include "program.h"

namespace na {

// Arguments of both MLP kernels (fused bf16 and generic fp32).
struct MlpArgs {
    // packed parameters (device)
    const uint16_t *stream_bf16;   // bf16 A-fragment stream
    const float *bias_bf16;        // [n_tiles][2][16]
    const uint16_t *stream_s16;    // bf16 A-fragment stream for the 16x16x32 kernel
    const float *bias_s16;         // [n_tiles16][16]
    const uint16_t *stream_split;  // fp16 (hi, lo) A-fragment stream of the split-precision kernel (mlp_split.hip)
    const float *stream_f32;       // fp32 fragment stream
    const float *bias_f32;
    const LayerF32 *layers;        // device copy of the fp32 program
    int32_t n_layers;
    int32_t input_ch, input_ch_views, W, lds_rows;
    int32_t multires, multires_views, i_embed;
    // points: pre-embedded rows [P, input_ch + input_ch_views] (fp32 kernel only), or explicit
    // pts [P,3], or rays + z_vals (pts = o + d*z)
    const float *embedded;
    const float *pts;
    const float *rays;             // [R, ray_stride]: o(3) d(3) ...
    const float *z_vals;           // [R, S]
    const float *viewdirs;         // row stride vd_stride, NULL without view branch
    int32_t ray_stride, vd_stride;
    int64_t P;                     // number of points = R * S
    int32_t S;                     // samples per ray (ray index = p / S)
    int32_t out_ch;
    float *out;                    // [P, out_ch]
    unsigned *tile_ctr;            // mlp_bf16_s16p_kernel: {next ticket, workgroups done} of this launch (tile_counter_slot), or NULL:
                                   // tiles dealt blockIdx, blockIdx + gridDim, ... (the static deal)
    // training: every saved / gradient array below has pad_points(P) rows (the kernels store the rows of a
    // workgroup's padding points unconditionally, which keeps their store counts compile-time constants).
    // Activations saved by the forward for the backward pass, bf16, one row per point in
    // the k-slot order of the fragments ("slot-major": position 32*ks + 8*q + j of a row holds the
    // feature that slot (ks, q, j) of the B fragment holds; program.h acc16_col / gen16_col)
    uint16_t *sv_e;                // [P, 32*KE16]  encoded xyz
    uint16_t *sv_d;                // [P, 32*KD16]  encoded view direction
    uint16_t *sv_h;                // [8][P, 256]   outputs of pts_linears.0..7 (post-ReLU)
    uint16_t *sv_feat;             // [P, 256]      feature_linear output
    uint16_t *sv_hv;               // [P, 128]      views_linears.0 output (post-ReLU)
    uint8_t *sv_bits;              // [8][P, 32 B] + [P, 16 B]: "activation > 0" bit rows of pts_linears.0..7 and
                                   // views_linears.0 (lane quarter q owns bytes q*NK .. of a row; bit layout: mlp_bf16_s16.hip save_bits)
    // backward inputs / outputs
    const float *g_raw;            // [P, 4] dL/draw
    uint16_t *g_rawb;              // [P, 4] bf16 copy of g_raw (GEMM operand); output_linear models: [P, 16], columns >= out_ch zero
    uint16_t *g_rawt;              // the same values transposed inside 32-point chunks, [P/32][4 point groups][4 columns][8 points]:
                                   // the 16 bytes at (group g, column i) are lane (i, g)'s MFMA operand of the head products (backward.hip)
    uint16_t *g_hv;                // [P, 128] dL/d(pre-activation) of views_linears.0, slot-major
    uint16_t *g_feat;              // [P, 256]
    uint16_t *g_h;                 // [8][P, 256]  dL/d(pre-activation) of pts_linears.0..7
    const uint16_t *stream_bwd;    // transposed-weight fragment stream
    float *g_pts;                  // [P, 3]  dL/dpts (explicit-points mode), overwritten; or NULL
    float *g_rays;                 // [R, 6]  dL/d(origin, direction) accumulated with atomics (rays mode); or NULL
    float *g_vd;                   // [R, 3]  dL/d(view direction) accumulated with atomics; or NULL
    // split-precision training (split.h): every saved / gradient array above is the plane of fp16 hi rows, these are the
    // planes of fp16 lo rows (same shapes and slot order); the bit rows are shared
    uint16_t *sv_e_lo, *sv_d_lo, *sv_h_lo, *sv_feat_lo, *sv_hv_lo;
    uint16_t *g_rawt_lo, *g_hv_lo, *g_feat_lo, *g_h_lo;
    const uint16_t *stream_bwd_split;   // transposed-weight stream as fp16 (hi, lo) fragments
    float *g_scale;                // [GRAD_SCALE_PARTS + 2]: partial maxima of |dL/draw| (gmax_kernel), then the loss scale S
                                   // and 1 / S as the dX-chain kernel derived them (read by the slab reductions)
#ifdef NERF_AMD_STAMPS
    unsigned long long *stamps;    // diagnostic build: [workgroups][waves][4] cycle sums (mlp_bf16_s16.hip)
#endif
};

// Rows of the training arrays: P rounded up to whole 256-point workgroups.
constexpr int64_t pad_points(int64_t P) { return (P + 255) & ~(int64_t)255; }

// A {ticket, done} pair for one launch of a kernel that deals its tiles dynamically (zero between launches: the last
// workgroup to leave resets it).  Round robin over 1024 pairs per device, allocated by tile_counters_init.
unsigned *tile_counter_slot(int device);
// ... for a launch on stream s, or NULL (= the static deal) when the deal is off or s is being captured into a graph: a
// captured launch would bake its pair into every replay, and a replay may run beside an eager launch that drew the same pair
unsigned *tile_counter_for(bool deal, hipStream_t s);
int tile_counters_init(int device);
extern std::atomic<int> g_variant;    // nerf_amd_set_tuning key 0 (A/B selection; relaxed loads in the launchers)
bool mlp_bf16_supported(int multires, int multires_views, int use_viewdirs);
int launch_mlp_bf16(const MlpArgs &a, int multires, int multires_views, int use_viewdirs,
                    int n_frags_used, int n_tiles, hipStream_t s);
int launch_mlp_f32(const MlpArgs &a, hipStream_t s);
// mlp_split.hip: fp32-class results from fp16 operand pairs, three MFMAs per product
bool mlp_split_supported(int multires, int multires_views, int use_viewdirs, int out_ch);
int launch_mlp_split(const MlpArgs &a, int multires, int multires_views, int use_viewdirs, int n_frags_used, int n_tiles,
                     hipStream_t s);
int launch_mlp_split_save(const MlpArgs &a, int multires, int multires_views, int use_viewdirs, int n_frags_used, int n_tiles,
                          hipStream_t s);
// mlp_bwd_split.hip: the dX chain on fp16 pairs; launches gmax_kernel (the loss scale) first
int launch_mlp_bwd_split(const MlpArgs &a, int multires, int multires_views, int use_viewdirs, int n_frags_used, hipStream_t s);
int launch_embed(const float *x, int64_t n, int multires, float *out, hipStream_t s);

// Parameter pointers of one model, passed to the pack kernels by value (no host->device copy).
constexpr int MAX_TENSORS = 72;     // D <= 64 pts_linears + 4 heads
struct PtrTable { const float *p[MAX_TENSORS]; };

// pack.hip
int launch_pack(const Program &p, const FragDesc *d_frags, const TileDesc *d_tiles, const LayerF32 *d_layers,
                const TensorDesc *d_tensors, const PtrTable &d_weight_ptrs, const PtrTable &d_bias_ptrs,
                uint16_t *stream_bf16, float *bias_bf16, float *stream_f32, float *bias_f32,
                const FragDesc *d_frags16, const TileDesc *d_tiles16, uint16_t *stream_s16, float *bias_s16,
                const FragDesc *d_frags_bwd, uint16_t *stream_bwd, const FragDesc *d_frags_split, uint16_t *stream_split,
                const FragDesc *d_frags_bwd_split, uint16_t *stream_bwd_split, const TrainLayerF32 *d_tlayers, float *stream_f32_t, int copies, hipStream_t s);
void pack_bf16_host(const Program &p, int shape, const float *const *w, const float *const *b, uint16_t *stream, float *bias);
int launch_mlp_bf16_s16(const MlpArgs &a, int multires, int multires_views, int use_viewdirs,
                        int n_frags_used, int n_tiles, hipStream_t s);
// training kernels: the view-branch model with multires 10/4 or 15/6, the output_linear model with multires 10 or 15
int launch_mlp_bf16_s16_save(const MlpArgs &a, int multires, int multires_views, int use_viewdirs, int n_frags_used, int n_tiles, hipStream_t s);
int launch_mlp_bwd_s16(const MlpArgs &a, int multires, int multires_views, int use_viewdirs, int n_frags_used, hipStream_t s);

// backward.hip
// split: the arrays of NERF_AMD_PREC_FP32_SPLIT training (hi and lo planes, loss-scale slots) instead of the bf16 ones
// train_f32.hip: exact-fp32 training for any architecture (NERF_AMD_PREC_FP32): forward with saves, dX chain, dW / db
bool train_f32_supported(const Program &p);
int64_t train_f32_workspace_bytes(const Program &p, int64_t P);
int launch_train_f32_forward(const Program &p, const MlpArgs &a, const TrainLayerF32 *d_tl, void *workspace, hipStream_t s);
int launch_train_f32_backward(const Program &p, const MlpArgs &a, const TrainLayerF32 *d_tl, const float *stream_t, void *workspace,
                              float *const *gw, float *const *gb, hipStream_t s);
bool train_supported(const Program &p);
int64_t train_workspace_bytes(const Program &p, int64_t P, bool split);
void train_fill_args(const Program &p, int64_t P, void *workspace, MlpArgs *a, bool split);
int train_param_grads(const Program &p, int64_t P, void *workspace, float *const *gw, float *const *gb, int device, hipStream_t s,
                      bool split, const float *g_raw);

// capi.hip: the library's side stream of a device and a pool of timing-less events (used by nerf_amd_render_batch and
// by the weight-gradient products, whose slab reductions run beside the next product)
int lane_acquire(int device, int n_events, hipStream_t *side, std::vector<hipEvent_t> *events);
void lane_release(int device, const std::vector<hipEvent_t> &events);
int lane_streams(int device, int n, hipStream_t *out);      // the device's side stream and up to two more

int launch_img2mse(const float *x, const float *y, int64_t n, float *out, float *partials, hipStream_t s);
int launch_img2mse_bwd(const float *x, const float *y, int64_t n, const float *g, float *gx, float *gy, hipStream_t s);
int launch_assemble_rays(const float *o, const float *d, const float *v, int64_t n, float near, float far, float *out, hipStream_t s);

// adam.hip
int launch_adam(int n, float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                const int64_t *numel, float step_size, double beta1, double beta2, float eps, float weight_decay,
                float bc2_sqrt, hipStream_t s, int64_t *step_dev = nullptr, const double *lr_dev = nullptr, float *scalars_dev = nullptr);

// render.hip
struct RenderCfgK {
    int32_t Nc, Ni, perturb, lindisp, white_bkgd;
};
int launch_coarse_z(const float *rays, int ray_stride, const float *t_vals, const float *t_rand,
                    int64_t R, int Nc, int lindisp, int perturb, float *z, hipStream_t s);
int launch_composite(const float *raw, int raw_ch, const float *z, const float *rays_d, int rays_d_stride,
                     const float *noise, int64_t R, int S, int white_bkgd, float *rgb, float *disp, float *acc,
                     float *weights, float *depth, hipStream_t s);
int launch_composite_bwd(const float *raw, int raw_ch, const float *z, const float *rays_d, int rays_d_stride,
                         const float *noise, int64_t R, int S, int white_bkgd, const float *g_rgb, const float *g_disp,
                         const float *g_acc, const float *g_depth, const float *g_weights, float *g_raw, float *g_rays_d,
                         hipStream_t s);
int launch_sample_pdf(const float *bins, const float *weights, const float *u, const float *t_lin,
                      int64_t R, int n_bins, int n_samples, float *samples, hipStream_t s);
int launch_resample(const float *z_coarse, const float *weights, const float *u, const float *t_lin,
                    int64_t R, int Nc, int Ni, float *z_fine, float *z_std, hipStream_t s);
// dynamic LDS the per-ray kernels ask for (4 rays per workgroup); launches beyond LDS_LIMIT_BYTES are refused
constexpr size_t LDS_LIMIT_BYTES = 160 * 1024;
size_t composite_bwd_lds_bytes(int S);
size_t sample_pdf_lds_bytes(int n_bins);
size_t resample_lds_bytes(int Nc, int Ni, bool with_composite);
// one compositing job (raw2outputs over R rays of S samples) and the resampling that may follow it
struct CompositeJob {
    const float *raw; int raw_ch; const float *z; const float *rays_d; int rays_d_stride; const float *noise;
    int64_t R; int S; int white_bkgd;
    float *rgb, *disp, *acc, *weights;
};
struct ResampleJob {
    const float *u, *t_lin; int Ni, pad_c, pad_s;
    float *z_fine, *z_std;
};
int launch_mid_stage(const CompositeJob &coarse, const ResampleJob &resample, const CompositeJob *final_of_previous_chunk,
                     hipStream_t s);
int launch_ndc_rays_bwd(int H, int W, double focal, float near, const float *rays_o, const float *rays_d, const float *g_oo,
                        const float *g_od, int64_t n, float *g_ro, float *g_rd, hipStream_t s);
int launch_ndc_rays(int H, int W, double focal, float near, const float *rays_o, const float *rays_d, int64_t n,
                    float *out_o, float *out_d, hipStream_t s);
int launch_get_rays_bwd(int H, int W, const double *K4, int64_t pix0, int64_t n, const float *g_o, const float *g_d,
                        float *g_c2w, hipStream_t s);
int launch_to8b(const float *x, int64_t n, uint8_t *out, hipStream_t s);
int launch_make_rays(int H, int W, const double *K4, const float *c2w, const float *c2w_static,
                     int64_t pix0, int64_t n, float near, float far, int use_viewdirs, int ndc,
                     float *rays_out, hipStream_t s);

}  // namespace na
