/*
 * nerf_amd.h -- C ABI of the MI355X (gfx950) NeRF volumetric-render hot path.
 *
 * Drop-in boundary for stanford-iprl-lab/nerf_shared's
 *   Renderer.render -> render_batch -> render_rays -> {NeRF.forward/MLP,
 *   raw2outputs, sample_pdf}
 * The reference is pure Python/PyTorch and has no FFI layer of its own
 * (SURVEY.md section 8b), so every entry point below names the Python
 * function (reference file:line) whose arithmetic it replaces; the Python
 * package nerf_shared_amd binds them with ctypes and keeps the reference's
 * class/function signatures.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch/C++ types.
 *  - Every data pointer is a DEVICE pointer to contiguous fp32 (row-major,
 *    innermost dimension last) unless a parameter says "host".  Buffers are
 *    borrowed: the caller (torch) owns them; the library never frees them.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream()
 *    .cuda_stream); all work is enqueued on it, nothing synchronises the host
 *    except nerf_amd_model_create/_update (which read host weights).
 *  - Return value: 0 on success, a negative NERF_AMD_E* code on failure;
 *    nerf_amd_last_error() returns a thread-local message.  No exception
 *    crosses the boundary.
 *  - Re-entrant; the only shared state is inside model handles, which must not
 *    be updated while a launch that uses them is being enqueued.  What the
 *    launchers cache (raised LDS limits, CU counts) is kept per device and
 *    updated atomically (csrc/launch_util.h).
 */
#ifndef NERF_AMD_H
#define NERF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_AMD_ABI_VERSION 7

#define NERF_AMD_OK            0
#define NERF_AMD_EINVAL       -1   /* bad argument / unsupported shape        */
#define NERF_AMD_EHIP         -2   /* a HIP runtime call failed               */
#define NERF_AMD_EUNSUPPORTED -3   /* architecture not supported by this path */
#define NERF_AMD_ENOMEM       -4

/* Arithmetic of the MLP (the only stage with a precision choice). */
#define NERF_AMD_PREC_FP32 0       /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32): parity mode       */
#define NERF_AMD_PREC_BF16 1       /* bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16;   */
                                   /* 32x32x16 for output_ch > 16 without a view branch)          */
#define NERF_AMD_PREC_FP32_SPLIT 2 /* fp32-class results on the 16-bit matrix pipe: operands as   */
                                   /* fp16 (hi, lo) pairs, three v_mfma_f32_16x16x32_f16 per      */
                                   /* product, fp32 accumulate (csrc/mlp_split.hip); needs        */
                                   /* |activation| < 65504                                        */

#define NERF_AMD_MAX_SKIPS 8

int         nerf_amd_abi_version(void);
const char *nerf_amd_last_error(void);

/* ------------------------------------------------------------------------
 * Model handle: packed weights of one reference NeRF.
 * Replaces: NeRF.__init__ parameter layout, /root/reference/nerf_shared/nerf.py:62-94.
 * ------------------------------------------------------------------------ */
typedef struct nerf_amd_arch {
    int32_t D;                 /* number of pts_linears                   (nerf.py:62)  */
    int32_t W;                 /* hidden width                                          */
    int32_t output_ch;         /* width of output_linear when !use_viewdirs             */
    int32_t use_viewdirs;
    int32_t multires;          /* L of the xyz encoding (input_ch = 3 + 6 L)            */
    int32_t multires_views;    /* L of the view-direction encoding                      */
    int32_t i_embed;           /* 0 = positional encoding, -1 = identity  (nerf.py:44)  */
    int32_t n_skips;
    int32_t skips[NERF_AMD_MAX_SKIPS];
} nerf_amd_arch;

typedef struct nerf_amd_model nerf_amd_model;   /* opaque, library-owned */

/*
 * Parameters are given in state_dict order as DEVICE pointers to the live fp32
 * nn.Linear tensors ([out,in] row-major weights, [out] biases):
 *   index 0..D-1 : pts_linears.i
 *   viewdirs     : D = feature_linear, D+1 = alpha_linear, D+2 = views_linears.0, D+3 = rgb_linear
 *   no viewdirs  : D = output_linear
 * The library re-packs them on the device (bf16 MFMA A-fragment stream and an
 * fp32 fragment stream) on `stream`; call nerf_amd_model_update again after the
 * parameters change (the Python shim keys this on tensor._version/data_ptr).
 */
int  nerf_amd_model_create(const nerf_amd_arch *arch, int device, nerf_amd_model **out);
int  nerf_amd_model_update(nerf_amd_model *m, const float *const *weights, const float *const *biases,
                           int n_tensors, void *stream);
/*
 * The same for some of the packed copies only (a training step in one precision needs two of the six; re-packing all
 * of them after every optimizer step is 22 us per model of a 1.6-ms step).  `copies`: bit-or of NERF_AMD_COPY_*.
 * `others_current` != 0: the parameters are the ones the OTHER copies were packed from (a copy is being added for the
 * same weights) -- they stay valid; 0: the parameters changed, the other copies become stale.  Every entry point
 * fails with NERF_AMD_EINVAL when the copy it needs is stale or was never packed.  nerf_amd_model_update =
 * NERF_AMD_COPY_ALL.
 */
#define NERF_AMD_COPY_BF16      1   /* bf16 fragment streams + bias tables: NERF_AMD_PREC_BF16 inference and training forward */
#define NERF_AMD_COPY_BWD       2   /* transposed bf16 stream: nerf_amd_field_backward in NERF_AMD_PREC_BF16 */
#define NERF_AMD_COPY_SPLIT     4   /* fp16 (hi, lo) pair stream + bias table: NERF_AMD_PREC_FP32_SPLIT */
#define NERF_AMD_COPY_BWD_SPLIT 8   /* transposed pair stream: nerf_amd_field_backward in split precision */
#define NERF_AMD_COPY_FP32      16  /* fp32 fragment stream + biases: NERF_AMD_PREC_FP32 (inference and training forward), nerf_amd_mlp_embedded */
#define NERF_AMD_COPY_FP32_BWD  32  /* transposed fp32 stream: nerf_amd_field_backward in NERF_AMD_PREC_FP32 */
#define NERF_AMD_COPY_ALL       63
int  nerf_amd_model_update_copies(nerf_amd_model *m, const float *const *weights, const float *const *biases,
                                  int n_tensors, int copies, int others_current, void *stream);
void nerf_amd_model_destroy(nerf_amd_model *m);
/* 1 when the fused bf16 kernel supports this architecture (D=8, W=256, skips=[4]). */
int  nerf_amd_model_supports_bf16(const nerf_amd_model *m);
/* 1 when the split-precision kernel (NERF_AMD_PREC_FP32_SPLIT) supports this architecture: the same set. */
int  nerf_amd_model_supports_split(const nerf_amd_model *m);
int  nerf_amd_model_out_ch(const nerf_amd_model *m);       /* 4 with viewdirs, else output_ch */

/* Host-side packing of one model into the bf16 fragment stream (test hook: lets
 * CPU tests check the fragment layout without a GPU).  Weights/biases are HOST
 * pointers here.  `stream_out` receives n_frags*512 uint16 (bf16 bits) and
 * `bias_out` the fp32 bias table; pass NULL to query sizes only. */
int  nerf_amd_pack_bf16_host(const nerf_amd_arch *arch, int shape /* 32: 32x32x16 stream, 16: 16x16x32 stream, 17: transposed
                                (backward) stream, 18 / 19: the split-precision forward / transposed streams (fp16 hi, lo fragments) */,
                             const float *const *weights,
                             const float *const *biases, int n_tensors,
                             uint16_t *stream_out, int64_t *n_frags, float *bias_out, int64_t *n_bias);

/* ------------------------------------------------------------------------
 * a1  Embedder.embed / get_embedder          nerf.py:16-58
 * x [n,3] -> out [n, 3+6*multires]   (frequency-major, sin then cos, xyz innermost)
 * ------------------------------------------------------------------------ */
int nerf_amd_embed(const float *x, int64_t n, int multires, float *out, void *stream);

/* ------------------------------------------------------------------------
 * a3/a4  NeRF.forward + NeRF.MLP              nerf.py:96-134
 * pts [n_rays*n_samples,3]; viewdirs [n_rays,3] (NULL when the model has no
 * view branch; every sample of ray r uses viewdirs[r], nerf.py:101)
 * -> out [n_rays*n_samples, out_ch].  No netchunk: the kernel tiles internally.
 * ------------------------------------------------------------------------ */
int nerf_amd_nerf_forward(const nerf_amd_model *m, const float *pts, const float *viewdirs,
                          int64_t n_rays, int32_t n_samples, float *out, int precision, void *stream);

/* ------------------------------------------------------------------------
 * a4  NeRF.MLP on already-embedded rows       nerf.py:110-134
 * x [n, input_ch + input_ch_views] -> out [n, out_ch]; exact-fp32 kernel.
 * ------------------------------------------------------------------------ */
int nerf_amd_mlp_embedded(const nerf_amd_model *m, const float *x, int64_t n, float *out, void *stream);

/* ------------------------------------------------------------------------
 * a13  utils.ndc_rays on explicit rays        utils.py:54-71
 * rays_o, rays_d [n,3] -> out_o, out_d [n,3]; focal is the Python float K[0][0].
 * ------------------------------------------------------------------------ */
int nerf_amd_ndc_rays(int32_t H, int32_t W, double focal, float near, const float *rays_o, const float *rays_d,
                      int64_t n, float *out_o, float *out_d, void *stream);

/* Backward of nerf_amd_ndc_rays (pose estimation on forward-facing scenes differentiates through the warp):
 * gradients of its two outputs [n,3] (either may be NULL = zero) -> gradients of rays_o / rays_d [n,3]
 * (either may be NULL = not wanted), overwritten. */
int nerf_amd_ndc_rays_backward(int32_t H, int32_t W, double focal, float near, const float *rays_o, const float *rays_d,
                               const float *g_out_o, const float *g_out_d, int64_t n, float *g_rays_o, float *g_rays_d,
                               void *stream);

/* ------------------------------------------------------------------------
 * a10  Renderer.raw2outputs                   render_utils.py:241-290
 * raw [R,S,raw_ch] (channels 0..2 rgb, 3 sigma), z_vals [R,S], rays_d with row
 * stride `rays_d_stride` floats, noise [R,S] or NULL (already scaled by
 * raw_noise_std).  Any output pointer may be NULL.
 * ------------------------------------------------------------------------ */
int nerf_amd_raw2outputs(const float *raw, int32_t raw_ch, const float *z_vals,
                         const float *rays_d, int32_t rays_d_stride, const float *noise,
                         int64_t R, int32_t S, int white_bkgd,
                         float *rgb_map, float *disp_map, float *acc_map, float *weights,
                         float *depth_map, void *stream);

/* Backward of raw2outputs (what autograd derives from render_utils.py:241-290): upstream gradients of
 * the five outputs (any may be NULL) -> g_raw [R,S,raw_ch] and, if asked, g_rays_d [R,3] (the
 * dependence of dists on |rays_d|, :259).  z_vals are constants.  SURVEY.md section 8f-1. */
int nerf_amd_raw2outputs_backward(const float *raw, int32_t raw_ch, const float *z_vals, const float *rays_d,
                                  int32_t rays_d_stride, const float *noise, int64_t R, int32_t S, int white_bkgd,
                                  const float *g_rgb_map, const float *g_disp_map, const float *g_acc_map,
                                  const float *g_depth_map, const float *g_weights, float *g_raw,
                                  float *g_rays_d /* [R,3] or NULL */, void *stream);

/* ------------------------------------------------------------------------
 * Training (what loss.backward() does through NeRF.forward, main.py:85-104).
 * Covered: the D=8, W=256, skips=[4] model with view branch and multires 10/4 or 15/6 (every config the
 * reference ships), or without view branch (the output_linear models of nerf.py:91-94,131-132 -- the default
 * use_viewdirs=False of config_parser.py:50 -- multires 10 or 15, output_ch <= 16: raw / g_raw are [P, output_ch],
 * rays are [R,8], viewdirs NULL); gradients of the parameters,
 * of the points / rays (through the positional encoding) and of the view directions.
 * `precision` (the same value in the three calls of one evaluation):
 *   NERF_AMD_PREC_BF16        bf16 operands / fp32 accumulation (the fast default)
 *   NERF_AMD_PREC_FP32_SPLIT  the reference's arithmetic class: every operand of the forward, of the dX chain and of the
 *                             weight-gradient products an fp16 (hi, lo) pair, three MFMAs per product, fp32 accumulation;
 *                             dL/draw is scaled by a power of two taken from its own maximum (csrc/split.h) and the scale
 *                             comes off exactly at the end.  Gradients agree with fp32 autograd to ~1e-6 relative.
 *   NERF_AMD_PREC_FP32        exact fp32 (v_mfma_f32_32x32x2_f32 everywhere, csrc/train_f32.hip) for ANY architecture
 *                             nerf_amd_model_create accepts -- what the reference's netdepth / netwidth / skips flags build
 *                             (config_parser.py:18-25); the same gradients (parameters, points / rays, view directions)
 *                             within 1e-6 of fp32 autograd, at the fp32 MFMA rate: the fallback for models outside the
 *                             family above, not a tuned path.  nerf_amd_model_supports_training(m, precision) says which
 *                             precisions a model trains in.
 *   forward_train : the fused forward (explicit pts + viewdirs, or rays + z_vals with pts = o + d z) that also saves every
 *                   layer's activations in `workspace` (nerf_amd_train_workspace bytes, 256-B aligned)
 *   backward      : dL/draw [P,4] -> gradients of every nn.Linear weight [out,in] and bias [out]
 *                   (fp32 device tensors in nerf_amd_model_update order, overwritten)
 * ------------------------------------------------------------------------ */
int     nerf_amd_model_supports_training(const nerf_amd_model *m, int precision);
int64_t nerf_amd_train_workspace(const nerf_amd_model *m, int64_t n_points, int precision);
int     nerf_amd_field_forward_train(const nerf_amd_model *m, const float *pts /* [R*S,3] or NULL */,
                                     const float *viewdirs /* [R,3], with pts */, const float *rays /* [R,11|8], without pts */,
                                     int32_t ray_ch, const float *z_vals, int64_t R, int32_t S, float *raw,
                                     void *workspace, int64_t workspace_bytes, int precision, void *stream);
int     nerf_amd_field_backward(const nerf_amd_model *m, const float *g_raw /* [R*S,4|output_ch] */,
                                const float *pts, const float *viewdirs, const float *rays, int32_t ray_ch,
                                const float *z_vals, int64_t R, int32_t S /* the forward_train inputs */,
                                void *workspace, int64_t workspace_bytes, float *const *grad_weights,
                                float *const *grad_biases, int n_tensors,
                                float *g_pts /* [R*S,3] overwritten, pts mode, or NULL */,
                                float *g_rays /* [R,6] dL/d(o,d), accumulated (pass zeros), rays mode, or NULL */,
                                float *g_viewdirs /* [R,3] accumulated (pass zeros), or NULL */, int precision, void *stream);

/* ------------------------------------------------------------------------
 * a11  utils.sample_pdf                       utils.py:74-117
 * bins [R,n_bins], weights [R,n_bins-1], u [R,n_samples] or NULL (then
 * u = t_lin[n_samples], the caller's torch.linspace(0,1,n_samples), det=True).
 * -> samples [R,n_samples]
 * ------------------------------------------------------------------------ */
int nerf_amd_sample_pdf(const float *bins, const float *weights, const float *u, const float *t_lin,
                        int64_t R, int32_t n_bins, int32_t n_samples, float *samples, void *stream);

/* ------------------------------------------------------------------------
 * a9  Renderer.render_rays                    render_utils.py:67-174
 * One call enqueues the whole two-pass pipeline for R rays:
 *   z_vals (lin / lindisp, optional stratified jitter) -> coarse field ->
 *   composite -> sample_pdf -> sort(cat) -> fine field -> composite.
 * ------------------------------------------------------------------------ */
/* The two per-ray stages of render_rays that have no Python-level counterpart of their own, exported
 * for the training path (which runs render_rays stage by stage so autograd can cut in):
 *   coarse_z : z_vals of the coarse pass, lin / lindisp + stratified jitter   render_utils.py:105-129
 *   resample : z_mid -> sample_pdf(weights[1:-1]) -> z_std -> sort(cat)       render_utils.py:140-148,168 */
int nerf_amd_coarse_z(const float *rays, int32_t ray_ch, const float *t_vals, const float *t_rand, int64_t R,
                      int32_t N_samples, int lindisp, int perturb, float *z_vals, void *stream);
int nerf_amd_resample(const float *z_coarse, const float *weights, const float *u, const float *t_lin, int64_t R,
                      int32_t N_samples, int32_t N_importance, float *z_fine, float *z_std, void *stream);

typedef struct nerf_amd_render_cfg {
    int32_t N_samples;
    int32_t N_importance;
    int32_t perturb;           /* perturb > 0                               (:115) */
    int32_t lindisp;
    int32_t white_bkgd;
    int32_t use_noise;         /* raw_noise_std > 0: noise0/noise1 are read (:262) */
    int32_t precision;         /* NERF_AMD_PREC_*                                  */
    int32_t reserved;
} nerf_amd_render_cfg;

typedef struct nerf_amd_render_io {
    /* inputs */
    const float *rays;         /* [R, ray_ch]: o(3) d(3) near far [viewdirs(3)]  (:98-103) */
    int32_t      ray_ch;       /* 8 or 11                                                  */
    int32_t      pad0;
    const float *t_vals;       /* [N_samples]  torch.linspace(0,1,N_samples)      (:105)   */
    const float *t_rand;       /* [R,N_samples] jitter draws, NULL unless perturb (:121)   */
    const float *noise0;       /* [R,N_samples] scaled sigma noise, coarse pass   (:264)   */
    const float *noise1;       /* [R,N_samples+N_importance], fine pass                    */
    const float *u;            /* [R,N_importance] sample_pdf draws; NULL => det  (utils.py:83-86) */
    const float *t_lin_imp;    /* [N_importance] torch.linspace(0,1,N_importance), used when u == NULL */
    const float *z_coarse;     /* optional [R,N_samples]: coarse depths already computed by nerf_amd_coarse_z (a caller
                                  that renders many chunks makes them for all rays in one launch); NULL = computed here
                                  from t_vals / t_rand */
    /* outputs (NULL = not wanted) */
    float *rgb_map, *disp_map, *acc_map;      /* [R,3] [R] [R]  from the last pass         */
    float *rgb0, *disp0, *acc0;               /* coarse-pass maps (N_importance > 0)       */
    float *z_std;                             /* [R]  std(z_samples, unbiased=False) (:168) */
    float *raw;                               /* [R,S_last,out_ch] raw of the last pass    */
    float *weights;                           /* [R,S_last]                                */
    float *z_vals;                            /* [R,S_last]                                */
    /* workspace: nerf_amd_render_rays_workspace(cfg, R, out_ch) bytes, 256-B aligned */
    void   *workspace;
    int64_t workspace_bytes;
} nerf_amd_render_io;

int64_t nerf_amd_render_rays_workspace(const nerf_amd_render_cfg *cfg, int64_t R, int32_t out_ch);
int     nerf_amd_render_rays(const nerf_amd_render_cfg *cfg, const nerf_amd_model *coarse,
                             const nerf_amd_model *fine /* NULL: reuse coarse (:150) */,
                             const nerf_amd_render_io *io, int64_t R, void *stream);

/* Renderer.render_batch (render_utils.py:51-65): the chunk loop around render_rays, as one call.
 * ios[i] / R[i] describe chunk i exactly as for nerf_amd_render_rays (own draws, own output rows);
 * consecutive chunks must have distinct workspaces (two buffers used alternately are enough).
 * Results are those of n calls of nerf_amd_render_rays, all on `stream`.  Between the two field kernels
 * of chunk k a single launch does the coarse compositing + resampling of chunk k AND the final compositing
 * of chunk k-1, so a chunk costs three dependent launches instead of four. */
int     nerf_amd_render_chunks(const nerf_amd_render_cfg *cfg, const nerf_amd_model *coarse,
                               const nerf_amd_model *fine, const nerf_amd_render_io *ios, const int64_t *R,
                               int32_t n_chunks, void *stream);

/* Renderer.render_batch (render_utils.py:51-65) over CONTIGUOUS whole-batch buffers: `io` describes all N rays exactly as
 * for nerf_amd_render_rays (rays [N,ray_ch], draws [N,.] -- made per chunk by the caller, in the reference's order, into
 * rows of one buffer each --, outputs [N,.]); z_coarse [N,N_samples] is normally given (one nerf_amd_coarse_z launch).
 * Results are bit-identical to per-chunk nerf_amd_render_rays calls (every kernel is per-ray independent), but the
 * launches are not tied to the caller's chunk size: rays are processed in groups of 32768, the two field kernels of
 * consecutive groups run back to back on `stream`, and the per-ray kernels (coarse compositing + resampling, final
 * compositing) run on a library-owned side stream BESIDE the next field kernel, ordered with events
 *      stream:  C0 C1 F0 C2 F1 C3 F2 ...          side:  M0  M1+Fin0  M2+Fin1 ...  Fin(n-1)
 * The call returns with `stream` ordered behind the last kernel of either stream.  Workspace:
 * nerf_amd_render_batch_workspace(cfg, N, out_ch) bytes (three groups' scratch in rotation), 256-B aligned. */
int64_t nerf_amd_render_batch_workspace(const nerf_amd_render_cfg *cfg, int64_t N, int32_t out_ch);
int     nerf_amd_render_batch(const nerf_amd_render_cfg *cfg, const nerf_amd_model *coarse,
                              const nerf_amd_model *fine, const nerf_amd_render_io *io, int64_t N, void *stream);

/* ------------------------------------------------------------------------
 * a12/a13 + ray-batch assembly of Renderer.render   utils.py:33-71, render_utils.py:200-226
 * Generates rays for flat pixel range [pix0, pix0+n) of an H x W image straight
 * into the [n, 8|11] batch layout (o, d, near, far, viewdirs), NDC-warped when
 * asked.  K = {fx, fy, cx, cy}; c2w is 12 floats (3x4 row-major), both HOST.
 * c2w_static (HOST, may be NULL) reproduces c2w_staticcam (:208-210).
 * ------------------------------------------------------------------------ */
int nerf_amd_make_rays(int32_t H, int32_t W, const double *K4, const float *c2w, const float *c2w_static,
                       int64_t pix0, int64_t n, float near, float far, int use_viewdirs, int ndc,
                       float *rays_out, void *stream);

/* Backward of get_rays with respect to the pose (utils.py:33-42; what the pose-estimation demo
 * differentiates, demo_est_rel_pose.py:87): gradients of rays_o / rays_d [n,3] (either may be NULL) for
 * flat pixels [pix0, pix0+n) -> g_c2w (12 floats, 3x4 row-major, DEVICE, overwritten). */
int nerf_amd_get_rays_backward(int32_t H, int32_t W, const double *K4, int64_t pix0, int64_t n, const float *g_rays_o,
                               const float *g_rays_d, float *g_c2w, void *stream);

/* Image output stage: utils.to8b (utils.py:30) as used by Renderer.render_from_batch_poses
 * (render_utils.py:312): out[i] = uint8(255 * clip(x[i], 0, 1)), truncating; NaN -> 0.
 * x [n] fp32 DEVICE (16-byte aligned), out [n] uint8 DEVICE (4-byte aligned). */
int nerf_amd_to8b(const float *x, int64_t n, uint8_t *out, void *stream);

/* utils.img2mse (utils.py:24; the loss of main.py:93-98): out[0] = mean((x - y)^2) over n fp32 DEVICE values.
 * One launch for n <= 16384 (a training batch); beyond that `partials` (256 floats, DEVICE) is required and a second
 * tiny launch sums the per-block partials in a fixed order.  _backward: gx = g[0] * 2 (x - y) / n, gy = -gx (either may
 * be NULL), g = the DEVICE scalar gradient of the loss. */
int nerf_amd_img2mse(const float *x, const float *y, int64_t n, float *out, float *partials, void *stream);
int nerf_amd_img2mse_backward(const float *x, const float *y, int64_t n, const float *g, float *gx, float *gy, void *stream);

/* Renderer.render(rays=...) batch assembly (render_utils.py:205-222) in one launch: out[i] = [rays_o[i] | rays_d[i] |
 * near | far | viewdir_src[i] / |viewdir_src[i]|]; viewdir_src NULL -> 8 columns, else 11.  All [n, 3] fp32 DEVICE,
 * contiguous.  (The c2w path has nerf_amd_make_rays.) */
int nerf_amd_assemble_rays(const float *rays_o, const float *rays_d, const float *viewdir_src, int64_t n, float near,
                           float far, float *out, void *stream);

/* Optimizer step of the training loop (main.py:104 on the torch.optim.Adam of utils.py:163-172): Adam for
 * n parameter tensors in one launch.  params / grads / exp_avg / exp_avg_sq: HOST arrays of n DEVICE pointers
 * to fp32 tensors of numel[i] elements (4-byte aligned; any n).  Per element, in fp32 and in the order of
 * torch/optim/adam.py _single_tensor_adam:
 *   g += weight_decay p (if != 0);  m += (1 - beta1)(g - m);  v = v beta2 + (1 - beta2) g g;
 *   p -= (lr / bias_correction1) * m / (sqrt(v) / sqrt(bias_correction2) + eps)
 * with bias_correction{1,2} = 1 - beta{1,2}^step evaluated in double from `step` (>= 1, the count INCLUDING
 * this step).  amsgrad / maximize are not provided. */
int nerf_amd_adam_step(int32_t n, float *const *params, const float *const *grads, float *const *exp_avg,
                       float *const *exp_avg_sq, const int64_t *numel, int64_t step, double lr, double beta1,
                       double beta2, double eps, double weight_decay, void *stream);

/* The same update for a CAPTURED training step (a HIP graph replays the arguments it was captured with, so nothing that
 * changes per step may be an argument): the step count *step_dev (int64, DEVICE; advanced by one on every execution) and
 * the learning rate *lr_dev (double, DEVICE; the caller rewrites it between replays for main.py:108-112's decay) live in
 * device memory; scalars_dev is 2 floats of DEVICE scratch.  n <= 64. */
int nerf_amd_adam_step_device(int32_t n, float *const *params, const float *const *grads, float *const *exp_avg,
                              float *const *exp_avg_sq, const int64_t *numel, int64_t *step_dev, const double *lr_dev,
                              double beta1, double beta2, double eps, double weight_decay, float *scalars_dev, void *stream);

/* ------------------------------------------------------------------------
 * Measurement hook (bench.py): while enabled, every field-MLP launch is
 * bracketed by hipEvents on its own stream.  nerf_amd_profile_collect waits for
 * the recorded events and returns, per class (0 = fp32 kernel, 1 = fused bf16
 * kernel, 2 = split-precision kernel), the number of launches, the summed device time in milliseconds and
 * the summed number of points, then forgets them.
 * ------------------------------------------------------------------------ */
int nerf_amd_profile_enable(int on);
/* Tuning knobs for A/B measurements (results are identical for every setting).
 * key 0: weight-pipeline shape of the fused bf16 kernel (0 = default; see mlp_bf16.hip launch_one). */
int nerf_amd_set_tuning(int key, int value);
int nerf_amd_profile_collect(int64_t launches[3], double total_ms[3], double total_points[3]);

#ifdef __cplusplus
}
#endif
#endif /* NERF_AMD_H */
