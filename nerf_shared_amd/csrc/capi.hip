// capi.hip -- the extern "C" boundary declared in include/nerf_amd.h.
#include <hip/hip_runtime.h>

#include <cmath>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <utility>
#include <new>
#include <string>
#include <vector>

#include "kernels.h"
#include "program.h"

using namespace na;

namespace {
thread_local std::string g_err;

std::string lds_msg(const char *what, size_t bytes) {
    return std::string(what) + ": the per-ray scratch needs " + std::to_string(bytes) + " bytes of LDS per workgroup, the CU has " +
           std::to_string(LDS_LIMIT_BYTES) + " (fewer samples per ray)";
}

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char *what) {
    return fail(NERF_AMD_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);    \
    } while (0)

// ---- measurement hook: hipEvent pairs around field launches
struct ProfRec { hipEvent_t a, b; int cls; double points; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_free;

template <class T>
int upload(T **dst, const std::vector<T> &src) {
    *dst = nullptr;
    if (src.empty()) return NERF_AMD_OK;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), src.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return NERF_AMD_OK;
}
size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

struct nerf_amd_model {
    Program prog;
    int device = 0;
    FragDesc *d_frags = nullptr;
    TileDesc *d_tiles = nullptr;
    LayerF32 *d_layers = nullptr;
    TensorDesc *d_tensors = nullptr;
    uint16_t *stream_bf16 = nullptr, *stream_s16 = nullptr, *stream_bwd = nullptr, *stream_split = nullptr;
    FragDesc *d_frags_bwd = nullptr, *d_frags_split = nullptr, *d_frags_bwd_split = nullptr;
    uint16_t *stream_bwd_split = nullptr;
    float *bias_bf16 = nullptr, *stream_f32 = nullptr, *bias_f32 = nullptr, *bias_s16 = nullptr;
    FragDesc *d_frags16 = nullptr;
    TileDesc *d_tiles16 = nullptr;
    TrainLayerF32 *d_tlayers = nullptr;  // train_f32.hip: per-layer descriptors and the transposed fp32 stream
    float *stream_f32_t = nullptr;
    int fresh = 0;                       // NERF_AMD_COPY_* of the packed copies that hold the current parameters
};

extern "C" {

int nerf_amd_abi_version(void) { return NERF_AMD_ABI_VERSION; }
const char *nerf_amd_last_error(void) { return g_err.c_str(); }

int nerf_amd_model_create(const nerf_amd_arch *arch, int device, nerf_amd_model **out) {
    if (!arch || !out) return fail(NERF_AMD_EINVAL, "null argument");
    *out = nullptr;
    nerf_amd_model *m = new (std::nothrow) nerf_amd_model;
    if (!m) return fail(NERF_AMD_ENOMEM, "out of host memory");
    const char *err = "";
    if (build_program(*arch, m->prog, &err) != 0) {
        delete m;
        return fail(NERF_AMD_EINVAL, err);
    }
    m->device = device;
    if (hipError_t e0 = hipSetDevice(device); e0 != hipSuccess) {
        delete m;
        return hip_fail(e0, "hipSetDevice");
    }
    const Program &p = m->prog;
    int rc;
    if ((rc = upload(&m->d_frags, p.frags)) || (rc = upload(&m->d_tiles, p.tiles)) ||
        (rc = upload(&m->d_frags16, p.frags16)) || (rc = upload(&m->d_tiles16, p.tiles16)) ||
        (rc = upload(&m->d_frags_bwd, p.frags_bwd)) || (rc = upload(&m->d_frags_split, p.frags_split)) ||
        (rc = upload(&m->d_frags_bwd_split, p.frags_bwd_split)) ||
        (rc = upload(&m->d_layers, p.layers)) || (rc = upload(&m->d_tensors, p.tensors)) || (rc = upload(&m->d_tlayers, p.tlayers))) {
        nerf_amd_model_destroy(m);
        return rc;
    }
    hipError_t e = hipSuccess;
    if (p.bf16_ok) {
        e = hipMalloc(reinterpret_cast<void **>(&m->stream_bf16), p.frags.size() * 1024);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->bias_bf16), p.tiles.size() * 32 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->stream_s16), p.frags16.size() * 1024);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->bias_s16), p.tiles16.size() * 16 * sizeof(float));
        if (e == hipSuccess && !p.frags_bwd.empty())
            e = hipMalloc(reinterpret_cast<void **>(&m->stream_bwd), p.frags_bwd.size() * 1024);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->stream_split), p.frags_split.size() * 1024);
        if (e == hipSuccess && !p.frags_bwd_split.empty())
            e = hipMalloc(reinterpret_cast<void **>(&m->stream_bwd_split), p.frags_bwd_split.size() * 1024);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->stream_f32), (size_t)p.f32_stream_floats * sizeof(float));
    if (e == hipSuccess && tile_counters_init(device) != NERF_AMD_OK) e = hipErrorOutOfMemory;      // the field kernel's ticket counters
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->bias_f32), (size_t)p.f32_bias_floats * sizeof(float));
    if (e == hipSuccess && p.f32_stream_t_floats > 0)
        e = hipMalloc(reinterpret_cast<void **>(&m->stream_f32_t), (size_t)p.f32_stream_t_floats * sizeof(float));
    if (e != hipSuccess) {
        nerf_amd_model_destroy(m);
        return hip_fail(e, "hipMalloc(model buffers)");
    }
    *out = m;
    return NERF_AMD_OK;
}

int nerf_amd_model_update_copies(nerf_amd_model *m, const float *const *weights, const float *const *biases,
                                 int n_tensors, int copies, int others_current, void *stream) {
    if (!m || !weights || !biases) return fail(NERF_AMD_EINVAL, "null argument");
    if (copies & ~NERF_AMD_COPY_ALL) return fail(NERF_AMD_EINVAL, "unknown copy bits");
    const Program &p = m->prog;
    if (n_tensors != (int)p.tensors.size())
        return fail(NERF_AMD_EINVAL, "expected " + std::to_string(p.tensors.size()) + " parameter tensors, got " + std::to_string(n_tensors));
    for (int i = 0; i < n_tensors; ++i)
        if (!weights[i] || !biases[i]) return fail(NERF_AMD_EINVAL, "null parameter pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_tensors > MAX_TENSORS) return fail(NERF_AMD_EINVAL, "too many parameter tensors");
    PtrTable wt, bt;
    std::memset(&wt, 0, sizeof(wt));
    std::memset(&bt, 0, sizeof(bt));
    for (int i = 0; i < n_tensors; ++i) { wt.p[i] = weights[i]; bt.p[i] = biases[i]; }
    int rc = launch_pack(p, m->d_frags, m->d_tiles, m->d_layers, m->d_tensors, wt, bt,
                         m->stream_bf16, m->bias_bf16, m->stream_f32, m->bias_f32,
                         m->d_frags16, m->d_tiles16, m->stream_s16, m->bias_s16, m->d_frags_bwd, m->stream_bwd,
                         m->d_frags_split, m->stream_split, m->d_frags_bwd_split, m->stream_bwd_split, m->d_tlayers, m->stream_f32_t,
                         copies, s);
    if (rc) return fail(rc, "pack launch failed");
    m->fresh = (others_current ? m->fresh : 0) | copies;
    return NERF_AMD_OK;
}

int nerf_amd_model_update(nerf_amd_model *m, const float *const *weights, const float *const *biases,
                          int n_tensors, void *stream) {
    return nerf_amd_model_update_copies(m, weights, biases, n_tensors, NERF_AMD_COPY_ALL, 0, stream);
}

void nerf_amd_model_destroy(nerf_amd_model *m) {
    if (!m) return;
    (void)hipFree(m->d_frags); (void)hipFree(m->d_tiles); (void)hipFree(m->d_layers); (void)hipFree(m->d_tensors);
    (void)hipFree(m->d_frags_bwd_split); (void)hipFree(m->stream_bwd_split);
    (void)hipFree(m->d_frags_bwd); (void)hipFree(m->stream_bwd); (void)hipFree(m->d_frags_split); (void)hipFree(m->stream_split);
    (void)hipFree(m->d_frags16); (void)hipFree(m->d_tiles16); (void)hipFree(m->stream_s16); (void)hipFree(m->bias_s16);
    (void)hipFree(m->stream_bf16); (void)hipFree(m->bias_bf16); (void)hipFree(m->stream_f32); (void)hipFree(m->bias_f32);
    (void)hipFree(m->d_tlayers); (void)hipFree(m->stream_f32_t);
    delete m;
}

int nerf_amd_model_supports_bf16(const nerf_amd_model *m) {
    if (!m) return 0;
    const nerf_amd_arch &a = m->prog.arch;
    return m->prog.bf16_ok && a.i_embed == 0 && mlp_bf16_supported(a.multires, a.multires_views, a.use_viewdirs);
}
int nerf_amd_model_supports_split(const nerf_amd_model *m) {
    if (!m) return 0;
    const nerf_amd_arch &a = m->prog.arch;
    return m->prog.bf16_ok && a.i_embed == 0 && mlp_split_supported(a.multires, a.multires_views, a.use_viewdirs, m->prog.out_ch);
}
int nerf_amd_model_out_ch(const nerf_amd_model *m) { return m ? m->prog.out_ch : 0; }

int nerf_amd_pack_bf16_host(const nerf_amd_arch *arch, int shape, const float *const *weights, const float *const *biases,
                            int n_tensors, uint16_t *stream_out, int64_t *n_frags, float *bias_out, int64_t *n_bias) {
    if (shape != 16 && shape != 32 && shape != 17 && shape != 18 && shape != 19)
        return fail(NERF_AMD_EINVAL, "shape must be 32 (32x32x16 stream), 16 (16x16x32 stream), 17 (backward stream), 18 (split-precision stream) or 19 (split-precision backward stream)");
    if (!arch) return fail(NERF_AMD_EINVAL, "null argument");
    Program p;
    const char *err = "";
    if (build_program(*arch, p, &err) != 0) return fail(NERF_AMD_EINVAL, err);
    if (!p.bf16_ok) return fail(NERF_AMD_EUNSUPPORTED, "architecture has no fused bf16 program (needs D=8, W=256, skips=[4])");
    if (n_frags) *n_frags = (int64_t)(shape == 19 ? p.frags_bwd_split.size() : shape == 18 ? p.frags_split.size() : shape == 17 ? p.frags_bwd.size() : shape == 16 ? p.frags16.size() : p.frags.size());
    if (n_bias) *n_bias = (shape == 17 || shape == 18 || shape == 19) ? 0 : shape == 16 ? (int64_t)p.tiles16.size() * 16 : (int64_t)p.tiles.size() * 32;
    if (stream_out || bias_out) {
        if (!weights || !biases || n_tensors != (int)p.tensors.size()) return fail(NERF_AMD_EINVAL, "bad parameter list");
        pack_bf16_host(p, shape, weights, biases, stream_out, bias_out);
    }
    return NERF_AMD_OK;
}

int nerf_amd_embed(const float *x, int64_t n, int multires, float *out, void *stream) {
    if (n < 0 || multires < 0 || multires > 20 || (n > 0 && (!x || !out))) return fail(NERF_AMD_EINVAL, "bad embed arguments");
    int rc = launch_embed(x, n, multires, out, static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "embed launch failed") : NERF_AMD_OK;
}

}  // extern "C"

namespace {

// The packed copy an entry point is about to read must hold the current parameters.
int need_copy(const nerf_amd_model *m, int copy) {
    if (m->fresh & copy) return NERF_AMD_OK;
    return fail(NERF_AMD_EINVAL, m->fresh ? "the packed copy of the parameters this call needs is stale or was never made (nerf_amd_model_update_copies)"
                                          : "model has no parameters yet (call nerf_amd_model_update)");
}

int run_field(const nerf_amd_model *m, MlpArgs a, int precision, hipStream_t s) {
    const Program &p = m->prog;
    if (int rc0 = need_copy(m, precision == NERF_AMD_PREC_BF16 ? NERF_AMD_COPY_BF16 : precision == NERF_AMD_PREC_FP32_SPLIT ? NERF_AMD_COPY_SPLIT : NERF_AMD_COPY_FP32))
        return rc0;
    a.stream_bf16 = m->stream_bf16; a.bias_bf16 = m->bias_bf16;
    a.stream_s16 = m->stream_s16; a.bias_s16 = m->bias_s16;
    a.stream_split = m->stream_split;
    a.stream_f32 = m->stream_f32; a.bias_f32 = m->bias_f32;
    a.layers = m->d_layers; a.n_layers = (int)p.layers.size();
    a.input_ch = p.input_ch; a.input_ch_views = p.input_ch_views; a.W = p.arch.W; a.lds_rows = p.lds_rows;
    a.multires = p.arch.multires; a.multires_views = p.arch.multires_views; a.i_embed = p.arch.i_embed;
    a.out_ch = p.out_ch;
    if (p.arch.use_viewdirs && !a.viewdirs) return fail(NERF_AMD_EINVAL, "model has a view branch but no viewdirs were given");
    if (!p.arch.use_viewdirs) a.viewdirs = nullptr;
    int rc;
    ProfRec rec{nullptr, nullptr, precision == NERF_AMD_PREC_BF16 ? 1 : precision == NERF_AMD_PREC_FP32_SPLIT ? 2 : 0, (double)a.P};
    bool prof = false;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof_on) {
            prof = true;
            if (!g_prof_free.empty()) {
                rec.a = g_prof_free.back().first; rec.b = g_prof_free.back().second;
                g_prof_free.pop_back();
            } else if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) {
                prof = false;
            }
        }
    }
    if (prof) (void)hipEventRecord(rec.a, s);
    struct Closer {
        bool on; ProfRec r; hipStream_t s;
        ~Closer() {
            if (!on) return;
            (void)hipEventRecord(r.b, s);
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_prof_recs.push_back(r);
        }
    } closer{prof, rec, s};
    if (precision == NERF_AMD_PREC_BF16) {
        if (!nerf_amd_model_supports_bf16(m))
            return fail(NERF_AMD_EUNSUPPORTED, "fused bf16 kernel needs D=8, W=256, skips=[4], multires/views in {(10,4),(15,6)}; use NERF_AMD_PREC_FP32");
        if (g_variant >= 100)   // A/B: the first-generation 32x32x16 kernel
            rc = launch_mlp_bf16(a, p.arch.multires, p.arch.multires_views, p.arch.use_viewdirs, p.n_frags_used, (int)p.tiles.size(), s);
        else {
            rc = launch_mlp_bf16_s16(a, p.arch.multires, p.arch.multires_views, p.arch.use_viewdirs, p.n_frags16_used, (int)p.tiles16.size(), s);
            if (rc == NERF_AMD_EUNSUPPORTED)   // e.g. output_ch > 16: the 32x32x16 kernel covers it
                rc = launch_mlp_bf16(a, p.arch.multires, p.arch.multires_views, p.arch.use_viewdirs, p.n_frags_used, (int)p.tiles.size(), s);
        }
    } else if (precision == NERF_AMD_PREC_FP32) {
        rc = launch_mlp_f32(a, s);
    } else if (precision == NERF_AMD_PREC_FP32_SPLIT) {
        if (!nerf_amd_model_supports_split(m))
            return fail(NERF_AMD_EUNSUPPORTED, "split-precision kernel needs D=8, W=256, skips=[4], multires/views in {(10,4),(15,6)} (or 10 / 15 without a view branch, output_ch <= 16); use NERF_AMD_PREC_FP32");
        rc = launch_mlp_split(a, p.arch.multires, p.arch.multires_views, p.arch.use_viewdirs, p.n_frags_split_used, (int)p.tiles16.size(), s);
    } else {
        return fail(NERF_AMD_EINVAL, "unknown precision");
    }
    return rc ? fail(rc, "field kernel launch failed") : NERF_AMD_OK;
}

}  // namespace

extern "C" {

int nerf_amd_nerf_forward(const nerf_amd_model *m, const float *pts, const float *viewdirs,
                          int64_t n_rays, int32_t n_samples, float *out, int precision, void *stream) {
    if (!m || n_rays < 0 || n_samples < 1) return fail(NERF_AMD_EINVAL, "bad forward arguments");
    if (n_rays == 0) return NERF_AMD_OK;
    if (!pts || !out) return fail(NERF_AMD_EINVAL, "null pts/out");
    MlpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.pts = pts; a.viewdirs = viewdirs; a.vd_stride = 3;
    a.P = n_rays * n_samples; a.S = n_samples; a.out = out;
    return run_field(m, a, precision, static_cast<hipStream_t>(stream));
}

int nerf_amd_mlp_embedded(const nerf_amd_model *m, const float *x, int64_t n, float *out, void *stream) {
    if (!m || n < 0) return fail(NERF_AMD_EINVAL, "bad MLP arguments");
    if (n == 0) return NERF_AMD_OK;
    if (!x || !out) return fail(NERF_AMD_EINVAL, "null x/out");
    MlpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.embedded = x;
    a.viewdirs = x;            // non-null marker: the view columns are inside x
    a.vd_stride = 0;
    a.P = n; a.S = 1; a.out = out;
    return run_field(m, a, NERF_AMD_PREC_FP32, static_cast<hipStream_t>(stream));
}

int nerf_amd_ndc_rays(int32_t H, int32_t W, double focal, float near, const float *rays_o, const float *rays_d,
                      int64_t n, float *out_o, float *out_d, void *stream) {
    if (H < 1 || W < 1 || n < 0 || (n > 0 && (!rays_o || !rays_d || !out_o || !out_d)))
        return fail(NERF_AMD_EINVAL, "bad ndc_rays arguments");
    int rc = launch_ndc_rays(H, W, focal, near, rays_o, rays_d, n, out_o, out_d, static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "ndc_rays launch failed") : NERF_AMD_OK;
}

int nerf_amd_ndc_rays_backward(int32_t H, int32_t W, double focal, float near, const float *rays_o, const float *rays_d,
                               const float *g_out_o, const float *g_out_d, int64_t n, float *g_rays_o, float *g_rays_d,
                               void *stream) {
    if (H < 1 || W < 1 || n < 0 || (n > 0 && (!rays_o || !rays_d)))
        return fail(NERF_AMD_EINVAL, "bad ndc_rays_backward arguments");
    int rc = launch_ndc_rays_bwd(H, W, focal, near, rays_o, rays_d, g_out_o, g_out_d, n, g_rays_o, g_rays_d,
                                 static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "ndc_rays backward launch failed") : NERF_AMD_OK;
}

namespace {
const char *TRAIN_COVER = "fused training kernels cover D=8, W=256, skips=[4] with view branch (multires 10/4 or 15/6) or without (multires 10 or 15, output_ch <= 16), in NERF_AMD_PREC_BF16 or NERF_AMD_PREC_FP32_SPLIT; NERF_AMD_PREC_FP32 trains any architecture";
// which training path a (model, precision) pair takes: 1 fused bf16 / split kernels, 2 the exact-fp32 path, 0 none
int train_path(const nerf_amd_model *m, int precision) {
    if ((precision == NERF_AMD_PREC_BF16 || precision == NERF_AMD_PREC_FP32_SPLIT) && train_supported(m->prog)) return 1;
    if (precision == NERF_AMD_PREC_FP32 && train_f32_supported(m->prog)) return 2;
    return 0;
}
}  // namespace

int nerf_amd_model_supports_training(const nerf_amd_model *m, int precision) {
    return m && train_path(m, precision) ? 1 : 0;
}

int64_t nerf_amd_train_workspace(const nerf_amd_model *m, int64_t n_points, int precision) {
    if (!m || n_points < 0) return -1;
    const int path = train_path(m, precision);
    if (path == 2) return train_f32_workspace_bytes(m->prog, n_points);
    if (path != 1) return -1;
    return train_workspace_bytes(m->prog, n_points, precision == NERF_AMD_PREC_FP32_SPLIT);
}

int nerf_amd_field_forward_train(const nerf_amd_model *m, const float *pts, const float *viewdirs, const float *rays,
                                 int32_t ray_ch, const float *z_vals, int64_t R, int32_t S, float *raw, void *workspace,
                                 int64_t workspace_bytes, int precision, void *stream) {
    if (!m || R < 0 || S < 1) return fail(NERF_AMD_EINVAL, "bad forward_train arguments");
    const bool vd = m->prog.arch.use_viewdirs != 0;
    if (!pts && ray_ch != (vd ? 11 : 8)) return fail(NERF_AMD_EINVAL, vd ? "rays must be [R,11]" : "rays must be [R,8] for a model without view branch");
    if (pts && vd && !viewdirs) return fail(NERF_AMD_EINVAL, "pts mode needs viewdirs [R,3]");
    const int path = train_path(m, precision);
    if (!path) return fail(NERF_AMD_EUNSUPPORTED, TRAIN_COVER);
    if (path == 2) {                                   // exact fp32, any architecture (train_f32.hip)
        if (int rc0 = need_copy(m, NERF_AMD_COPY_FP32)) return rc0;
        if (R == 0) return NERF_AMD_OK;
        const int64_t P = R * S;
        if ((!pts && (!rays || !z_vals)) || !raw || !workspace || workspace_bytes < train_f32_workspace_bytes(m->prog, P))
            return fail(NERF_AMD_EINVAL, "null pointer or workspace too small");
        const Program &p = m->prog;
        MlpArgs a;
        std::memset(&a, 0, sizeof(a));
        a.stream_f32 = m->stream_f32; a.bias_f32 = m->bias_f32; a.layers = m->d_layers; a.n_layers = (int)p.layers.size();
        a.input_ch = p.input_ch; a.input_ch_views = p.input_ch_views; a.W = p.arch.W; a.lds_rows = p.lds_rows;
        a.multires = p.arch.multires; a.multires_views = p.arch.multires_views; a.i_embed = p.arch.i_embed;
        if (pts) { a.pts = pts; a.viewdirs = vd ? viewdirs : nullptr; a.vd_stride = 3; }
        else { a.rays = rays; a.ray_stride = ray_ch; a.z_vals = z_vals; a.viewdirs = vd ? rays + 8 : nullptr; a.vd_stride = ray_ch; }
        a.P = P; a.S = S; a.out = raw; a.out_ch = p.out_ch;
        int rc = launch_train_f32_forward(p, a, m->d_tlayers, workspace, static_cast<hipStream_t>(stream));
        return rc ? fail(rc, "exact-fp32 training forward launch failed") : NERF_AMD_OK;
    }
    if (int rc0 = need_copy(m, precision == NERF_AMD_PREC_FP32_SPLIT ? NERF_AMD_COPY_SPLIT : NERF_AMD_COPY_BF16)) return rc0;
    if (R == 0) return NERF_AMD_OK;
    const bool split = precision == NERF_AMD_PREC_FP32_SPLIT;
    const int64_t P = R * S;
    if ((!pts && (!rays || !z_vals)) || !raw || !workspace || workspace_bytes < train_workspace_bytes(m->prog, P, split))
        return fail(NERF_AMD_EINVAL, "null pointer or workspace too small");
    MlpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.stream_s16 = m->stream_s16; a.bias_s16 = m->bias_s16; a.stream_split = m->stream_split;
    if (pts) { a.pts = pts; a.viewdirs = vd ? viewdirs : nullptr; a.vd_stride = 3; }
    else { a.rays = rays; a.ray_stride = ray_ch; a.z_vals = z_vals; a.viewdirs = vd ? rays + 8 : nullptr; a.vd_stride = ray_ch; }
    a.P = P; a.S = S; a.out = raw; a.out_ch = m->prog.out_ch;
    train_fill_args(m->prog, P, workspace, &a, split);
    const nerf_amd_arch &ar = m->prog.arch;
    int rc = split ? launch_mlp_split_save(a, ar.multires, ar.multires_views, vd, m->prog.n_frags_split_used, (int)m->prog.tiles16.size(), static_cast<hipStream_t>(stream))
                   : launch_mlp_bf16_s16_save(a, ar.multires, ar.multires_views, vd, m->prog.n_frags16_used, (int)m->prog.tiles16.size(), static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "training forward launch failed") : NERF_AMD_OK;
}

int nerf_amd_field_backward(const nerf_amd_model *m, const float *g_raw, const float *pts, const float *viewdirs,
                            const float *rays, int32_t ray_ch, const float *z_vals, int64_t R, int32_t S,
                            void *workspace, int64_t workspace_bytes, float *const *grad_weights,
                            float *const *grad_biases, int n_tensors, float *g_pts, float *g_rays, float *g_viewdirs,
                            int precision, void *stream) {
    const int64_t n_points = R * S;
    if (!m || R < 0 || S < 1 || !grad_weights || !grad_biases) return fail(NERF_AMD_EINVAL, "bad backward arguments");
    const bool vd = m->prog.arch.use_viewdirs != 0;
    if ((!pts && (ray_ch != (vd ? 11 : 8) || !rays || !z_vals)) || (pts && vd && !viewdirs))
        return fail(NERF_AMD_EINVAL, "backward needs the forward's inputs (pts + viewdirs, or rays [R,11] + z_vals; [R,8] without view branch)");
    const int path = train_path(m, precision);
    if (!path) return fail(NERF_AMD_EUNSUPPORTED, TRAIN_COVER);
    if (n_tensors != (int)m->prog.tensors.size()) return fail(NERF_AMD_EINVAL, "wrong number of gradient tensors");
    if (n_points == 0) return NERF_AMD_OK;
    if (path == 2) {                                   // exact fp32, any architecture (train_f32.hip)
        if (int rc0 = need_copy(m, NERF_AMD_COPY_FP32_BWD)) return rc0;
        if (!g_raw || !workspace || workspace_bytes < train_f32_workspace_bytes(m->prog, n_points))
            return fail(NERF_AMD_EINVAL, "null pointer or workspace too small");
        for (int i = 0; i < n_tensors; ++i)
            if (!grad_weights[i] || !grad_biases[i]) return fail(NERF_AMD_EINVAL, "null gradient pointer");
        const Program &p = m->prog;
        MlpArgs a;
        std::memset(&a, 0, sizeof(a));
        a.layers = m->d_layers; a.n_layers = (int)p.layers.size(); a.lds_rows = p.lds_rows; a.out_ch = p.out_ch;
        a.input_ch = p.input_ch; a.input_ch_views = p.input_ch_views; a.W = p.arch.W;
        a.multires = p.arch.multires; a.multires_views = p.arch.multires_views; a.i_embed = p.arch.i_embed;
        if (pts) { a.pts = pts; a.viewdirs = vd ? viewdirs : nullptr; a.vd_stride = 3; }
        else { a.rays = rays; a.ray_stride = ray_ch; a.z_vals = z_vals; a.viewdirs = vd ? rays + 8 : nullptr; a.vd_stride = ray_ch; }
        a.g_pts = g_pts; a.g_rays = g_rays; a.g_vd = vd ? g_viewdirs : nullptr;
        a.P = n_points; a.S = S; a.g_raw = g_raw;
        int rc = launch_train_f32_backward(p, a, m->d_tlayers, m->stream_f32_t, workspace, grad_weights, grad_biases, static_cast<hipStream_t>(stream));
        return rc ? fail(rc, "exact-fp32 backward launch failed") : NERF_AMD_OK;
    }
    const bool split = precision == NERF_AMD_PREC_FP32_SPLIT;
    if (int rc0 = need_copy(m, split ? NERF_AMD_COPY_BWD_SPLIT : NERF_AMD_COPY_BWD)) return rc0;
    if (!g_raw || !workspace || workspace_bytes < train_workspace_bytes(m->prog, n_points, split))
        return fail(NERF_AMD_EINVAL, "null pointer or workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MlpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.stream_bwd = m->stream_bwd; a.stream_bwd_split = m->stream_bwd_split; a.g_raw = g_raw; a.P = n_points; a.S = S; a.out_ch = m->prog.out_ch;
    if (pts) { a.pts = pts; a.viewdirs = vd ? viewdirs : nullptr; a.vd_stride = 3; }
    else { a.rays = rays; a.ray_stride = ray_ch; a.z_vals = z_vals; a.viewdirs = vd ? rays + 8 : nullptr; a.vd_stride = ray_ch; }
    a.g_pts = g_pts; a.g_rays = g_rays; a.g_vd = vd ? g_viewdirs : nullptr;
    train_fill_args(m->prog, n_points, workspace, &a, split);
    const nerf_amd_arch &ar = m->prog.arch;
    int rc = split ? launch_mlp_bwd_split(a, ar.multires, ar.multires_views, vd, m->prog.n_frags_bwd_split_used, s)
                   : launch_mlp_bwd_s16(a, ar.multires, ar.multires_views, vd, m->prog.n_frags_bwd_used, s);
    if (rc) return fail(rc, "backward kernel launch failed");
    rc = train_param_grads(m->prog, n_points, workspace, grad_weights, grad_biases, m->device, s, split, g_raw);
    return rc ? fail(rc, "weight-gradient GEMMs failed") : NERF_AMD_OK;
}

int nerf_amd_raw2outputs(const float *raw, int32_t raw_ch, const float *z_vals, const float *rays_d,
                         int32_t rays_d_stride, const float *noise, int64_t R, int32_t S, int white_bkgd,
                         float *rgb_map, float *disp_map, float *acc_map, float *weights, float *depth_map,
                         void *stream) {
    if (R < 0 || S < 1 || raw_ch < 4 || (R > 0 && (!raw || !z_vals || !rays_d))) return fail(NERF_AMD_EINVAL, "bad raw2outputs arguments");
    int rc = launch_composite(raw, raw_ch, z_vals, rays_d, rays_d_stride, noise, R, S, white_bkgd, rgb_map, disp_map,
                              acc_map, weights, depth_map, static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "composite launch failed") : NERF_AMD_OK;
}

int nerf_amd_raw2outputs_backward(const float *raw, int32_t raw_ch, const float *z_vals, const float *rays_d,
                                  int32_t rays_d_stride, const float *noise, int64_t R, int32_t S, int white_bkgd,
                                  const float *g_rgb_map, const float *g_disp_map, const float *g_acc_map,
                                  const float *g_depth_map, const float *g_weights, float *g_raw, float *g_rays_d,
                                  void *stream) {
    if (R < 0 || S < 1 || raw_ch < 4 || (R > 0 && (!raw || !z_vals || !rays_d || !g_raw)))
        return fail(NERF_AMD_EINVAL, "bad raw2outputs_backward arguments");
    int rc = launch_composite_bwd(raw, raw_ch, z_vals, rays_d, rays_d_stride, noise, R, S, white_bkgd, g_rgb_map,
                                  g_disp_map, g_acc_map, g_depth_map, g_weights, g_raw, g_rays_d,
                                  static_cast<hipStream_t>(stream));
    if (rc == NERF_AMD_EINVAL && composite_bwd_lds_bytes(S) > LDS_LIMIT_BYTES) return fail(rc, lds_msg("raw2outputs_backward", composite_bwd_lds_bytes(S)));
    return rc ? fail(rc, "composite backward launch failed") : NERF_AMD_OK;
}

int nerf_amd_sample_pdf(const float *bins, const float *weights, const float *u, const float *t_lin,
                        int64_t R, int32_t n_bins, int32_t n_samples, float *samples, void *stream) {
    if (R < 0 || n_bins < 2 || n_samples < 0 || (R > 0 && (!bins || !weights || !samples || (!u && !t_lin))))
        return fail(NERF_AMD_EINVAL, "bad sample_pdf arguments");
    int rc = launch_sample_pdf(bins, weights, u, t_lin, R, n_bins, n_samples, samples, static_cast<hipStream_t>(stream));
    if (rc == NERF_AMD_EINVAL && sample_pdf_lds_bytes(n_bins) > LDS_LIMIT_BYTES) return fail(rc, lds_msg("sample_pdf", sample_pdf_lds_bytes(n_bins)));
    return rc ? fail(rc, "sample_pdf launch failed") : NERF_AMD_OK;
}

int nerf_amd_coarse_z(const float *rays, int32_t ray_ch, const float *t_vals, const float *t_rand, int64_t R,
                      int32_t N_samples, int lindisp, int perturb, float *z_vals, void *stream) {
    if (R < 0 || N_samples < 1 || ray_ch < 8 || (R > 0 && (!rays || !t_vals || !z_vals || (perturb && !t_rand))))
        return fail(NERF_AMD_EINVAL, "bad coarse_z arguments");
    int rc = launch_coarse_z(rays, ray_ch, t_vals, perturb ? t_rand : nullptr, R, N_samples, lindisp, perturb, z_vals,
                             static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "coarse_z launch failed") : NERF_AMD_OK;
}

int nerf_amd_resample(const float *z_coarse, const float *weights, const float *u, const float *t_lin, int64_t R,
                      int32_t N_samples, int32_t N_importance, float *z_fine, float *z_std, void *stream) {
    if (R < 0 || N_samples < 3 || N_importance < 1 || (R > 0 && (!z_coarse || !weights || !z_fine || (!u && !t_lin))))
        return fail(NERF_AMD_EINVAL, "bad resample arguments");
    int rc = launch_resample(z_coarse, weights, u, t_lin, R, N_samples, N_importance, z_fine, z_std,
                             static_cast<hipStream_t>(stream));
    if (rc == NERF_AMD_EINVAL && resample_lds_bytes(N_samples, N_importance, false) > LDS_LIMIT_BYTES)
        return fail(rc, lds_msg("resample", resample_lds_bytes(N_samples, N_importance, false)));
    return rc ? fail(rc, "resample launch failed") : NERF_AMD_OK;
}

int64_t nerf_amd_render_rays_workspace(const nerf_amd_render_cfg *cfg, int64_t R, int32_t out_ch) {
    if (!cfg || R < 0) return -1;
    const size_t Nc = cfg->N_samples, Nf = cfg->N_samples + cfg->N_importance;
    size_t b = 0;
    b += align_up(R * Nc * sizeof(float));                 // z coarse
    b += align_up(R * Nc * out_ch * sizeof(float));        // raw coarse
    b += align_up(R * Nc * sizeof(float));                 // weights coarse
    if (cfg->N_importance > 0) {
        b += align_up(R * Nf * sizeof(float));             // z fine
        b += align_up(R * Nf * out_ch * sizeof(float));    // raw fine
    }
    return (int64_t)b;
}

}  // extern "C"

// ---- render_rays over one chunk's buffers, in stages: z_vals, coarse field, coarse compositing (+ resampling),
// fine field, final compositing.  nerf_amd_render_chunks interleaves the stages of consecutive chunks.
namespace {
struct ChunkPlan {
    const nerf_amd_render_cfg *cfg;
    const nerf_amd_model *coarse, *fm;
    const nerf_amd_render_io *io;
    int64_t R;
    int Nc, Ni, Nf, och;
    bool has_vd;
    float *z_c, *raw_c, *w_c, *z_f, *raw_f;
};

int plan_chunk(const nerf_amd_render_cfg *cfg, const nerf_amd_model *coarse, const nerf_amd_model *fine,
               const nerf_amd_render_io *io, int64_t R, ChunkPlan *p) {
    if (!cfg || !coarse || !io || R < 0) return fail(NERF_AMD_EINVAL, "null argument");
    const int Nc = cfg->N_samples, Ni = cfg->N_importance, Nf = Nc + Ni;
    if (Nc < 1 || Ni < 0) return fail(NERF_AMD_EINVAL, "bad sample counts");
    // fewer than three coarse samples leave sample_pdf without an interior weight: the reference's cdf is then EMPTY
    // (zeros_like(cdf[..., :1]) of an empty cumsum, utils.py:78-79) and its gather raises an index error
    if (Ni > 0 && Nc < 3) return fail(NERF_AMD_EINVAL, "hierarchical sampling needs N_samples >= 3 (with fewer the reference's sample_pdf fails too: empty cdf)");
    if (Ni > 0 && resample_lds_bytes(Nc, Ni, true) > LDS_LIMIT_BYTES)      // refuse before anything is launched
        return fail(NERF_AMD_EINVAL, lds_msg("render_rays (compositing + resampling)", resample_lds_bytes(Nc, Ni, true)));
    if (!io->rays || (io->ray_ch != 8 && io->ray_ch != 11)) return fail(NERF_AMD_EINVAL, "rays must be [R,8] or [R,11]");
    if (!io->t_vals) return fail(NERF_AMD_EINVAL, "t_vals missing");
    if (cfg->perturb && !io->t_rand && !io->z_coarse) return fail(NERF_AMD_EINVAL, "perturb set but t_rand missing");
    if (cfg->use_noise && (!io->noise0 || (Ni > 0 && !io->noise1))) return fail(NERF_AMD_EINVAL, "use_noise set but noise missing");
    if (Ni > 0 && !io->u && !io->t_lin_imp) return fail(NERF_AMD_EINVAL, "need u or t_lin_imp for sample_pdf");
    const nerf_amd_model *fm = fine ? fine : coarse;
    const int och = coarse->prog.out_ch;
    if (Ni > 0 && fm->prog.out_ch != och) return fail(NERF_AMD_EINVAL, "coarse and fine models disagree on output channels");
    if (och < 4) return fail(NERF_AMD_EINVAL, "field must output at least 4 channels");
    const bool has_vd = io->ray_ch > 8;
    if ((coarse->prog.arch.use_viewdirs != 0) != has_vd || (fm->prog.arch.use_viewdirs != 0) != has_vd)
        return fail(NERF_AMD_EINVAL, "ray batch width does not match the models' use_viewdirs");
    if (io->workspace_bytes < nerf_amd_render_rays_workspace(cfg, R, och) || !io->workspace)
        return fail(NERF_AMD_EINVAL, "workspace too small");
    p->cfg = cfg; p->coarse = coarse; p->fm = fm; p->io = io; p->R = R;
    p->Nc = Nc; p->Ni = Ni; p->Nf = Nf; p->och = och; p->has_vd = has_vd;

    char *w = static_cast<char *>(io->workspace);
    auto take = [&](size_t bytes) { float *q = reinterpret_cast<float *>(w); w += align_up(bytes); return q; };
    p->z_c = take(R * Nc * sizeof(float));
    p->raw_c = take(R * Nc * och * sizeof(float));
    p->w_c = take(R * Nc * sizeof(float));
    p->z_f = p->raw_f = nullptr;
    if (Ni > 0) {
        p->z_f = take(R * (size_t)Nf * sizeof(float));
        p->raw_f = take(R * (size_t)Nf * och * sizeof(float));
        if (io->z_vals) p->z_f = io->z_vals;
        if (io->raw) p->raw_f = io->raw;
    } else {
        if (io->z_vals) p->z_c = io->z_vals;
        if (io->raw) p->raw_c = io->raw;
        if (io->weights) p->w_c = io->weights;
    }
    return NERF_AMD_OK;
}

int stage_z(ChunkPlan &p, hipStream_t s) {
    const nerf_amd_render_io *io = p.io;
    int rc = 0;
    if (io->z_coarse) {
        if (p.Ni == 0 && io->z_vals)      // the caller wants z_vals back: they are its own coarse depths
            rc = hipMemcpyAsync(io->z_vals, io->z_coarse, p.R * p.Nc * sizeof(float), hipMemcpyDeviceToDevice, s) == hipSuccess
                     ? 0 : NERF_AMD_EHIP;
        else
            p.z_c = const_cast<float *>(io->z_coarse);
    } else {
        rc = launch_coarse_z(io->rays, io->ray_ch, io->t_vals, p.cfg->perturb ? io->t_rand : nullptr, p.R, p.Nc,
                             p.cfg->lindisp, p.cfg->perturb, p.z_c, s);
    }
    return rc ? fail(rc, "coarse_z launch failed") : NERF_AMD_OK;
}

int stage_field(const ChunkPlan &p, bool fine_pass, hipStream_t s) {
    MlpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.rays = p.io->rays; a.ray_stride = p.io->ray_ch;
    a.viewdirs = p.has_vd ? p.io->rays + 8 : nullptr; a.vd_stride = p.io->ray_ch;
    if (!fine_pass) {
        a.z_vals = p.z_c; a.P = p.R * p.Nc; a.S = p.Nc; a.out = p.raw_c;
        return run_field(p.coarse, a, p.cfg->precision, s);
    }
    a.z_vals = p.z_f; a.P = p.R * (int64_t)p.Nf; a.S = p.Nf; a.out = p.raw_f;
    return run_field(p.fm, a, p.cfg->precision, s);
}

CompositeJob final_job(const ChunkPlan &p) {       // compositing of the last pass of a chunk
    const nerf_amd_render_io *io = p.io;
    CompositeJob j;
    std::memset(&j, 0, sizeof(j));
    const bool two_pass = p.Ni > 0;
    j.raw = two_pass ? p.raw_f : p.raw_c; j.raw_ch = p.och; j.z = two_pass ? p.z_f : p.z_c;
    j.rays_d = io->rays + 3; j.rays_d_stride = io->ray_ch;
    j.noise = p.cfg->use_noise ? (two_pass ? io->noise1 : io->noise0) : nullptr;
    j.R = p.R; j.S = two_pass ? p.Nf : p.Nc; j.white_bkgd = p.cfg->white_bkgd;
    j.rgb = io->rgb_map; j.disp = io->disp_map; j.acc = io->acc_map; j.weights = two_pass ? io->weights : p.w_c;
    return j;
}

int stage_final(const ChunkPlan &p, hipStream_t s) {
    const CompositeJob j = final_job(p);
    int rc = launch_composite(j.raw, j.raw_ch, j.z, j.rays_d, j.rays_d_stride, j.noise, j.R, j.S, j.white_bkgd, j.rgb, j.disp,
                              j.acc, j.weights, nullptr, s);
    return rc ? fail(rc, "composite launch failed") : NERF_AMD_OK;
}

// coarse compositing + resampling of `p` (N_importance > 0), together with the final compositing of `prev` if given
int stage_mid(const ChunkPlan &p, const ChunkPlan *prev, hipStream_t s) {
    const nerf_amd_render_io *io = p.io;
    CompositeJob cj;
    std::memset(&cj, 0, sizeof(cj));
    cj.raw = p.raw_c; cj.raw_ch = p.och; cj.z = p.z_c; cj.rays_d = io->rays + 3; cj.rays_d_stride = io->ray_ch;
    cj.noise = p.cfg->use_noise ? io->noise0 : nullptr; cj.R = p.R; cj.S = p.Nc; cj.white_bkgd = p.cfg->white_bkgd;
    cj.rgb = io->rgb0; cj.disp = io->disp0; cj.acc = io->acc0; cj.weights = nullptr;
    ResampleJob rj;
    std::memset(&rj, 0, sizeof(rj));
    rj.u = io->u; rj.t_lin = io->t_lin_imp; rj.Ni = p.Ni; rj.z_fine = p.z_f; rj.z_std = io->z_std;
    CompositeJob fj;
    if (prev) fj = final_job(*prev);
    int rc = launch_mid_stage(cj, rj, prev ? &fj : nullptr, s);
    if (rc == NERF_AMD_EINVAL && resample_lds_bytes(p.Nc, p.Ni, true) > LDS_LIMIT_BYTES)
        return fail(rc, lds_msg("render_rays (compositing + resampling)", resample_lds_bytes(p.Nc, p.Ni, true)));
    return rc ? fail(rc, "composite/resample launch failed") : NERF_AMD_OK;
}
}  // namespace

extern "C" {

int nerf_amd_render_rays(const nerf_amd_render_cfg *cfg, const nerf_amd_model *coarse, const nerf_amd_model *fine,
                         const nerf_amd_render_io *io, int64_t R, void *stream) {
    if (R == 0 && cfg && coarse && io) return NERF_AMD_OK;
    ChunkPlan p;
    int rc = plan_chunk(cfg, coarse, fine, io, R, &p);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = stage_z(p, s)) || (rc = stage_field(p, false, s))) return rc;
    if (p.Ni == 0) return stage_final(p, s);
    if ((rc = stage_mid(p, nullptr, s)) || (rc = stage_field(p, true, s))) return rc;
    return stage_final(p, s);
}

int nerf_amd_render_chunks(const nerf_amd_render_cfg *cfg, const nerf_amd_model *coarse, const nerf_amd_model *fine,
                           const nerf_amd_render_io *ios, const int64_t *R, int32_t n_chunks, void *stream) {
    if (n_chunks < 0 || (n_chunks > 0 && (!ios || !R))) return fail(NERF_AMD_EINVAL, "bad chunk list");
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<ChunkPlan> plans;
    plans.reserve(n_chunks);
    for (int i = 0; i < n_chunks; ++i) {
        if (R[i] <= 0) continue;
        ChunkPlan p;
        int rc = plan_chunk(cfg, coarse, fine, &ios[i], R[i], &p);
        if (rc) return rc;
        plans.push_back(p);
    }
    const int n = (int)plans.size();
    // the final compositing of chunk k-1 runs while chunk k is under way: neighbours need their own workspaces
    for (int i = 0; i + 1 < n; ++i)
        if (plans[i].io->workspace == plans[i + 1].io->workspace)
            return fail(NERF_AMD_EINVAL, "consecutive chunks must use distinct workspaces");
    int rc = 0;
    for (int k = 0; k < n && !rc; ++k) {
        ChunkPlan &p = plans[k];
        if ((rc = stage_z(p, s)) || (rc = stage_field(p, false, s))) break;
        if (p.Ni == 0) { rc = stage_final(p, s); continue; }
        // one launch: this chunk's coarse compositing + resampling, and the previous chunk's final compositing
        if ((rc = stage_mid(p, k > 0 ? &plans[k - 1] : nullptr, s))) break;
        rc = stage_field(p, true, s);
    }
    if (!rc && n > 0 && plans[n - 1].Ni > 0) rc = stage_final(plans[n - 1], s);
    return rc;
}

}  // extern "C"

// ---- Renderer.render_batch as ONE call over contiguous whole-batch buffers -------------------------------------
namespace {
constexpr int64_t SUPER_CHUNK_RAYS = 32768;      // rays per launch group (the reference's own default chunk)
constexpr int BATCH_SLOTS = 3;                   // workspaces in rotation: a group is live from its coarse field to its final compositing

// The side stream and a pool of timing-less events per device.  Streams are created once; events are handed out and
// taken back under the mutex (an event can be re-recorded as soon as every wait on it has been ENQUEUED).
struct SideLane {
    hipStream_t stream = nullptr;
    hipStream_t more[2] = {nullptr, nullptr};      // further streams for work that splits more than two ways (lane_streams)
    std::vector<hipEvent_t> free_events;
};
std::mutex g_lane_mu;
SideLane g_lanes[64];

}  // namespace
namespace na {
namespace {
// Streams and events belong to the device that is current when they are created: make `device` current for the
// creation calls and put the caller's device back afterwards.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
}  // namespace

int lane_acquire(int device, int n_events, hipStream_t *side, std::vector<hipEvent_t> *events) {
    if (device < 0 || device >= 64) return fail(NERF_AMD_EINVAL, "device index out of range");
    std::lock_guard<std::mutex> lk(g_lane_mu);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice(lane device)");
    SideLane &l = g_lanes[device];
    if (!l.stream) HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
    *side = l.stream;
    events->clear();
    for (int i = 0; i < n_events; ++i) {
        hipEvent_t e;
        if (!l.free_events.empty()) { e = l.free_events.back(); l.free_events.pop_back(); }
        else if (hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming); err != hipSuccess) {
            for (hipEvent_t got : *events) l.free_events.push_back(got);      // hand back what was already taken
            events->clear();
            return hip_fail(err, "hipEventCreateWithFlags");
        }
        events->push_back(e);
    }
    return NERF_AMD_OK;
}
int lane_streams(int device, int n, hipStream_t *out) {
    if (device < 0 || device >= 64 || n < 1 || n > 3) return fail(NERF_AMD_EINVAL, "lane_streams: device or count out of range");
    std::lock_guard<std::mutex> lk(g_lane_mu);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice(lane device)");
    SideLane &l = g_lanes[device];
    if (!l.stream) HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
    out[0] = l.stream;
    for (int i = 1; i < n; ++i) {
        if (!l.more[i - 1]) HIP_TRY(hipStreamCreateWithFlags(&l.more[i - 1], hipStreamNonBlocking));
        out[i] = l.more[i - 1];
    }
    return NERF_AMD_OK;
}
void lane_release(int device, const std::vector<hipEvent_t> &events) {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    for (hipEvent_t e : events) g_lanes[device].free_events.push_back(e);
}
}  // namespace na
namespace {

// rows [r0, r0 + R) of a whole-batch io, with `ws` as its workspace
nerf_amd_render_io io_rows(const nerf_amd_render_cfg *cfg, const nerf_amd_render_io &io, int64_t r0, int och, void *ws, int64_t ws_bytes) {
    nerf_amd_render_io o = io;
    const int64_t Nc = cfg->N_samples, Ni = cfg->N_importance, Sl = Nc + Ni;
    auto f = [&](const float *p, int64_t row_floats) { return p ? p + r0 * row_floats : nullptr; };
    auto g = [&](float *p, int64_t row_floats) { return p ? p + r0 * row_floats : nullptr; };
    o.rays = f(io.rays, io.ray_ch);
    o.t_rand = f(io.t_rand, Nc); o.noise0 = f(io.noise0, Nc); o.noise1 = f(io.noise1, Sl);
    o.u = f(io.u, Ni); o.z_coarse = f(io.z_coarse, Nc);
    o.rgb_map = g(io.rgb_map, 3); o.disp_map = g(io.disp_map, 1); o.acc_map = g(io.acc_map, 1);
    o.rgb0 = g(io.rgb0, 3); o.disp0 = g(io.disp0, 1); o.acc0 = g(io.acc0, 1); o.z_std = g(io.z_std, 1);
    o.raw = g(io.raw, Sl * och); o.weights = g(io.weights, Sl); o.z_vals = g(io.z_vals, Sl);
    o.workspace = ws; o.workspace_bytes = ws_bytes;
    return o;
}

int64_t n_super_chunks(int64_t N) { return N <= SUPER_CHUNK_RAYS + SUPER_CHUNK_RAYS / 2 ? 1 : (N + SUPER_CHUNK_RAYS - 1) / SUPER_CHUNK_RAYS; }
}  // namespace

extern "C" {

int64_t nerf_amd_render_batch_workspace(const nerf_amd_render_cfg *cfg, int64_t N, int32_t out_ch) {
    if (!cfg || N < 0) return -1;
    const int64_t n = n_super_chunks(N);
    const int64_t per = nerf_amd_render_rays_workspace(cfg, n == 1 ? N : SUPER_CHUNK_RAYS, out_ch);
    return per * (n < BATCH_SLOTS ? n : BATCH_SLOTS);
}

int nerf_amd_render_batch(const nerf_amd_render_cfg *cfg, const nerf_amd_model *coarse, const nerf_amd_model *fine,
                          const nerf_amd_render_io *io, int64_t N, void *stream) {
    if (!cfg || !coarse || !io || N < 0) return fail(NERF_AMD_EINVAL, "null argument");
    if (N == 0) return NERF_AMD_OK;
    const int64_t n = n_super_chunks(N);
    if (n == 1) return nerf_amd_render_rays(cfg, coarse, fine, io, N, stream);
    const int och = coarse->prog.out_ch;
    const int64_t per = nerf_amd_render_rays_workspace(cfg, SUPER_CHUNK_RAYS, och);
    const int slots = (int)(n < BATCH_SLOTS ? n : BATCH_SLOTS);
    if (!io->workspace || io->workspace_bytes < per * slots) return fail(NERF_AMD_EINVAL, "workspace too small (nerf_amd_render_batch_workspace)");
    if (!io->z_coarse && cfg->perturb && !io->t_rand) return fail(NERF_AMD_EINVAL, "perturb set but neither z_coarse nor t_rand given");

    std::vector<nerf_amd_render_io> ios((size_t)n);
    std::vector<ChunkPlan> plans((size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        const int64_t r0 = k * SUPER_CHUNK_RAYS, R = (k + 1 == n) ? N - r0 : SUPER_CHUNK_RAYS;
        ios[k] = io_rows(cfg, *io, r0, och, static_cast<char *>(io->workspace) + (k % slots) * per, per);
        int rc = plan_chunk(cfg, coarse, fine, &ios[k], R, &plans[k]);
        if (rc) return rc;
    }
    hipStream_t main_s = static_cast<hipStream_t>(stream), side = nullptr;
    const int device = coarse->device;
    const bool two_pass = plans[0].Ni > 0;
    std::vector<hipEvent_t> ev;
    int rc = lane_acquire(device, (int)(3 * n + 1), &side, &ev);
    if (rc) return rc;
    hipEvent_t *ec = ev.data(), *em = ev.data() + n, *ef = ev.data() + 2 * n;   // coarse field done / side-stream stage done / fine field done
    hipEvent_t start = ev[3 * n];
    auto H = [&](hipError_t e, const char *what) { if (e != hipSuccess && !rc) rc = hip_fail(e, what); };
    // the side stream starts behind everything the caller has enqueued so far (inputs, packed weights)
    H(hipEventRecord(start, main_s), "hipEventRecord");
    H(hipStreamWaitEvent(side, start, 0), "hipStreamWaitEvent");

    auto coarse_field = [&](int64_t k) {
        if (k >= slots) H(hipStreamWaitEvent(main_s, em[k - slots + (two_pass ? 1 : 0)], 0), "hipStreamWaitEvent");   // its workspace is free again
        if (!rc) rc = stage_z(plans[k], main_s);
        if (!rc) rc = stage_field(plans[k], false, main_s);
        H(hipEventRecord(ec[k], main_s), "hipEventRecord");
    };
    if (two_pass) {
        // main:  C0 C1 F0 C2 F1 C3 F2 ...        side:  M0  M1+Fin0  M2+Fin1 ... Fin(n-1)
        // M = coarse compositing + resampling (needs C), Fin = final compositing (needs F), both per-ray kernels that run
        // beside the next field kernel instead of between two of them.  A/B 44 (round 4): the fine-pass field kernels on a
        // stream of their own -- F(k) needs M(k), not C(k+1), and with both in flight the workgroups of the one fill the CUs
        // the other's last tiles leave free: 0.4 % faster per view (tools/micro/view_ab.py), bit-identical.  Not the default:
        // with two field kernels resident at once a launch's duration -- what the roofline is read from, in bench.py's events
        // and in rocprofv3's kernel stats alike -- no longer says what the kernel does.
        hipStream_t fine_s = main_s;
        hipStream_t lanes2[2];
        if (g_variant == 44 && lane_streams(device, 2, lanes2) == NERF_AMD_OK) {
            fine_s = lanes2[1];
            H(hipStreamWaitEvent(fine_s, start, 0), "hipStreamWaitEvent");
        }
        coarse_field(0);
        for (int64_t k = 0; k < n && !rc; ++k) {
            if (k + 1 < n) coarse_field(k + 1);
            H(hipStreamWaitEvent(side, ec[k], 0), "hipStreamWaitEvent");
            if (k > 0) H(hipStreamWaitEvent(side, ef[k - 1], 0), "hipStreamWaitEvent");
            if (!rc) rc = stage_mid(plans[k], k > 0 ? &plans[k - 1] : nullptr, side);
            H(hipEventRecord(em[k], side), "hipEventRecord");
            H(hipStreamWaitEvent(fine_s, em[k], 0), "hipStreamWaitEvent");
            if (!rc) rc = stage_field(plans[k], true, fine_s);
            H(hipEventRecord(ef[k], fine_s), "hipEventRecord");
        }
        H(hipStreamWaitEvent(side, ef[n - 1], 0), "hipStreamWaitEvent");
        if (!rc) rc = stage_final(plans[n - 1], side);
    } else {
        for (int64_t k = 0; k < n && !rc; ++k) {
            coarse_field(k);
            H(hipStreamWaitEvent(side, ec[k], 0), "hipStreamWaitEvent");
            if (!rc) rc = stage_final(plans[k], side);
            H(hipEventRecord(em[k], side), "hipEventRecord");
        }
    }
    // the caller's stream continues behind the last per-ray kernel
    H(hipEventRecord(start, side), "hipEventRecord");
    H(hipStreamWaitEvent(main_s, start, 0), "hipStreamWaitEvent");
    lane_release(device, ev);
    return rc;
}

}  // extern "C"

extern "C" {

int nerf_amd_set_tuning(int key, int value) {
    if (key == 0 && value >= 0 && value <= 115) { g_variant.store(value, std::memory_order_relaxed); return NERF_AMD_OK; }
    return fail(NERF_AMD_EINVAL, "unknown tuning key/value");
}

int nerf_amd_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return NERF_AMD_OK;
}

int nerf_amd_profile_collect(int64_t launches[3], double total_ms[3], double total_points[3]) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int c = 0; c < 3; ++c) { launches[c] = 0; total_ms[c] = 0.0; total_points[c] = 0.0; }
    for (const ProfRec &r : g_prof_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            launches[r.cls] += 1; total_ms[r.cls] += ms; total_points[r.cls] += r.points;
        }
        g_prof_free.emplace_back(r.a, r.b);
    }
    g_prof_recs.clear();
    return NERF_AMD_OK;
}

int nerf_amd_get_rays_backward(int32_t H, int32_t W, const double *K4, int64_t pix0, int64_t n, const float *g_rays_o,
                               const float *g_rays_d, float *g_c2w, void *stream) {
    if (H < 1 || W < 1 || !K4 || pix0 < 0 || n < 0 || pix0 + n > (int64_t)H * W || !g_c2w)
        return fail(NERF_AMD_EINVAL, "bad get_rays_backward arguments");
    int rc = launch_get_rays_bwd(H, W, K4, pix0, n, g_rays_o, g_rays_d, g_c2w, static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "get_rays backward launch failed") : NERF_AMD_OK;
}

int nerf_amd_img2mse(const float *x, const float *y, int64_t n, float *out, float *partials, void *stream) {
    if (n < 1 || !x || !y || !out) return fail(NERF_AMD_EINVAL, "bad img2mse arguments");
    int rc = na::launch_img2mse(x, y, n, out, partials, static_cast<hipStream_t>(stream));
    return rc ? fail(rc, rc == NERF_AMD_EINVAL ? "img2mse: n > 16384 needs the partials buffer" : "img2mse launch failed") : NERF_AMD_OK;
}

int nerf_amd_img2mse_backward(const float *x, const float *y, int64_t n, const float *g, float *gx, float *gy, void *stream) {
    if (n < 1 || !x || !y || !g) return fail(NERF_AMD_EINVAL, "bad img2mse_backward arguments");
    int rc = na::launch_img2mse_bwd(x, y, n, g, gx, gy, static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "img2mse_backward launch failed") : NERF_AMD_OK;
}

int nerf_amd_assemble_rays(const float *rays_o, const float *rays_d, const float *viewdir_src, int64_t n, float near,
                           float far, float *out, void *stream) {
    if (n < 0 || (n > 0 && (!rays_o || !rays_d || !out))) return fail(NERF_AMD_EINVAL, "bad assemble_rays arguments");
    int rc = na::launch_assemble_rays(rays_o, rays_d, viewdir_src, n, near, far, out, static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "assemble_rays launch failed") : NERF_AMD_OK;
}

int nerf_amd_adam_step(int32_t n, float *const *params, const float *const *grads, float *const *exp_avg,
                       float *const *exp_avg_sq, const int64_t *numel, int64_t step, double lr, double beta1,
                       double beta2, double eps, double weight_decay, void *stream) {
    if (n < 0 || step < 1 || (n > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel)))
        return fail(NERF_AMD_EINVAL, "bad adam arguments");
    for (int i = 0; i < n; ++i) {
        if (numel[i] < 0 || numel[i] > 0x7fffffff) return fail(NERF_AMD_EINVAL, "adam: tensor size out of range");
        if (numel[i] > 0 && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i])) return fail(NERF_AMD_EINVAL, "adam: null tensor pointer");
    }
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    int rc = na::launch_adam(n, params, grads, exp_avg, exp_avg_sq, numel, (float)(lr / bc1), beta1, beta2,
                             (float)eps, (float)weight_decay, (float)std::sqrt(bc2), static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "adam launch failed") : NERF_AMD_OK;
}

int nerf_amd_adam_step_device(int32_t n, float *const *params, const float *const *grads, float *const *exp_avg,
                              float *const *exp_avg_sq, const int64_t *numel, int64_t *step_dev, const double *lr_dev, double beta1,
                              double beta2, double eps, double weight_decay, float *scalars_dev, void *stream) {
    if (n < 0 || !step_dev || !lr_dev || !scalars_dev || (n > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel)))
        return fail(NERF_AMD_EINVAL, "bad adam arguments");
    if (n > 64) return fail(NERF_AMD_EINVAL, "adam (device scalars): at most 64 tensors per call (one count advance per call)");
    for (int i = 0; i < n; ++i) {
        if (numel[i] < 0 || numel[i] > 0x7fffffff) return fail(NERF_AMD_EINVAL, "adam: tensor size out of range");
        if (numel[i] > 0 && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i])) return fail(NERF_AMD_EINVAL, "adam: null tensor pointer");
    }
    int rc = na::launch_adam(n, params, grads, exp_avg, exp_avg_sq, numel, 0.0f, beta1, beta2, (float)eps, (float)weight_decay, 1.0f,
                             static_cast<hipStream_t>(stream), step_dev, lr_dev, scalars_dev);
    return rc ? fail(rc, "adam launch failed") : NERF_AMD_OK;
}

int nerf_amd_to8b(const float *x, int64_t n, uint8_t *out, void *stream) {
    if (n < 0 || (n > 0 && (!x || !out))) return fail(NERF_AMD_EINVAL, "bad to8b arguments");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 3))
        return fail(NERF_AMD_EINVAL, "to8b: x must be 16-byte and out 4-byte aligned");
    int rc = launch_to8b(x, n, out, static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "to8b launch failed") : NERF_AMD_OK;
}

int nerf_amd_make_rays(int32_t H, int32_t W, const double *K4, const float *c2w, const float *c2w_static,
                       int64_t pix0, int64_t n, float near, float far, int use_viewdirs, int ndc,
                       float *rays_out, void *stream) {
    if (H < 1 || W < 1 || !K4 || !c2w || pix0 < 0 || n < 0 || pix0 + n > (int64_t)H * W || (n > 0 && !rays_out))
        return fail(NERF_AMD_EINVAL, "bad make_rays arguments");
    int rc = launch_make_rays(H, W, K4, c2w, c2w_static, pix0, n, near, far, use_viewdirs, ndc, rays_out,
                              static_cast<hipStream_t>(stream));
    return rc ? fail(rc, "make_rays launch failed") : NERF_AMD_OK;
}

}  // extern "C"
