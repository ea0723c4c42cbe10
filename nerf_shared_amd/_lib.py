"""ctypes binding of libnerf_amd.so (C ABI: include/nerf_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C nerf_shared_amd/csrc``.  There is no CPU or PyTorch fallback: if the
library is missing, importing this module raises, and every product entry
point refuses non-ROCm tensors.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint16, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# NERF_AMD_LIB selects another build of the same library (diagnostic builds such as -DNERF_AMD_STAMPS)
LIB_PATH = os.environ.get("NERF_AMD_LIB") or os.path.join(_HERE, "libnerf_amd.so")

ABI_VERSION = 7
PREC_FP32, PREC_BF16, PREC_FP32_SPLIT = 0, 1, 2
# packed copies of a model's parameters (include/nerf_amd.h NERF_AMD_COPY_*) and which of them a call needs
COPY_BF16, COPY_BWD, COPY_SPLIT, COPY_BWD_SPLIT, COPY_FP32, COPY_FP32_BWD, COPY_ALL = 1, 2, 4, 8, 16, 32, 63
COPY_OF = {PREC_BF16: COPY_BF16, PREC_FP32_SPLIT: COPY_SPLIT, PREC_FP32: COPY_FP32}                    # inference
TRAIN_COPIES = {PREC_BF16: COPY_BF16 | COPY_BWD, PREC_FP32_SPLIT: COPY_SPLIT | COPY_BWD_SPLIT,       # forward_train + backward
                PREC_FP32: COPY_FP32 | COPY_FP32_BWD}
MAX_SKIPS = 8

EXPORTS = (
    "nerf_amd_abi_version", "nerf_amd_last_error",
    "nerf_amd_model_create", "nerf_amd_model_update", "nerf_amd_model_update_copies", "nerf_amd_model_destroy",
    "nerf_amd_model_supports_bf16", "nerf_amd_model_supports_split", "nerf_amd_model_out_ch", "nerf_amd_pack_bf16_host",
    "nerf_amd_embed", "nerf_amd_nerf_forward", "nerf_amd_mlp_embedded", "nerf_amd_ndc_rays", "nerf_amd_raw2outputs", "nerf_amd_raw2outputs_backward", "nerf_amd_sample_pdf",
    "nerf_amd_render_rays_workspace", "nerf_amd_render_rays", "nerf_amd_render_chunks", "nerf_amd_make_rays",
    "nerf_amd_render_batch_workspace", "nerf_amd_render_batch",
    "nerf_amd_profile_enable", "nerf_amd_profile_collect", "nerf_amd_set_tuning",
    "nerf_amd_model_supports_training", "nerf_amd_train_workspace", "nerf_amd_field_forward_train",
    "nerf_amd_field_backward", "nerf_amd_coarse_z", "nerf_amd_resample", "nerf_amd_get_rays_backward", "nerf_amd_to8b", "nerf_amd_ndc_rays_backward", "nerf_amd_adam_step", "nerf_amd_adam_step_device", "nerf_amd_img2mse", "nerf_amd_img2mse_backward", "nerf_amd_assemble_rays",
)


class Arch(Structure):
    _fields_ = [("D", c_int32), ("W", c_int32), ("output_ch", c_int32), ("use_viewdirs", c_int32),
                ("multires", c_int32), ("multires_views", c_int32), ("i_embed", c_int32),
                ("n_skips", c_int32), ("skips", c_int32 * MAX_SKIPS)]


class RenderCfg(Structure):
    _fields_ = [("N_samples", c_int32), ("N_importance", c_int32), ("perturb", c_int32), ("lindisp", c_int32),
                ("white_bkgd", c_int32), ("use_noise", c_int32), ("precision", c_int32), ("reserved", c_int32)]


class RenderIO(Structure):
    _fields_ = [("rays", c_void_p), ("ray_ch", c_int32), ("pad0", c_int32),
                ("t_vals", c_void_p), ("t_rand", c_void_p), ("noise0", c_void_p), ("noise1", c_void_p),
                ("u", c_void_p), ("t_lin_imp", c_void_p), ("z_coarse", c_void_p),
                ("rgb_map", c_void_p), ("disp_map", c_void_p), ("acc_map", c_void_p),
                ("rgb0", c_void_p), ("disp0", c_void_p), ("acc0", c_void_p),
                ("z_std", c_void_p), ("raw", c_void_p), ("weights", c_void_p), ("z_vals", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_int64)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "nerf_shared_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nerf_shared_amd/csrc` (needs hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    pp_f = POINTER(c_void_p)
    sig = {
        "nerf_amd_abi_version": (c_int, []),
        "nerf_amd_last_error": (c_char_p, []),
        "nerf_amd_model_create": (c_int, [POINTER(Arch), c_int, POINTER(c_void_p)]),
        "nerf_amd_model_update": (c_int, [c_void_p, pp_f, pp_f, c_int, c_void_p]),
        "nerf_amd_model_update_copies": (c_int, [c_void_p, pp_f, pp_f, c_int, c_int, c_int, c_void_p]),
        "nerf_amd_model_destroy": (None, [c_void_p]),
        "nerf_amd_model_supports_bf16": (c_int, [c_void_p]),
        "nerf_amd_model_supports_split": (c_int, [c_void_p]),
        "nerf_amd_model_out_ch": (c_int, [c_void_p]),
        "nerf_amd_pack_bf16_host": (c_int, [POINTER(Arch), c_int, pp_f, pp_f, c_int, POINTER(c_uint16), POINTER(c_int64),
                                            POINTER(c_float), POINTER(c_int64)]),
        "nerf_amd_embed": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
        "nerf_amd_nerf_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int, c_void_p]),
        "nerf_amd_mlp_embedded": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
        "nerf_amd_ndc_rays": (c_int, [c_int32, c_int32, c_double, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_void_p]),
        "nerf_amd_raw2outputs": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32,
                                         c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
        "nerf_amd_raw2outputs_backward": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32,
                                                  c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                  c_void_p, c_void_p]),
        "nerf_amd_sample_pdf": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                        c_void_p, c_void_p]),
        "nerf_amd_render_rays_workspace": (c_int64, [POINTER(RenderCfg), c_int64, c_int32]),
        "nerf_amd_render_rays": (c_int, [POINTER(RenderCfg), c_void_p, c_void_p, POINTER(RenderIO), c_int64, c_void_p]),
        "nerf_amd_render_chunks": (c_int, [POINTER(RenderCfg), c_void_p, c_void_p, POINTER(RenderIO), POINTER(c_int64),
                                           c_int32, c_void_p]),
        "nerf_amd_render_batch_workspace": (c_int64, [POINTER(RenderCfg), c_int64, c_int32]),
        "nerf_amd_render_batch": (c_int, [POINTER(RenderCfg), c_void_p, c_void_p, POINTER(RenderIO), c_int64, c_void_p]),
        "nerf_amd_make_rays": (c_int, [c_int32, c_int32, POINTER(c_double), POINTER(c_float), POINTER(c_float),
                                       c_int64, c_int64, c_float, c_float, c_int, c_int, c_void_p, c_void_p]),
        "nerf_amd_set_tuning": (c_int, [c_int, c_int]),
        "nerf_amd_coarse_z": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_int, c_int, c_void_p, c_void_p]),
        "nerf_amd_resample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                      c_void_p]),
        "nerf_amd_model_supports_training": (c_int, [c_void_p, c_int]),
        "nerf_amd_train_workspace": (c_int64, [c_void_p, c_int64, c_int]),
        "nerf_amd_field_forward_train": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32,
                                                 c_void_p, c_void_p, c_int64, c_int, c_void_p]),
        "nerf_amd_field_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32,
                                            c_void_p, c_int64, pp_f, pp_f, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
        "nerf_amd_get_rays_backward": (c_int, [c_int32, c_int32, POINTER(c_double), c_int64, c_int64, c_void_p, c_void_p,
                                               c_void_p, c_void_p]),
        "nerf_amd_to8b": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
        "nerf_amd_img2mse": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
        "nerf_amd_img2mse_backward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
        "nerf_amd_assemble_rays": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p]),
        "nerf_amd_adam_step": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double,
                               c_double, c_double, c_double, c_void_p]),
        "nerf_amd_adam_step_device": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double,
                                      c_double, c_double, c_double, c_void_p, c_void_p]),
        "nerf_amd_ndc_rays_backward": (c_int, [c_int32, c_int32, c_double, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int64, c_void_p, c_void_p, c_void_p]),
        "nerf_amd_profile_enable": (c_int, [c_int]),
        "nerf_amd_profile_collect": (c_int, [POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.nerf_amd_abi_version() != ABI_VERSION:
        raise ImportError("libnerf_amd.so ABI version %d, binding expects %d" % (lib.nerf_amd_abi_version(), ABI_VERSION))
    return lib


lib = _load()


class NerfAmdError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = lib.nerf_amd_last_error()
        raise NerfAmdError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def make_arch(D, W, output_ch, skips, use_viewdirs, multires, multires_views, i_embed):
    skips = [int(s) for s in skips]
    if len(skips) > MAX_SKIPS:
        raise ValueError("at most %d skip connections are supported" % MAX_SKIPS)
    a = Arch()
    a.D, a.W, a.output_ch, a.use_viewdirs = int(D), int(W), int(output_ch), int(bool(use_viewdirs))
    a.multires, a.multires_views, a.i_embed, a.n_skips = int(multires), int(multires_views), int(i_embed), len(skips)
    for i, s in enumerate(skips):
        a.skips[i] = s
    return a


def ptr(t):
    """Device pointer of a tensor, or None."""
    return None if t is None else t.data_ptr()


def require_device(t, name):
    """The product runs on ROCm devices only -- no CPU fallback."""
    if not t.is_cuda:
        raise NerfAmdError("%s is on %s; nerf_shared_amd runs on ROCm (cuda) tensors only -- there is no CPU path"
                           % (name, t.device))


def stream_of(device):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
