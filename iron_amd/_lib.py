"""ctypes binding of libiron_hip.so (include/iron_hip.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libiron_hip.so")

IRON_OK = 0
IRON_NET_NERF = 2
IRON_ERR_UNSUPPORTED = -2
IRON_NET_SDF, IRON_NET_RENDER = 0, 1
MODES = {"idr": 0, "no_view_dir": 1, "no_normal": 2, "points_only": 3}


class IronError(RuntimeError):
    pass


class iron_linear(C.Structure):
    _fields_ = [("weight_v", C.c_void_p), ("weight_g", C.c_void_p), ("bias", C.c_void_p),
                ("out_dim", C.c_int32), ("in_dim", C.c_int32)]


class iron_net_desc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_linear", C.c_int32), ("d_hidden", C.c_int32), ("d_out", C.c_int32),
                ("multires", C.c_int32), ("multires_view", C.c_int32), ("skip_layer", C.c_int32),
                ("mode", C.c_int32), ("d_feature", C.c_int32), ("squeeze_out", C.c_int32),
                ("squeeze_out_scale", C.c_float), ("output_bias", C.c_float), ("output_scale", C.c_float),
                ("scale", C.c_float)]


class iron_composite_params(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("diffuse_albedo", "specular_albedo", "specular_roughness", "metallic_eta",
                                          "metallic_k", "dielectric_eta", "env_light")]


class iron_neus_composite_args(C.Structure):
    _fields_ = ([(k, C.c_void_p) for k in ("dists", "pts", "dirs", "sdf", "grad", "color", "bg_dists", "bg_density", "bg_color",
                                           "background_rgb")]
                + [("n", C.c_int64), ("m", C.c_int32), ("mo", C.c_int32), ("inv_s", C.c_float), ("cos_anneal_ratio", C.c_float)]
                + [(k, C.c_void_p) for k in ("out_color", "weights", "cdf", "inside_sphere", "weight_sum", "weight_max",
                                             "gradient_error_acc")])


class iron_trace_params(C.Structure):
    _fields_ = [("sdf_threshold", C.c_float), ("sphere_tracing_iters", C.c_int32), ("n_steps", C.c_int32),
                ("chunk", C.c_int64)]


class iron_shade_nets(C.Structure):
    _fields_ = [("sdf", C.c_void_p), ("diffuse_albedo", C.c_void_p), ("specular_albedo", C.c_void_p),
                ("specular_roughness", C.c_void_p)]


class iron_shade_comp_nets(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("sdf", "diffuse_albedo", "specular_albedo", "specular_roughness", "metallic", "dielectric",
                                          "metallic_eta", "metallic_k", "dielectric_eta")]


COMP_OUT_FIELDS = ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo", "specular_roughness", "metallic_eta",
                   "metallic_k", "dielectric_eta", "normal", "metallic_rgb", "metallic", "dielectric_rgb", "dielectric")


class iron_shade_comp_out(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in COMP_OUT_FIELDS]


class iron_shade_out(C.Structure):
    _fields_ = [("color", C.c_void_p), ("diffuse_color", C.c_void_p), ("specular_color", C.c_void_p),
                ("diffuse_albedo", C.c_void_p), ("specular_albedo", C.c_void_p),
                ("specular_roughness", C.c_void_p), ("normal", C.c_void_p)]


TRACE_STATS_FIELDS = ("n_evals", "n_sphere_conv", "n_sampler", "n_bisect", "n_conv", "n_evals_ref", "n_evals_sphere",
                      "reserved")
PROF_KINDS = ("sphere", "sampler", "bisect_a", "bisect_b", "sdf_grad", "material", "ggx", "sdf_forward")

# every symbol include/iron_hip.h declares: name -> (restype, argtypes)
_P, _I32, _I64, _F, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
SYMBOLS = {
    "iron_version": (C.c_int, []),
    "iron_strerror": (C.c_char_p, [C.c_int]),
    "iron_last_hip_error": (C.c_int, []),
    "iron_net_create": (C.c_int, [C.POINTER(_P), C.POINTER(iron_net_desc), C.POINTER(iron_linear), _P]),
    "iron_net_destroy": (C.c_int, [_P]),
    "iron_sdf_forward": (C.c_int, [_P, _P, _I64, _P, _I32, _P]),
    "iron_sdf_get_all_workspace_bytes": (_SZ, [_P, _I64]),
    "iron_sdf_get_all": (C.c_int, [_P, _P, _I64, _P, _P, _P, _P, _SZ, _P]),
    "iron_render_forward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P]),
    "iron_camera_rays": (C.c_int, [C.POINTER(_F), C.POINTER(_F), _P, _I64, _P, _P, _P, _P]),
    "iron_intersect_sphere": (C.c_int, [_P, _P, _I64, _F, _P, _P, _P, _P]),
    "iron_ggx_colocated": (C.c_int, [_F, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "iron_composite_colocated": (C.c_int, [C.c_float, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P]),
    "iron_coloc_head": (C.c_int, [_I32, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "iron_grid_points": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "iron_neus_linspace": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _P]),
    "iron_neus_outside_z": (C.c_int, [_P, _P, _I64, _I32, _F, _P, _P]),
    "iron_neus_points": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _P]),
    "iron_neus_up_sample": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _F, _P, _P]),
    "iron_neus_merge": (C.c_int, [_P, _P, _I32, _P, _P, _I32, _I64, _P, _P, _P]),
    "iron_neus_mid_points": (C.c_int, [_P, _P, _P, _I64, _I32, _F, _I32, _P, _P, _P, _P]),
    "iron_neus_need_background": (C.c_int, [_P, _I64, _I32, _I32, _P, _P]),
    "iron_neus_composite": (C.c_int, [C.POINTER(iron_neus_composite_args), _P]),
    "iron_neus_composite_alpha": (C.c_int, [C.POINTER(iron_neus_composite_args), _P, _P]),
    "iron_neus_outside_composite": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P]),
    "iron_neus_sample_pdf": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "iron_smith_g1": (C.c_int, [_P, _P, _I64, _P, _P]),
    "iron_nerf_forward": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P]),
    "iron_edge_walk": (C.c_int, [_P, _P, _I64, _P, _I32, _F, _F, _P, _P, _P]),
    "iron_morph_closing3x3": (C.c_int, [_P, _I32, _I32, _P, _P, _P]),
    "iron_sobel_magnitude": (C.c_int, [_P, _I32, _I32, _P, _P]),
    "iron_edge_pixels": (C.c_int, [_P, _P, _I64, C.POINTER(_F), C.POINTER(_F), _I32, _I32, _P, _P, _P]),
    "iron_fill_holes": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P]),
    "iron_edge_sides": (C.c_int, [_P, _P, C.POINTER(_F), _I64, _P, _P, _P]),
    "iron_edge_blend": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "iron_trace_workspace_bytes": (_SZ, [_I64, C.POINTER(iron_trace_params)]),
    "iron_trace": (C.c_int, [_P, C.POINTER(iron_trace_params), _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P,
                             _P, _SZ, _P]),
    "iron_trace_stage": (C.c_int, [_I32, _P, C.POINTER(iron_trace_params), _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P,
                                   _P, _SZ, _P]),
    "iron_trace_phase": (C.c_int, [_I32, _P, C.POINTER(iron_trace_params), _P, _P, _P, _P, _P, _P, _P, _I64, _P,
                                   _I64, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "iron_net_numeric_status": (C.c_int, [_P, C.POINTER(C.c_int32), _P]),
    "iron_net_force_exact": (C.c_int, [_P, _I32]),
    "iron_set_cu_limit": (_I32, [_I32]),
    "iron_set_trace_split": (_I32, [_I32]),
    "iron_profile_enable": (C.c_int, [_I32]),
    "iron_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(_I64)]),
    "iron_shade_workspace_bytes": (_SZ, [_I64]),
    "iron_shade_composite_workspace_bytes": (_SZ, [_I64]),
    "iron_shade_composite": (C.c_int, [C.POINTER(iron_shade_comp_nets), _F, _P, _P, _P, _P, _P, _P, _I64,
                                       C.POINTER(iron_shade_comp_out), _P, _SZ, _P]),
    "iron_shade_ggx": (C.c_int, [C.POINTER(iron_shade_nets), _F, _I32, _P, _P, _P, _P, _P, _P, _I64,
                                 C.POINTER(iron_shade_out), _P, _SZ, _P]),
}

# ---- libiron_train.so (include/iron_train.h): backward passes, loaded on first use ----
class iron_train_layer(C.Structure):
    _fields_ = [("weight_v", C.c_void_p), ("weight_g", C.c_void_p), ("bias", C.c_void_p), ("d_weight_v", C.c_void_p),
                ("d_weight_g", C.c_void_p), ("d_bias", C.c_void_p), ("out_dim", C.c_int32), ("in_dim", C.c_int32)]


class iron_sdf_train_desc(C.Structure):
    _fields_ = [("n_linear", C.c_int32), ("multires", C.c_int32), ("skip_layer", C.c_int32), ("layers", C.POINTER(iron_train_layer))]


class iron_render_train_desc(C.Structure):
    _fields_ = [("n_linear", C.c_int32), ("mode", C.c_int32), ("multires", C.c_int32), ("multires_view", C.c_int32),
                ("d_feature", C.c_int32), ("d_out", C.c_int32), ("skip_layer", C.c_int32), ("squeeze_out", C.c_int32),
                ("output_bias", C.c_float), ("output_scale", C.c_float), ("squeeze_out_scale", C.c_float),
                ("layers", C.POINTER(iron_train_layer))]


class iron_composite_grads_in(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("d_rgb", "d_specular_rgb", "d_metallic_rgb", "d_dielectric_rgb", "d_env_light_out")]


class iron_composite_grads_out(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("d_light", "d_distance", "d_normal", "d_viewdir", "d_diffuse_albedo", "d_specular_albedo",
                                          "d_specular_roughness", "d_metallic_eta", "d_metallic_k", "d_dielectric_eta", "d_env_light")]


class iron_nerf_train_desc(C.Structure):
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("d_in", C.c_int32), ("d_in_view", C.c_int32), ("multires", C.c_int32),
                ("multires_view", C.c_int32), ("skip", C.c_int32), ("layers", C.POINTER(iron_train_layer))]


class iron_neus_composite_grads(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("d_color", "d_weight_sum", "d_weights", "d_gradient_error", "relax_count", "d_sdf", "d_grad",
                                          "d_sample_color", "d_inv_s", "d_bg_density", "d_bg_color")]


TRAIN_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libiron_train.so")
TRAIN_SYMBOLS = {
    "iron_train_gemm_workspace_bytes": (_SZ, [_I32, _I32, _I32]),
    "iron_train_gemm": (C.c_int, [_I32, _I32, _I32, _I32, _I32, _P, _I32, _P, _I32, _F, _P, _I32, _P, _SZ, _P]),
    "iron_sdf_backward_workspace_bytes": (_SZ, [C.POINTER(iron_sdf_train_desc), _I64]),
    "iron_sdf_backward": (C.c_int, [C.POINTER(iron_sdf_train_desc), _P, _I64, _P, _P, _P, _P, _SZ, _P]),
    "iron_render_backward_workspace_bytes": (_SZ, [C.POINTER(iron_render_train_desc), _I64]),
    "iron_render_backward": (C.c_int, [C.POINTER(iron_render_train_desc), _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "iron_ggx_colocated_backward": (C.c_int, [_F] + [_P] * 8 + [_I64] + [_P] * 10 + [_P]),
    "iron_composite_colocated_backward": (C.c_int, [_F, _P, _P, _P, C.POINTER(iron_composite_params), _P, _P, _I64,
                                                    C.POINTER(iron_composite_grads_in), C.POINTER(iron_composite_grads_out), _P]),
    "iron_coloc_head_backward": (C.c_int, [_I32, _F, _F, _F] + [_P] * 6 + [_I64] + [_P] * 10 + [_P]),
    "iron_nerf_backward_workspace_bytes": (_SZ, [C.POINTER(iron_nerf_train_desc), _I64]),
    "iron_nerf_backward": (C.c_int, [C.POINTER(iron_nerf_train_desc), _P, _P, _I64, _P, _P, _P, _SZ, _P]),
    "iron_neus_composite_backward": (C.c_int, [C.POINTER(iron_neus_composite_args), C.POINTER(iron_neus_composite_grads), _P]),
    "iron_train_last_hip_error": (C.c_int, []),
    "iron_train_last_blas_status": (C.c_int, []),
    "iron_train_numeric_status": (C.c_int, [_I32, _P]),
}

_lock = threading.Lock()
_lib: Optional[C.CDLL] = None
_train_lib: Optional[C.CDLL] = None


def load_train() -> C.CDLL:
    """dlopen libiron_train.so (the backward passes) and bind every symbol of include/iron_train.h."""
    global _train_lib
    with _lock:
        if _train_lib is not None:
            return _train_lib
        path = os.environ.get("IRON_TRAIN_LIB") or TRAIN_LIB_PATH
        if not os.path.exists(path):
            raise IronError("libiron_train.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no autograd fallback for the HIP operators)" % path)
        lib = C.CDLL(path)
        for name, (res, args) in TRAIN_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _train_lib = lib
        return lib


def check_train(status: int) -> None:
    if status != IRON_OK:
        lib = load_train()
        msg = load().iron_strerror(status).decode()
        if status == -3:
            msg += " [hipError_t=%d]" % lib.iron_train_last_hip_error()
        raise IronError("libiron_train: %s" % msg)


def load() -> C.CDLL:
    """dlopen the library and bind every declared symbol; raises if anything is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = os.environ.get("IRON_HIP_LIB") or LIB_PATH  # override: A/B of kernel variants (tools/variants.py)
        if not os.path.exists(path):
            raise IronError(
                "libiron_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (iron_amd has no CPU / eager fallback)" % path)
        lib = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.iron_version() != 1:
            raise IronError("libiron_hip.so ABI version mismatch")
        _lib = lib
        return lib


def check(status: int) -> None:
    if status != IRON_OK:
        lib = load()
        msg = lib.iron_strerror(status).decode()
        if status == -3:
            msg += " [hipError_t=%d]" % lib.iron_last_hip_error()
        raise IronError("libiron_hip: %s" % msg)


def profile_enable(on: bool) -> None:
    check(load().iron_profile_enable(1 if on else 0))


def profile_read() -> dict:
    """{kernel kind: (milliseconds, launches)} accumulated since the last read (blocks on the events)."""
    ms = (C.c_double * len(PROF_KINDS))()
    cnt = (C.c_int64 * len(PROF_KINDS))()
    check(load().iron_profile_read(ms, cnt))
    return {k: (ms[i], cnt[i]) for i, k in enumerate(PROF_KINDS)}


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    """The product path is HIP-only: refuse CPU tensors instead of silently computing elsewhere."""
    if not t.is_cuda:
        raise IronError("%s must be a CUDA (ROCm) tensor: iron_amd has no CPU path" % name)
    if t.dtype != torch.float32:
        raise IronError("%s must be float32, got %s" % (name, t.dtype))
    return t.contiguous()


# ---- call-scoped workspaces ----------------------------------------------------------------------------------------------------
# Several entries take a caller-owned workspace whose contents mean nothing once the call's kernels have run (the tape of the
# reverse-mode get_all: 1 MiB per resident workgroup, the shading workspace, the single-call tracer's lists).  A fresh torch.empty
# per call goes through the caching allocator, which now and then answers a 256 MiB request with a hipFree / hipMalloc pair --
# a device synchronisation in the middle of a frame (C2 ran 9.8 or 17.6 ms per batch depending on it).  One growing buffer per
# (device, stream, purpose) instead: calls on one stream are ordered, so the next call may overwrite it.
_workspaces = {}


def workspace(nbytes: int, device, tag: str):
    """A uint8 CUDA buffer of at least `nbytes` (>= 16) for the CURRENT stream of `device`, valid until the next workspace() call
    with the same tag on that stream."""
    import torch
    nbytes = max(int(nbytes), 16)
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _workspaces.pop(key, None)
        buf = torch.empty(nbytes + (nbytes >> 3), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
