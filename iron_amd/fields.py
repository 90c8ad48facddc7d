"""SDFNetwork / RenderingNetwork with the reference's constructor and method surface
(models/fields.py:9-137, 141-239), forward passes routed to the gfx950 HIP kernels.

Parameters live in torch exactly as in the reference (`lin{l}.weight_g [out,1]`, `lin{l}.weight_v
[out,in]`, `lin{l}.bias`; old-style weight_norm, fields.py:75-76), so reference checkpoints load
with `load_state_dict`.  The HIP side keeps a packed copy (weight norm folded, MFMA fragment order)
that is rebuilt whenever a parameter's version counter changes.

There is no CPU / eager compute path here: tensors must be CUDA (ROCm) fp32.
SDFNetwork.get_all(is_training=True) / .gradient and RenderingNetwork.forward are differentiable through
iron_amd.autograd (HIP forward + closed-form HIP backward); NeRF is forward only.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .embedder import get_embedder


class _Linear(nn.Module):
    """One (optionally weight-normed) linear layer's parameters, named like the reference's."""

    def __init__(self, lin: nn.Linear, weight_norm: bool):
        super().__init__()
        self.in_features, self.out_features = lin.in_features, lin.out_features
        self.has_weight_norm = weight_norm
        w = lin.weight.data
        if weight_norm:
            # nn.utils.weight_norm(dim=0): g = ||v||_row [out,1], v = w
            self.weight_g = nn.Parameter(torch.norm_except_dim(w, 2, 0).data)
            self.weight_v = nn.Parameter(w.clone())
        else:
            self.weight = nn.Parameter(w.clone())
        self.bias = nn.Parameter(lin.bias.data.clone())

    def effective_weight(self) -> torch.Tensor:
        if self.has_weight_norm:
            return torch._weight_norm(self.weight_v, self.weight_g, 0)
        return self.weight


class _NetHandle:
    """Owns an iron_net_t*."""

    def __init__(self, handle: int, device: torch.device):
        self.handle = handle
        self.device = device

    def __del__(self):
        try:
            if self.handle:
                _lib.load().iron_net_destroy(self.handle)
        except Exception:
            pass
        self.handle = 0


class _HipNet(nn.Module):
    """Shared packing/caching logic."""

    _kind = _lib.IRON_NET_SDF

    def _layers(self):
        return [getattr(self, "lin%d" % l) for l in range(self.num_layers - 1)]

    def _desc(self) -> _lib.iron_net_desc:  # pragma: no cover - overridden
        raise NotImplementedError

    def _param_key(self):
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters())

    def invalidate(self) -> None:
        """Drop the packed device copy: the next kernel call re-packs from the parameters as they are now.  hip_net() notices
        assignments and in-place ops on the parameters themselves (optimizer steps, load_state_dict, p.add_()): they bump
        torch's version counter, which is part of the cache key.  Writes THROUGH `p.data` (p.data.copy_(), EMA or clipping code
        that goes around autograd) bump nothing: call invalidate() (alias: repack()) after those."""
        self.__dict__.pop("_hip_cache", None)

    repack = invalidate

    def hip_net(self) -> _NetHandle:
        """The packed device copy of the current parameters (rebuilt if they changed; see invalidate() for the one case
        torch gives no sign of)."""
        key = self._param_key()
        cached = self.__dict__.get("_hip_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        layers = self._layers()
        dev = layers[0].bias.device
        if dev.type != "cuda":
            raise _lib.IronError("network parameters must live on a CUDA (ROCm) device; call .cuda() first")
        lib = _lib.load()
        arr = (_lib.iron_linear * len(layers))()
        keep = []
        for i, lin in enumerate(layers):
            if getattr(lin, "has_weight_norm", False):
                v = lin.weight_v.detach().float().contiguous()
                g = lin.weight_g.detach().float().contiguous().view(-1)
                keep += [v, g]
                arr[i].weight_v, arr[i].weight_g = v.data_ptr(), g.data_ptr()
            else:
                v = lin.weight.detach().float().contiguous()
                keep.append(v)
                arr[i].weight_v, arr[i].weight_g = v.data_ptr(), None
            b = lin.bias.detach().float().contiguous()
            keep.append(b)
            arr[i].bias = b.data_ptr()
            arr[i].out_dim, arr[i].in_dim = lin.out_features, lin.in_features
        desc = self._desc()
        out = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.iron_net_create(C.byref(out), C.byref(desc), arr, _lib.stream_ptr(dev)))
        handle = _NetHandle(out.value, dev)
        self.__dict__["_hip_cache"] = (key, handle)
        if self.__dict__.get("_force_exact"):
            _lib.check(lib.iron_net_force_exact(handle.handle, 1))
        return handle

    # ---- numeric envelope of the default (split-fp16, "h2") core: include/iron_hip.h, csrc/envelope.hip -----------------------
    def numeric_status(self) -> dict:
        """Synchronises the current stream and reports the packed network's envelope status: `overflow_seen` (some call on the h2
        core returned non-finite values because an activation or feature left fp16's range), `exact_core` (the network runs on the
        exact-fp32 MFMA core: after an overflow, or force_exact()), `pending` (the call that just finished overflowed; the next
        one will run on the exact core)."""
        h = self.hip_net()
        st = C.c_int32(0)
        with torch.cuda.device(h.device):
            _lib.check(_lib.load().iron_net_numeric_status(h.handle, C.byref(st), _lib.stream_ptr(h.device)))
        return {"overflow_seen": bool(st.value & 1), "exact_core": bool(st.value & 2), "pending": bool(st.value & 4)}

    def force_exact(self, on: bool = True) -> None:
        """Pin this network to the exact-fp32 MFMA core (survives re-packing), or return it to the default core."""
        self.__dict__["_force_exact"] = bool(on)
        _lib.check(_lib.load().iron_net_force_exact(self.hip_net().handle, 1 if on else 0))

    def _rerun_if_overflowed(self, call):
        """IRON_H2_OVERFLOW=rerun: pay one synchronisation per call, and when the call left the h2 core's range run it again (the
        library has moved the network to the exact-fp32 core by then).  Default: no synchronisation -- the offending call returns
        non-finite values and every later call is exact (numeric_status() tells)."""
        out = call()
        if _OVERFLOW_RERUN and self.numeric_status()["pending"]:
            out = call()
        return out


_OVERFLOW_RERUN = os.environ.get("IRON_H2_OVERFLOW", "") == "rerun"


# IDR-style SDF MLP (reference: models/fields.py:9-137)
class SDFNetwork(_HipNet):
    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, bias=0.5, scale=1,
                 geometric_init=True, weight_norm=True, inside_outside=False):
        super().__init__()
        dims = [d_in] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.embed_fn_fine = None
        self.multires = multires
        if multires > 0:
            embed_fn, input_ch = get_embedder(multires, input_dims=d_in)
            self.embed_fn_fine = embed_fn
            dims[0] = input_ch
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        self.d_in, self.d_out, self.d_hidden = d_in, d_out, d_hidden

        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)  # default init first: keeps the RNG stream of the reference
            if geometric_init:  # SAL/IDR geometric initialisation (fields.py:47-73): a sphere of radius `bias`
                if l == self.num_layers - 2:
                    mean = np.sqrt(np.pi) / np.sqrt(dims[l])
                    if not inside_outside:
                        torch.nn.init.normal_(lin.weight, mean=mean, std=0.0001)
                        torch.nn.init.constant_(lin.bias, -bias)
                    else:
                        torch.nn.init.normal_(lin.weight, mean=-mean, std=0.0001)
                        torch.nn.init.constant_(lin.bias, bias)
                elif multires > 0 and l == 0:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                    torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif multires > 0 and l in self.skip_in:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            setattr(self, "lin" + str(l), _Linear(lin, weight_norm))

    def _desc(self) -> _lib.iron_net_desc:
        d = _lib.iron_net_desc()
        d.kind = _lib.IRON_NET_SDF
        d.n_linear = self.num_layers - 1
        d.d_hidden = self.d_hidden
        d.d_out = self.d_out
        d.multires = self.multires
        d.multires_view = 0
        if len(self.skip_in) > 1 or self.d_in != 3:
            raise _lib.IronError("unsupported SDFNetwork shape for the gfx950 kernels")
        d.skip_layer = self.skip_in[0] if self.skip_in else -1
        d.scale = float(self.scale)
        return d

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        """[..., 3] -> [..., d_out]  (fields.py:82-98).  Under grad mode with trainable parameters the result is attached to them
        (through the get_all operator and its backward); under torch.no_grad() -- the tracer, validation -- it is the plain kernel."""
        if self._attached():
            sdf, feat, _ = self.get_all(inputs, is_training=True)
            return torch.cat([sdf, feat], dim=-1)
        return self._run(inputs, self.d_out)

    def _attached(self) -> bool:
        from .autograd import any_requires_grad
        return any_requires_grad(*self.parameters())

    def _run(self, inputs: torch.Tensor, out_cols: int) -> torch.Tensor:
        x = _lib.require_cuda_f32(inputs.detach(), "inputs")
        sh = list(x.shape[:-1])
        x = x.reshape(-1, 3)
        net = self.hip_net()
        out = torch.empty((x.shape[0], out_cols), dtype=torch.float32, device=x.device)

        def call():
            with torch.cuda.device(x.device):
                _lib.check(_lib.load().iron_sdf_forward(net.handle, x.data_ptr(), x.shape[0], out.data_ptr(), out_cols,
                                                        _lib.stream_ptr(x.device)))
        self._rerun_if_overflowed(call)
        return out.reshape(sh + [out_cols])

    def sdf(self, x: torch.Tensor) -> torch.Tensor:
        """fields.py:100-101: [..., 1]; uses the sdf-only kernel (no 256-wide feature epilogue) unless the result has to be
        attached to trainable parameters (grad mode)."""
        if self._attached():
            return self.get_all(x, is_training=True)[0]
        return self._run(x, 1)

    def sdf_hidden_appearance(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward(x)

    def gradient(self, x: torch.Tensor) -> torch.Tensor:
        """d sdf / dx (fields.py:106-118).  Under grad mode with trainable parameters the result is attached to them (the
        reference's create_graph=True; this is what its eikonal loss differentiates), otherwise values only."""
        from .autograd import any_requires_grad
        return self.get_all(x, is_training=any_requires_grad(*self.parameters()))[2]

    def get_all(self, x: torch.Tensor, is_training: bool = True):
        """sdf [...,1], feature [...,d_out-1], gradient [...,3]  (fields.py:120-137)."""
        if is_training:
            # attached to the parameters (to second order through the gradient): HIP forward + iron_sdf_backward
            from .autograd import SDFGetAllFn, _layer_params
            if x.grad_fn is not None and torch.is_grad_enabled():
                # the reference propagates through x under create_graph (fields.py:127-134); the closed-form backward here returns
                # parameter gradients only.  A LEAF x with requires_grad=True stays allowed: the reference sets that flag in place.
                raise NotImplementedError("SDFNetwork.get_all(is_training=True): x is the result of a differentiable computation and "
                                          "would silently lose its gradient; pass x.detach() (the stage-2 / NeuS callers evaluate at "
                                          "detached points) or differentiate w.r.t. x with a torch reference of the network")
            sh = list(x.shape[:-1])
            sdf, feat, grad = SDFGetAllFn.apply(self, x.detach().reshape(-1, 3), *_layer_params(self))
            return sdf.reshape(sh + [1]), feat.reshape(sh + [self.d_out - 1]), grad.reshape(sh + [3])
        xx = _lib.require_cuda_f32(x.detach(), "x")
        sh = list(xx.shape[:-1])
        xx = xx.reshape(-1, 3)
        n = xx.shape[0]
        net = self.hip_net()
        sdf = torch.empty((n, 1), dtype=torch.float32, device=xx.device)
        feat = torch.empty((n, self.d_out - 1), dtype=torch.float32, device=xx.device)
        grad = torch.empty((n, 3), dtype=torch.float32, device=xx.device)
        def call():
            with torch.cuda.device(xx.device):
                ws, ws_bytes = _get_all_workspace(net, n, xx.device)
                _lib.check(_lib.load().iron_sdf_get_all(net.handle, xx.data_ptr(), n, sdf.data_ptr(), feat.data_ptr(),
                                                        grad.data_ptr(), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(xx.device)))
        self._rerun_if_overflowed(call)
        return sdf.reshape(sh + [1]), feat.reshape(sh + [self.d_out - 1]), grad.reshape(sh + [3])

    @torch.no_grad()
    def get_sdf_and_gradient(self, x: torch.Tensor):
        """get_all without the feature rows (the edge walk, raytracer.py:453 / :683, discards them):
        sdf [...,1], gradient [...,3]; skips the 256x256 feature layer."""
        xx = _lib.require_cuda_f32(x.detach(), "x")
        sh = list(xx.shape[:-1])
        xx = xx.reshape(-1, 3)
        n = xx.shape[0]
        sdf = torch.empty((n, 1), dtype=torch.float32, device=xx.device)
        grad = torch.empty((n, 3), dtype=torch.float32, device=xx.device)
        with torch.cuda.device(xx.device):
            net = self.hip_net()
            ws, ws_bytes = _get_all_workspace(net, n, xx.device)
            _lib.check(_lib.load().iron_sdf_get_all(net.handle, xx.data_ptr(), n, sdf.data_ptr(), None,
                                                    grad.data_ptr(), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(xx.device)))
        return sdf.reshape(sh + [1]), grad.reshape(sh + [3])


def _get_all_workspace(net, n, device):
    """The tape of the reverse-mode get_all kernel (include/iron_hip.h: iron_sdf_get_all): (tensor or None, bytes)."""
    nbytes = int(_lib.load().iron_sdf_get_all_workspace_bytes(net.handle, n))
    if nbytes == 0:
        return None, 0
    return _lib.workspace(nbytes, device, "get_all"), nbytes


# IDR-style material MLP (reference: models/fields.py:141-239)
class RenderingNetwork(_HipNet):
    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires=0,
                 multires_view=0, squeeze_out=True, squeeze_out_scale=1.0, output_bias=0.0, output_scale=1.0,
                 skip_in=()):
        super().__init__()
        self.mode = mode
        self.squeeze_out = squeeze_out
        self.d_feature, self.d_out, self.d_hidden = d_feature, d_out, d_hidden
        self.multires, self.multires_view = multires, multires_view
        dims = [d_in + d_feature] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.embed_fn = None
        if multires > 0:
            self.embed_fn, input_ch = get_embedder(multires)
            dims[0] += input_ch - 3
        self.embedview_fn = None
        if multires_view > 0:
            self.embedview_fn, input_ch = get_embedder(multires_view)
            dims[0] += input_ch - 3
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        for l in range(self.num_layers - 1):
            if l in self.skip_in:
                dims[l] += dims[0]
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            setattr(self, "lin" + str(l), _Linear(nn.Linear(dims[l], out_dim), weight_norm))
        self.output_bias = output_bias
        self.output_scale = output_scale
        self.squeeze_out_scale = squeeze_out_scale

    def _desc(self) -> _lib.iron_net_desc:
        d = _lib.iron_net_desc()
        d.kind = _lib.IRON_NET_RENDER
        d.n_linear = self.num_layers - 1
        d.d_hidden = self.d_hidden
        d.d_out = self.d_out
        d.multires = self.multires
        d.multires_view = self.multires_view
        d.skip_layer = self.skip_in[0] if self.skip_in else -1
        if len(self.skip_in) > 1:
            raise _lib.IronError("unsupported RenderingNetwork shape for the gfx950 kernels")
        d.mode = _lib.MODES[self.mode]
        d.d_feature = self.d_feature
        d.squeeze_out = 1 if self.squeeze_out else 0
        d.squeeze_out_scale = float(self.squeeze_out_scale)
        d.output_bias = float(self.output_bias)
        d.output_scale = float(self.output_scale)
        d.scale = 1.0
        return d

    def forward(self, points, normals, view_dirs, feature_vectors) -> torch.Tensor:
        """[...,3] x3 (+ [...,d_feature]) -> [..., d_out]  (fields.py:203-239).  Differentiable (inputs and parameters) when
        called under grad mode with anything that requires grad: HIP forward + iron_render_backward."""
        from .autograd import RenderNetFn, _layer_params, any_requires_grad
        params = _layer_params(self)
        if any_requires_grad(points, normals, view_dirs, feature_vectors, *params):
            return RenderNetFn.apply(self, points, normals, view_dirs, feature_vectors, *params)
        return self._forward_values(points, normals, view_dirs, feature_vectors)

    def _forward_values(self, points, normals, view_dirs, feature_vectors) -> torch.Tensor:
        p = _lib.require_cuda_f32(points.detach(), "points")
        sh = list(p.shape[:-1])
        p = p.reshape(-1, 3)
        n = p.shape[0]
        nrm = _lib.require_cuda_f32(normals.detach(), "normals").reshape(-1, 3) if normals is not None else None
        vd = _lib.require_cuda_f32(view_dirs.detach(), "view_dirs").reshape(-1, 3) if view_dirs is not None else None
        ft = _lib.require_cuda_f32(feature_vectors.detach(), "feature_vectors").reshape(-1, self.d_feature)
        net = self.hip_net()
        out = torch.empty((n, self.d_out), dtype=torch.float32, device=p.device)
        def call():
            with torch.cuda.device(p.device):
                _lib.check(_lib.load().iron_render_forward(net.handle, p.data_ptr(), _lib.ptr(nrm), _lib.ptr(vd),
                                                           ft.data_ptr(), n, out.data_ptr(), _lib.stream_ptr(p.device)))
        self._rerun_if_overflowed(call)
        return out.reshape(sh + [self.d_out])


# NeRF background field of the stage-1 renderer (reference: models/fields.py:243-327)
class NeRF(_HipNet):
    """Same constructor, parameter names (pts_linears.N / views_linears.0 / feature_linear / alpha_linear / rgb_linear, plain
    nn.Linear) and RNG consumption as the reference; forward runs on the HIP kernel `k_nerf` (iron_nerf_forward)."""

    def __init__(self, D=8, W=256, d_in=3, d_in_view=3, multires=0, multires_view=0, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        from .embedder import get_embedder
        self.D, self.W, self.d_in, self.d_in_view = D, W, d_in, d_in_view
        self.multires, self.multires_view = multires, multires_view
        self.input_ch, self.input_ch_view = 3, 3
        self.embed_fn = self.embed_fn_view = None
        if multires > 0:
            self.embed_fn, self.input_ch = get_embedder(multires, input_dims=d_in)
        if multires_view > 0:
            self.embed_fn_view, self.input_ch_view = get_embedder(multires_view, input_dims=d_in_view)
        self.skips = list(skips)
        self.use_viewdirs = use_viewdirs
        self.pts_linears = nn.ModuleList(
            [nn.Linear(self.input_ch, W)]
            + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.input_ch, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(self.input_ch_view + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)

    def _layers(self):
        return list(self.pts_linears) + [self.alpha_linear, self.feature_linear, self.views_linears[0], self.rgb_linear]

    def _desc(self) -> _lib.iron_net_desc:
        if not self.use_viewdirs:
            raise NotImplementedError("NeRF(use_viewdirs=False): the reference's forward asserts False there (fields.py:326-327)")
        if self.d_in != 4 or self.d_in_view != 3 or len(self.skips) > 1:
            raise _lib.IronError("the HIP NeRF kernel is built for the stage-1 background field (d_in=4, d_in_view=3, one skip)")
        d = _lib.iron_net_desc()
        d.kind = _lib.IRON_NET_NERF
        d.n_linear = self.D + 4
        d.d_hidden = self.W
        d.d_out = 4
        d.multires, d.multires_view = self.multires, self.multires_view
        d.skip_layer = self.skips[0] if self.skips else -1
        d.scale = 1.0
        return d

    def forward(self, input_pts, input_views):
        """[...,4], [...,3] -> (alpha [...,1], rgb [...,3])  (fields.py:299-325).  Under grad mode the outputs are attached to the
        parameters (HIP forward + iron_nerf_backward); inputs that require grad are refused (no input gradient is built: the
        reference feeds positions computed without grad)."""
        from .autograd import NeRFFn, _layer_params, any_requires_grad, refuse_grad
        refuse_grad("NeRF.forward w.r.t. its inputs", input_pts, input_views)
        params = _layer_params(self)
        if any_requires_grad(*params):
            return NeRFFn.apply(self, input_pts, input_views, *params)
        return self._forward_values(input_pts, input_views)

    def _forward_values(self, input_pts, input_views):
        p = _lib.require_cuda_f32(input_pts.detach(), "input_pts")
        sh = list(p.shape[:-1])
        p = p.reshape(-1, 4)
        v = _lib.require_cuda_f32(input_views.detach(), "input_views").reshape(-1, 3)
        n = p.shape[0]
        alpha = torch.empty(n, dtype=torch.float32, device=p.device)
        rgb = torch.empty((n, 3), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(_lib.load().iron_nerf_forward(self.hip_net().handle, p.data_ptr(), v.data_ptr(), n, alpha.data_ptr(),
                                                     rgb.data_ptr(), _lib.stream_ptr(p.device)))
        return alpha.reshape(sh + [1]), rgb.reshape(sh + [3])


class SingleVarianceNetwork(nn.Module):
    """models/fields.py:415-421."""

    def __init__(self, init_val):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(init_val)))

    def forward(self, x):
        return torch.ones([len(x), 1], device=self.variance.device) * torch.exp(self.variance * 10.0)
