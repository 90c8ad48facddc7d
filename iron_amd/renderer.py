"""Stage-1 NeuS volume renderer on the HIP kernels -- mirror of `models/renderer.py:128-453`.

`NeuSRenderer.render` keeps the reference's signature and output dictionary.  Every array stage runs as a HIP kernel
behind the C ABI: sample placement, hierarchical up-sampling (inverse CDF), sorted merges and compositing are the
per-ray kernels of csrc/neus.hip; the network evaluations between them are the batched SDF / colour / NeRF kernels the
field classes of `iron_amd.fields` already front.  Under grad mode with trainable networks the render cores are attached
to the parameters (iron_amd.autograd), so the reference's stage-1 loss.backward() works.
There is no CPU path -- tensors must live on the GPU and the HIP library must load.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib


def extract_fields(bound_min, bound_max, resolution, query_func, max_points: int = 1 << 22) -> np.ndarray:
    """models/renderer.py:9-31: query_func on the resolution^3 lattice between the bounds -> float32 [res, res, res] (numpy).
    The lattice is generated on the GPU in x-slabs of at most `max_points` points (the reference walks 64^3 blocks)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    axes = [torch.linspace(float(bound_min[i]), float(bound_max[i]), resolution).to(dev) for i in range(3)]
    u = np.zeros([resolution, resolution, resolution], dtype=np.float32)
    slab = max(1, min(resolution, max_points // (resolution * resolution)))
    lib = _lib.load()
    with torch.no_grad(), torch.cuda.device(dev):
        for x0 in range(0, resolution, slab):
            nx = min(slab, resolution - x0)
            pts = torch.empty((nx * resolution * resolution, 3), dtype=torch.float32, device=dev)
            _lib.check(lib.iron_grid_points(axes[0][x0:x0 + nx].data_ptr(), axes[1].data_ptr(), axes[2].data_ptr(), nx, resolution,
                                            resolution, pts.data_ptr(), _lib.stream_ptr(dev)))
            u[x0:x0 + nx] = query_func(pts).reshape(nx, resolution, resolution).detach().cpu().numpy()
    return u


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func):
    """models/renderer.py:34-42.  Marching cubes is the third-party `mcubes` there too; it is not part of this build."""
    try:
        import mcubes
    except ImportError as e:  # same dependency as the reference
        raise ImportError("extract_geometry needs PyMCubes (`mcubes`), as in models/renderer.py:36") from e
    u = extract_fields(bound_min, bound_max, resolution, query_func)
    vertices, triangles = mcubes.marching_cubes(u, threshold)
    b_max_np = torch.as_tensor(bound_max).detach().cpu().numpy()
    b_min_np = torch.as_tensor(bound_min).detach().cpu().numpy()
    vertices = vertices / (resolution - 1.0) * (b_max_np - b_min_np)[None, :] + b_min_np[None, :]
    return vertices, triangles


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    return _lib.require_cuda_f32(t.detach(), name).contiguous()


def sample_pdf(bins, weights, n_samples, det=False):
    """models/renderer.py:45-75: n_samples depths per row drawn from the piecewise-constant density `weights` (+1e-5) over
    `bins` by inverting its CDF -- at the regular positions linspace(0.5/n, 1-0.5/n) when det, else at uniform random numbers
    (torch's generator of the device; the reference draws them on its default device).  bins [n,m], weights [n,m-1]."""
    b, w = _f32(bins, "bins"), _f32(weights, "weights")
    if b.dim() != 2 or w.shape != (b.shape[0], b.shape[1] - 1):
        raise _lib.IronError("sample_pdf expects bins [n,m] and weights [n,m-1]")
    n = b.shape[0]
    u = None if det else torch.rand((n, int(n_samples)), dtype=torch.float32, device=b.device)
    out = torch.empty((n, int(n_samples)), dtype=torch.float32, device=b.device)
    with torch.cuda.device(b.device):
        _lib.check(_lib.load().iron_neus_sample_pdf(b.data_ptr(), w.data_ptr(), _lib.ptr(u), n, b.shape[1], int(n_samples), out.data_ptr(),
                                                    _lib.stream_ptr(b.device)))
    return out


class NeRFRenderer:
    """models/renderer.py:78-126: constructor and attributes as in the reference.  Its render() unpacks `sampled_color,
    density = self.nerf(pts, dirs)` -- the return order of tcnn_fields.TCNNNeRF (tiny-cuda-nn, a third-party CUDA library
    that is neither vendored in the reference nor installable here); with models.fields.NeRF, which returns (alpha, rgb), that
    line mis-shapes in the reference as well.  The only caller is render_volume_tcnn.py.  Out of this build's scope:
    render() says so instead of computing something unpinned."""

    def __init__(self, nerf, n_samples, n_importance, n_outside, up_sample_steps, perturb):
        self.nerf = nerf
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.up_sample_steps = up_sample_steps
        self.perturb = perturb

    def render(self, rays_o, rays_d, near, far, background_dist=0, sample_dist=0.01, background_rgb=None, cos_anneal_ratio=None):
        raise _lib.IronError("NeRFRenderer.render belongs to the tiny-cuda-nn experiment of render_volume_tcnn.py (its nerf must return "
                             "(colour, density) like tcnn_fields.TCNNNeRF); not part of the sphere-trace / NeuS path built here -- use "
                             "NeuSRenderer.render, whose background pass is this field evaluated by iron_nerf_forward")


class NeuSRenderer:
    """models/renderer.py:128-149 (same constructor arguments, same attribute names)."""

    def __init__(self, nerf, sdf_network, deviation_network, color_network, n_samples, n_importance, n_outside, up_sample_steps,
                 perturb):
        self.nerf = nerf
        self.sdf_network = sdf_network
        self.deviation_network = deviation_network
        self.color_network = color_network
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.up_sample_steps = up_sample_steps
        self.perturb = perturb
        self._consts = {}   # per device: the two constant sample rows
        self._inv_s = None  # (variance version, value): 1/s is a parameter, read back once per update, not per batch

    def _constants(self, dev):
        c = self._consts.get(dev)
        if c is None:
            lin = torch.linspace(0.0, 1.0, self.n_samples, device=dev)
            rev = None
            if self.n_outside > 0:
                rev = torch.flip(torch.linspace(1e-3, 1.0 - 1.0 / (self.n_outside + 1.0), self.n_outside, device=dev), dims=[-1]).contiguous()
            c = self._consts[dev] = (lin, rev)
        return c

    def invalidate(self) -> None:
        """Forget the cached 1/s (and the networks' packed copies): for parameter writes through `.data`, which torch's version
        counter does not see (iron_amd.fields._HipNet.invalidate)."""
        self._inv_s = None
        for net in (self.sdf_network, self.color_network, self.nerf):
            if net is not None and hasattr(net, "invalidate"):
                net.invalidate()

    def _inverse_s(self, dev) -> float:
        """deviation_network(zeros[1,3])[:, :1].clip(1e-6, 1e6) (renderer.py:283) as a host scalar, cached per parameter version."""
        var = getattr(self.deviation_network, "variance", None)
        key = None if var is None else (var.data_ptr(), var._version)
        if key is None or self._inv_s is None or self._inv_s[0] != key:
            val = float(self.deviation_network(torch.zeros([1, 3], device=dev))[0, 0].clip(1e-6, 1e6))
            self._inv_s = (key, val)
        return self._inv_s[1]

    # ---- per-ray kernels -----------------------------------------------------------------------------------------
    @staticmethod
    def _points(rays_o, rays_d, z):
        n, m = z.shape
        pts = torch.empty((n * m, 3), dtype=torch.float32, device=z.device)
        _lib.check(_lib.load().iron_neus_points(rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), n, m, pts.data_ptr(),
                                                _lib.stream_ptr(z.device)))
        return pts

    @staticmethod
    def _merge(z_a, s_a, z_b, s_b):
        n, ma = z_a.shape
        mb = z_b.shape[1]
        z = torch.empty((n, ma + mb), dtype=torch.float32, device=z_a.device)
        s = torch.empty_like(z) if s_a is not None else None
        _lib.check(_lib.load().iron_neus_merge(z_a.data_ptr(), _lib.ptr(s_a), ma, z_b.data_ptr(), _lib.ptr(s_b), mb, n, z.data_ptr(),
                                               _lib.ptr(s), _lib.stream_ptr(z.device)))
        return z, s

    @staticmethod
    def _mid_points(rays_o, rays_d, z, sample_dist, outside):
        n, m = z.shape
        dists = torch.empty_like(z)
        pts = torch.empty((n * m, 4 if outside else 3), dtype=torch.float32, device=z.device)
        dirs = torch.empty((n * m, 3), dtype=torch.float32, device=z.device)
        _lib.check(_lib.load().iron_neus_mid_points(rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), n, m, float(sample_dist),
                                                    int(outside), dists.data_ptr(), pts.data_ptr(), dirs.data_ptr(),
                                                    _lib.stream_ptr(z.device)))
        return dists, pts, dirs

    def up_sample(self, rays_o, rays_d, z_vals, sdf, n_importance, inv_s):
        """renderer.py:189-232: n_importance new depths per ray from the section weights at sharpness inv_s."""
        rays_o, rays_d, z_vals, sdf = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z_vals, "z_vals"), _f32(sdf, "sdf")
        n, m = z_vals.shape
        new_z = torch.empty((n, n_importance), dtype=torch.float32, device=z_vals.device)
        with torch.cuda.device(z_vals.device):
            _lib.check(_lib.load().iron_neus_up_sample(rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(), sdf.data_ptr(), n, m,
                                                       int(n_importance), float(inv_s), new_z.data_ptr(),
                                                       _lib.stream_ptr(z_vals.device)))
        return new_z

    def cat_z_vals(self, rays_o, rays_d, z_vals, new_z_vals, sdf, last=False):
        """renderer.py:234-248: merge the new depths in; unless `last`, evaluate and carry their sdf along."""
        rays_o, rays_d, z_vals, new_z_vals = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z_vals, "z_vals"), _f32(new_z_vals, "new_z")
        with torch.cuda.device(z_vals.device):
            if last:
                z, _ = self._merge(z_vals, None, new_z_vals, None)
                return z, sdf
            new_sdf = self.sdf_network.sdf(self._points(rays_o, rays_d, new_z_vals)).reshape(new_z_vals.shape)
            return self._merge(z_vals, _f32(sdf, "sdf"), new_z_vals, new_sdf)

    def extract_geometry(self, bound_min, bound_max, resolution, threshold=0.0):
        """renderer.py:455-462."""
        return extract_geometry(bound_min, bound_max, resolution=resolution, threshold=threshold,
                                query_func=lambda pts: -self.sdf_network.sdf(pts))

    # ---- the two cores under their own names (models/renderer.py:151-187, 250-344) ---------------------------------
    def _refuse_training(self, what, *nets):
        from .autograd import any_requires_grad
        params = [p for net in nets if net is not None for p in net.parameters()]
        if torch.is_grad_enabled() and any_requires_grad(*params):
            raise _lib.IronError("%s on its own is inference-only; NeuSRenderer.render() attaches both cores to the parameters "
                                 "(iron_amd.autograd) -- call it, or wrap this call in torch.no_grad()" % what)

    def render_core_outside(self, rays_o, rays_d, z_vals, sample_dist, nerf, background_rgb=None):
        """models/renderer.py:151-187: the NeRF++ background pass on its own -> color, sampled_color, alpha, weights."""
        self._refuse_training("render_core_outside", nerf)
        rays_o, rays_d, z_vals = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z_vals, "z_vals")
        n, mo = z_vals.shape
        dev = z_vals.device
        with torch.no_grad(), torch.cuda.device(dev):
            dists, pts, dirs = self._mid_points(rays_o, rays_d, z_vals, sample_dist, True)
            density, rgb = nerf(pts, dirs)
            density, rgb = density.reshape(-1).contiguous(), rgb.contiguous()
            alpha, weights = torch.empty_like(dists), torch.empty_like(dists)
            color = torch.empty((n, 3), dtype=torch.float32, device=dev)
            bg = None if background_rgb is None else _f32(background_rgb, "background_rgb").reshape(-1)
            _lib.check(_lib.load().iron_neus_outside_composite(density.data_ptr(), dists.data_ptr(), rgb.data_ptr(), _lib.ptr(bg), n, mo,
                                                               alpha.data_ptr(), weights.data_ptr(), color.data_ptr(), _lib.stream_ptr(dev)))
        return {"color": color, "sampled_color": rgb.reshape(n, mo, 3), "alpha": alpha, "weights": weights}

    def render_core(self, rays_o, rays_d, z_vals, sample_dist, sdf_network, deviation_network, color_network, background_alpha=None,
                    background_sampled_color=None, background_rgb=None, cos_anneal_ratio=0.0):
        """models/renderer.py:250-344 under its own name and signature (render() fuses it with the background pass):
        section mid points, get_all, colour network, logistic-CDF alpha, blend with the outside pass's alpha / colour,
        transmittance scan -> the reference's dict (color, sdf, dists, gradients, s_val, mid_z_vals, weights, cdf,
        gradient_error, inside_sphere)."""
        self._refuse_training("render_core", sdf_network, deviation_network, color_network)
        rays_o, rays_d, z_vals = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(z_vals, "z_vals")
        n, m = z_vals.shape
        dev = z_vals.device
        lib = _lib.load()
        with torch.no_grad(), torch.cuda.device(dev):
            dists, pts, dirs = self._mid_points(rays_o, rays_d, z_vals, sample_dist, False)
            sdf, feat, grad = sdf_network.get_all(pts, is_training=False)
            color_in = color_network(pts, grad, dirs, feat).contiguous()
            inv_s = float(deviation_network(torch.zeros([1, 3], device=dev))[0, 0].clip(1e-6, 1e6))
            mo = m
            bg_alpha = bg_color = None
            if background_alpha is not None:
                bg_alpha = _f32(background_alpha, "background_alpha")
                mo = bg_alpha.shape[1]
                bg_color = _f32(background_sampled_color, "background_sampled_color").reshape(n * mo, 3)
            a = _lib.iron_neus_composite_args()
            sdf_c, grad_c = sdf.reshape(-1).contiguous(), grad.contiguous()
            weights = torch.empty((n, mo), dtype=torch.float32, device=dev)
            cdf, inside = torch.empty((n, m), dtype=torch.float32, device=dev), torch.empty((n, m), dtype=torch.float32, device=dev)
            out_color = torch.empty((n, 3), dtype=torch.float32, device=dev)
            wsum, wmax = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
            gerr = torch.zeros(2, dtype=torch.float32, device=dev)
            bg = None if background_rgb is None else _f32(background_rgb, "background_rgb").reshape(-1)
            a.dists, a.pts, a.dirs, a.sdf, a.grad, a.color = (dists.data_ptr(), pts.data_ptr(), dirs.data_ptr(), sdf_c.data_ptr(),
                                                              grad_c.data_ptr(), color_in.data_ptr())
            a.bg_dists = a.bg_density = None
            a.bg_color = _lib.ptr(bg_color)
            a.background_rgb = _lib.ptr(bg)
            a.n, a.m, a.mo, a.inv_s, a.cos_anneal_ratio = n, m, mo, inv_s, float(cos_anneal_ratio)
            a.out_color, a.weights, a.cdf, a.inside_sphere = out_color.data_ptr(), weights.data_ptr(), cdf.data_ptr(), inside.data_ptr()
            a.weight_sum, a.weight_max, a.gradient_error_acc = wsum.data_ptr(), wmax.data_ptr(), gerr.data_ptr()
            if bg_alpha is not None:
                _lib.check(lib.iron_neus_composite_alpha(C.byref(a), bg_alpha.data_ptr(), _lib.stream_ptr(dev)))
            else:
                _lib.check(lib.iron_neus_composite(C.byref(a), _lib.stream_ptr(dev)))
            mid_z = z_vals + dists * 0.5
        return {"color": out_color, "sdf": sdf.reshape(-1, 1), "dists": dists, "gradients": grad.reshape(n, m, 3),
                "s_val": torch.full((n * m, 1), 1.0 / inv_s, dtype=torch.float32, device=dev), "mid_z_vals": mid_z, "weights": weights,
                "cdf": cdf, "gradient_error": gerr[0] / (gerr[1] + 1e-5), "inside_sphere": inside}

    # ---- render ---------------------------------------------------------------------------------------------------
    def _trainable(self) -> bool:
        from .autograd import any_requires_grad
        params = []
        for net in (self.sdf_network, self.color_network, self.nerf, self.deviation_network):
            if net is not None:
                params += list(net.parameters())
        return any_requires_grad(*params)

    def render(self, rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0) -> Dict[str, torch.Tensor]:
        """renderer.py:346-453.  The sample placement (linspace, optional stratified jitter, the four up-sampling rounds, merges)
        runs without grad exactly as in the reference (:387); under grad mode with trainable networks the two render cores are
        attached to the parameters -- get_all / colour net / NeRF / compositing are the differentiable operators of
        iron_amd.autograd -- so render_volume.py's loss.backward() works; otherwise plain tensors come back."""
        from .autograd import NeusCompositeFn
        training = self._trainable()
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        rays_o, rays_d = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d")
        dev = rays_o.device
        batch = rays_o.shape[0]
        near = _f32(near, "near").reshape(-1).expand(batch).contiguous()
        far = _f32(far, "far").reshape(-1).expand(batch).contiguous()
        lib = _lib.load()
        sample_dist = 2.0 / self.n_samples
        with torch.cuda.device(dev):
            with torch.no_grad():
                st = _lib.stream_ptr(dev)
                lin, rev = self._constants(dev)
                z_vals = torch.empty((batch, self.n_samples), dtype=torch.float32, device=dev)
                _lib.check(lib.iron_neus_linspace(near.data_ptr(), far.data_ptr(), lin.data_ptr(), batch, self.n_samples, z_vals.data_ptr(), st))
                z_out = None
                if perturb > 0:  # :369-378 (random numbers come from torch's generator of this device)
                    z_vals = z_vals + (torch.rand([batch, 1], device=dev) - 0.5) * 2.0 / self.n_samples
                if self.n_outside > 0:
                    z_out = torch.empty((batch, self.n_outside), dtype=torch.float32, device=dev)
                    if perturb > 0:
                        zo = torch.linspace(1e-3, 1.0 - 1.0 / (self.n_outside + 1.0), self.n_outside, device=dev)
                        mids = 0.5 * (zo[1:] + zo[:-1])
                        upper, lower = torch.cat([mids, zo[-1:]], -1), torch.cat([zo[:1], mids], -1)
                        zo = lower[None, :] + (upper - lower)[None, :] * torch.rand([batch, self.n_outside], device=dev)
                        z_out = far[:, None] / torch.flip(zo, dims=[-1]) + 1.0 / self.n_samples
                    else:
                        # far / flip(linspace(1e-3, 1 - 1/(n_outside+1))) + 1/n_samples: one ascending row per ray (:361-381)
                        _lib.check(lib.iron_neus_outside_z(far.data_ptr(), rev.data_ptr(), batch, self.n_outside, 1.0 / self.n_samples,
                                                           z_out.data_ptr(), st))
                n_samples = self.n_samples
                if self.n_importance > 0:
                    sdf = self.sdf_network.sdf(self._points(rays_o, rays_d, z_vals)).reshape(batch, self.n_samples)
                    for i in range(self.up_sample_steps):
                        new_z = self.up_sample(rays_o, rays_d, z_vals, sdf, self.n_importance // self.up_sample_steps, 64 * 2 ** i)
                        z_vals, sdf = self.cat_z_vals(rays_o, rays_d, z_vals, new_z, sdf, last=(i + 1 == self.up_sample_steps))
                    n_samples = self.n_samples + self.n_importance
                bg_dists = bg_pts = bg_dirs = None
                if self.n_outside > 0:
                    z_feed, _ = self._merge(z_vals, None, z_out.contiguous(), None)
                    bg_dists, bg_pts, bg_dirs = self._mid_points(rays_o, rays_d, z_feed, sample_dist, True)
                dists, pts, dirs = self._mid_points(rays_o, rays_d, z_vals, sample_dist, False)

            # render_core_outside (:151-187) and render_core (:250-344): attached to the parameters when training
            with torch.set_grad_enabled(training):
                density = bg_color = None
                inside_idx = None
                if self.n_outside > 0:
                    # The blend (:300-312) multiplies the background by (1 - inside_sphere): the field is evaluated only where that is
                    # not zero -- the n_outside far samples and the few inside samples beyond the unit sphere -- and the rest of the
                    # [n, n_samples + n_outside] row stays zero (the reference evaluates all of it and multiplies by 0).
                    mo = bg_dists.shape[1]
                    need = torch.empty((batch, mo), dtype=torch.uint8, device=dev)
                    _lib.check(lib.iron_neus_need_background(pts.data_ptr(), batch, n_samples, mo, need.data_ptr(), _lib.stream_ptr(dev)))
                    idx = need.reshape(-1).nonzero(as_tuple=False).reshape(-1)
                    d_sel, c_sel = self.nerf(bg_pts.index_select(0, idx), bg_dirs.index_select(0, idx))
                    density = torch.zeros((batch * mo, 1), dtype=torch.float32, device=dev).index_copy(0, idx, d_sel)
                    bg_color = torch.zeros((batch * mo, 3), dtype=torch.float32, device=dev).index_copy(0, idx, c_sel)
                    # ... and the colour network's output is multiplied by inside_sphere in the same blend: it runs on the inside samples
                    inside_idx = (need[:, :n_samples] == 0).reshape(-1).nonzero(as_tuple=False).reshape(-1)
                    if inside_idx.numel() == batch * n_samples:
                        inside_idx = None
                sdf, feat, grad = self.sdf_network.get_all(pts, is_training=training)
                if inside_idx is None:
                    color = self.color_network(pts, grad, dirs, feat)
                else:
                    c_in = self.color_network(pts.index_select(0, inside_idx), grad.index_select(0, inside_idx), dirs.index_select(0, inside_idx),
                                              feat.index_select(0, inside_idx))
                    color = torch.zeros((batch * n_samples, 3), dtype=torch.float32, device=dev).index_copy(0, inside_idx, c_in)
                if training:
                    inv_s = self.deviation_network(torch.zeros([1, 3], device=dev))[0, 0].clip(1e-6, 1e6)
                    s_val = (1.0 / inv_s).reshape(1, 1).expand(batch, 1)
                else:
                    inv_s = self._inverse_s(dev)
                    s_val = torch.full((batch, 1), 1.0 / inv_s, dtype=torch.float32, device=dev)
                out_color, weights, wsum, gradient_error, cdf, inside, wmax = NeusCompositeFn.apply(
                    sdf, grad, color, inv_s, density, bg_color, dists, pts, dirs, bg_dists, background_rgb, float(cos_anneal_ratio))
        return {
            "color_fine": out_color,
            "s_val": s_val,
            "cdf_fine": cdf,
            "weight_sum": wsum,
            "weight_max": wmax,
            "gradients": grad.reshape(batch, n_samples, 3),
            "weights": weights,
            "gradient_error": gradient_error,
            "inside_sphere": inside,
        }
