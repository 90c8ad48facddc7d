"""torch.autograd.Function wrappers that make the HIP operators trainable (BASELINE config C3).

The reference trains by running its ordinary torch modules under autograd (render_surface.py:533-653).  Here each
operator's forward is the inference HIP kernel (libiron_hip.so) and its backward is the closed-form HIP pass (hand-written split-fp16 MFMA GEMMs, no BLAS) of
libiron_train.so (include/iron_train.h):

    SDFGetAllFn      SDFNetwork.get_all(is_training=True) / .gradient       models/fields.py:106-137 (second order)
    RenderNetFn      RenderingNetwork.forward                               models/fields.py:203-239
    GGXColocatedFn   GGXColocatedRenderer.forward                           models/renderer_ggx.py:82-146

so `render_camera(..., is_training=True)`, the reference's `reparam_points`, its `render_fn` closure and its loss code
run unchanged on top and `loss.backward()` fills `.grad` of weight_g / weight_v / bias / light exactly as there.  There is
no fallback: a missing library or a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
from torch.autograd.function import once_differentiable

from . import _lib


def _layer_params(net) -> List[torch.Tensor]:
    """Flat parameter list in the order the Functions take / return them: per layer (weight_g, weight_v, bias) or
    (weight, bias)."""
    out = []
    for lin in net._layers():
        if getattr(lin, "has_weight_norm", False):
            out += [lin.weight_g, lin.weight_v, lin.bias]
        else:
            out += [lin.weight, lin.bias]
    return out


def _train_layers(net, device):
    """iron_train_layer array over the module's current parameters + freshly allocated gradient buffers (returned in
    _layer_params order)."""
    layers = net._layers()
    arr = (_lib.iron_train_layer * len(layers))()
    keep, grads = [], []
    for i, lin in enumerate(layers):
        wn = getattr(lin, "has_weight_norm", False)
        v = (lin.weight_v if wn else lin.weight).detach()
        b = lin.bias.detach()
        v = _lib.require_cuda_f32(v, "weight")
        b = _lib.require_cuda_f32(b, "bias")
        dv, db = torch.empty_like(v), torch.empty_like(b)
        arr[i].weight_v, arr[i].bias, arr[i].d_weight_v, arr[i].d_bias = v.data_ptr(), b.data_ptr(), dv.data_ptr(), db.data_ptr()
        arr[i].out_dim, arr[i].in_dim = v.shape[0], v.shape[1]
        keep += [v, b]
        if wn:
            g = _lib.require_cuda_f32(lin.weight_g.detach(), "weight_g")
            dg = torch.empty_like(g)
            arr[i].weight_g, arr[i].d_weight_g = g.data_ptr(), dg.data_ptr()
            keep.append(g)
            grads += [dg, dv, db]
        else:
            arr[i].weight_g, arr[i].d_weight_g = None, None
            grads += [dv, db]
    return arr, keep, grads


def _opt(t: Optional[torch.Tensor], shape=None) -> Optional[torch.Tensor]:
    if t is None:
        return None
    t = _lib.require_cuda_f32(t, "upstream gradient")
    return t.reshape(shape) if shape is not None else t


class SDFGetAllFn(torch.autograd.Function):
    """(sdf [n,1], feature [n,d_out-1], gradient [n,3]) = get_all(x); differentiable w.r.t. the network parameters, to
    second order through `gradient` (the tangent sweep of iron_sdf_backward).  x itself gets no gradient: the reference
    evaluates get_all on the tracer's detached hit points (raytracer.py:622) and on sampled eikonal points."""

    @staticmethod
    def forward(ctx, net, x, *params):
        ctx.set_materialize_grads(False)
        ctx.net = net
        ctx.param_versions = tuple(p._version for p in params)  # backward re-reads the module's parameters: they must not have moved
        with torch.no_grad():
            sdf, feat, grad = net.get_all(x, is_training=False)
        ctx.save_for_backward(x.detach())
        return sdf, feat, grad

    @staticmethod
    @once_differentiable  # closed-form first-order backward: differentiating it again raises instead of returning zeros
    def backward(ctx, d_sdf, d_feat, d_grad):
        net = ctx.net
        (x,) = ctx.saved_tensors
        x = _lib.require_cuda_f32(x, "x").reshape(-1, 3)
        n = x.shape[0]
        dev = x.device
        if ctx.param_versions != tuple(p._version for p in _layer_params(net)):
            raise RuntimeError("a parameter of the SDFNetwork was modified in place between get_all(is_training=True) and backward(); "
                               "the closed-form backward re-evaluates the forward from the CURRENT parameters (as autograd's own version "
                               "check would, this refuses instead of returning a gradient of a different function)")
        if net.scale != 1 or len(net.skip_in) > 1:
            raise _lib.IronError("SDF backward supports scale = 1 and at most one skip layer")
        lib = _lib.load_train()
        with torch.cuda.device(dev):
            arr, keep, grads = _train_layers(net, dev)
            desc = _lib.iron_sdf_train_desc()
            desc.n_linear, desc.multires = net.num_layers - 1, net.multires
            desc.skip_layer = net.skip_in[0] if net.skip_in else -1
            desc.layers = arr
            d_sdf, d_feat, d_grad = _opt(d_sdf, (-1,)), _opt(d_feat, (n, -1)), _opt(d_grad, (-1, 3))
            nbytes = lib.iron_sdf_backward_workspace_bytes(C.byref(desc), n)
            if nbytes == 0:
                raise _lib.IronError("unsupported SDFNetwork shape for the backward pass")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check_train(lib.iron_sdf_backward(C.byref(desc), x.data_ptr(), n, _lib.ptr(d_sdf), _lib.ptr(d_feat), _lib.ptr(d_grad),
                                                   ws.data_ptr(), nbytes, _lib.stream_ptr(dev)))
        del keep
        return (None, None) + tuple(grads)


class RenderNetFn(torch.autograd.Function):
    """out = RenderingNetwork(points, normals, view_dirs, features); differentiable w.r.t. all four inputs and the
    parameters."""

    @staticmethod
    def forward(ctx, net, points, normals, view_dirs, feats, *params):
        ctx.net = net
        with torch.no_grad():
            out = net._forward_values(points, normals, view_dirs, feats)
        ctx.has = (normals is not None, view_dirs is not None)
        ctx.shapes = (points.shape, None if normals is None else normals.shape, None if view_dirs is None else view_dirs.shape, feats.shape)
        ctx.save_for_backward(points.detach(), feats.detach(), *([normals.detach()] if normals is not None else []),
                              *([view_dirs.detach()] if view_dirs is not None else []))
        return out

    @staticmethod
    @once_differentiable  # closed-form first-order backward: differentiating it again raises instead of returning zeros
    def backward(ctx, d_out):
        net = ctx.net
        saved = list(ctx.saved_tensors)
        pts, feats = saved[0], saved[1]
        k = 2
        nrm = vd = None
        if ctx.has[0]:
            nrm = saved[k]; k += 1
        if ctx.has[1]:
            vd = saved[k]
        pts = _lib.require_cuda_f32(pts, "points").reshape(-1, 3)
        n = pts.shape[0]
        dev = pts.device
        feats = _lib.require_cuda_f32(feats, "features").reshape(n, -1)
        nrm = _lib.require_cuda_f32(nrm, "normals").reshape(-1, 3) if nrm is not None else None
        vd = _lib.require_cuda_f32(vd, "view_dirs").reshape(-1, 3) if vd is not None else None
        lib = _lib.load_train()
        with torch.cuda.device(dev):
            arr, keep, grads = _train_layers(net, dev)
            desc = _lib.iron_render_train_desc()
            desc.n_linear, desc.mode = net.num_layers - 1, _lib.MODES[net.mode]
            desc.multires, desc.multires_view = net.multires, net.multires_view
            desc.d_feature, desc.d_out = net.d_feature, net.d_out
            if len(net.skip_in) > 1:
                raise _lib.IronError("RenderingNetwork backward supports at most one skip layer")
            desc.skip_layer = net.skip_in[0] if net.skip_in else -1
            desc.squeeze_out = 1 if net.squeeze_out else 0
            desc.output_bias, desc.output_scale, desc.squeeze_out_scale = float(net.output_bias), float(net.output_scale), float(net.squeeze_out_scale)
            desc.layers = arr
            g = _lib.require_cuda_f32(d_out, "d_out").reshape(n, net.d_out)
            need = ctx.needs_input_grad  # (net, points, normals, view_dirs, feats, *params)
            d_pts = torch.empty_like(pts) if need[1] else None
            d_nrm = torch.empty_like(nrm) if (nrm is not None and need[2]) else None
            d_vd = torch.empty_like(vd) if (vd is not None and need[3]) else None
            d_ft = torch.empty_like(feats) if need[4] else None
            nbytes = lib.iron_render_backward_workspace_bytes(C.byref(desc), n)
            if nbytes == 0:
                raise _lib.IronError("unsupported RenderingNetwork shape for the backward pass")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check_train(lib.iron_render_backward(C.byref(desc), pts.data_ptr(), _lib.ptr(nrm), _lib.ptr(vd), feats.data_ptr(), n, g.data_ptr(),
                                                      _lib.ptr(d_pts), _lib.ptr(d_nrm), _lib.ptr(d_vd), _lib.ptr(d_ft), ws.data_ptr(), nbytes,
                                                      _lib.stream_ptr(dev)))
        del keep
        sp, sn, sv, sf = ctx.shapes
        return (None,
                d_pts.reshape(sp) if d_pts is not None else None,
                d_nrm.reshape(sn) if d_nrm is not None else None,
                d_vd.reshape(sv) if d_vd is not None else None,
                d_ft.reshape(sf) if d_ft is not None else None) + tuple(grads)


class GGXColocatedFn(torch.autograd.Function):
    """(diffuse_rgb, specular_rgb, rgb) = GGXColocatedRenderer(light, distance, normal, viewdir, kd, ks, roughness)."""

    @staticmethod
    def forward(ctx, renderer, light, distance, normal, viewdir, kd, ks, rough):
        ctx.set_materialize_grads(False)
        ctx.renderer = renderer
        light_t = light if torch.is_tensor(light) else None
        ctx.light_shape = None if light_t is None else light_t.shape
        ctx.light_value = float(light)
        with torch.no_grad():
            out = renderer._forward_values(ctx.light_value, distance, normal, viewdir, kd, ks, rough)
        ctx.shapes = (distance.shape, normal.shape, viewdir.shape, kd.shape, ks.shape, rough.shape)
        ctx.save_for_backward(distance.detach(), normal.detach(), viewdir.detach(), kd.detach(), ks.detach(), rough.detach())
        return out["diffuse_rgb"], out["specular_rgb"], out["rgb"]

    @staticmethod
    @once_differentiable  # closed-form first-order backward: differentiating it again raises instead of returning zeros
    def backward(ctx, g_diff, g_spec, g_rgb):
        dist, nrm, vd, kd, ks, rough = ctx.saved_tensors
        nrm = _lib.require_cuda_f32(nrm, "normal").reshape(-1, 3)
        n = nrm.shape[0]
        dev = nrm.device
        sh = list(ctx.shapes[1][:-1])
        dist = _lib.require_cuda_f32(dist, "distance").reshape(-1)
        vd = _lib.require_cuda_f32(vd, "viewdir").reshape(-1, 3)
        kd = _lib.require_cuda_f32(kd, "diffuse_albedo").reshape(-1, 3)
        ks = _lib.require_cuda_f32(ks.expand(sh + [3]), "specular_albedo").reshape(-1, 3)
        rough = _lib.require_cuda_f32(rough, "specular_roughness").reshape(-1)
        t1, t2 = ctx.renderer._tables_on(dev)
        g_diff, g_spec, g_rgb = _opt(g_diff, (-1, 3)), _opt(g_spec, (-1, 3)), _opt(g_rgb, (-1, 3))
        lib = _lib.load_train()
        with torch.cuda.device(dev):
            d_light = torch.zeros(1, dtype=torch.float32, device=dev)
            d_dist, d_rough = torch.empty_like(dist), torch.empty_like(rough)
            d_nrm, d_vd, d_kd, d_ks = (torch.empty((n, 3), dtype=torch.float32, device=dev) for _ in range(4))
            _lib.check_train(lib.iron_ggx_colocated_backward(ctx.light_value, dist.data_ptr(), nrm.data_ptr(), vd.data_ptr(), kd.data_ptr(),
                                                             ks.data_ptr(), rough.data_ptr(), t1.data_ptr(), t2.data_ptr(), n, _lib.ptr(g_diff),
                                                             _lib.ptr(g_spec), _lib.ptr(g_rgb), d_light.data_ptr(), d_dist.data_ptr(),
                                                             d_nrm.data_ptr(), d_vd.data_ptr(), d_kd.data_ptr(), d_ks.data_ptr(),
                                                             d_rough.data_ptr(), _lib.stream_ptr(dev)))
        s_dist, s_nrm, s_vd, s_kd, s_ks, s_rough = ctx.shapes
        d_ks = d_ks.reshape(sh + [3])
        if s_ks[-1] == 1:  # a [...,1] albedo was broadcast over the channels
            d_ks = d_ks.sum(dim=-1, keepdim=True)
        return (None, d_light.reshape(ctx.light_shape) if ctx.light_shape is not None else None, d_dist.reshape(s_dist), d_nrm.reshape(s_nrm),
                d_vd.reshape(s_vd), d_kd.reshape(s_kd), d_ks.reshape(s_ks), d_rough.reshape(s_rough))


class CompositeFn(torch.autograd.Function):
    """(rgb, specular_rgb, metallic_rgb, dielectric_rgb, env_light) = CompositeRenderer(light, distance, normal, viewdir, kd, ks,
    roughness, metallic_eta, metallic_k, dielectric_eta, env_light) (models/renderer_ggx.py:781-858).  env_light None = the
    point-light branch (the last output is then an empty placeholder).  "diffuse_rgb" is the same tensor as "rgb" in the
    reference, so the caller maps both keys to the first output."""

    NAMES = ("diffuse_albedo", "specular_albedo", "specular_roughness", "metallic_eta", "metallic_k", "dielectric_eta")

    @staticmethod
    def forward(ctx, renderer, light, distance, normal, viewdir, kd, ks, rough, m_eta, m_k, d_eta, env_light):
        ctx.set_materialize_grads(False)
        ctx.renderer = renderer
        ctx.light_shape = light.shape if torch.is_tensor(light) else None
        ctx.light_value = float(light)
        ctx.use_env = env_light is not None
        params = dict(zip(CompositeFn.NAMES, (kd, ks, rough, m_eta, m_k, d_eta)))
        params["metallic"] = params["dielectric"] = rough  # read and ignored by the reference (:829-831)
        if ctx.use_env:
            params["env_light"] = env_light
        with torch.no_grad():
            out = renderer._forward_values(ctx.light_value, distance, normal, viewdir, params, ctx.use_env)
        tensors = [normal, viewdir, kd, ks, rough, m_eta, m_k, d_eta, env_light if ctx.use_env else distance]
        ctx.shapes = tuple(t.shape for t in tensors)
        ctx.dist_shape = distance.shape if torch.is_tensor(distance) else None
        ctx.save_for_backward(*(t.detach() for t in tensors))
        env_out = out["env_light"] if ctx.use_env else normal.new_zeros(0)
        return out["rgb"], out["specular_rgb"], out["metallic_rgb"], out["dielectric_rgb"], env_out

    @staticmethod
    @once_differentiable  # closed-form first-order backward: differentiating it again raises instead of returning zeros
    def backward(ctx, g_rgb, g_spec, g_met, g_die, g_env):
        nrm, vd, kd, ks, rough, m_eta, m_k, d_eta, last = ctx.saved_tensors
        nrm = _lib.require_cuda_f32(nrm, "normal").reshape(-1, 3)
        n = nrm.shape[0]
        dev = nrm.device
        sh = list(ctx.shapes[0][:-1])

        def vec(t, name):
            t = _lib.require_cuda_f32(t, name)
            return (t.expand(sh + [3]) if t.shape[-1] != 3 else t).reshape(-1, 3).contiguous()

        def sca(t, name):
            return _lib.require_cuda_f32(t, name).reshape(-1)

        vd = vec(vd, "viewdir")
        kd_f, ks_f = vec(kd, "diffuse_albedo"), vec(ks, "specular_albedo")
        maps = [sca(t, k) for t, k in zip((rough, m_eta, m_k, d_eta), CompositeFn.NAMES[2:])]
        last = sca(last, "env_light" if ctx.use_env else "distance")
        t1, t2 = ctx.renderer._tables_on(dev)
        ups = [_opt(g, (-1, 3)) for g in (g_rgb, g_spec, g_met, g_die)]
        g_env = _opt(g_env, (-1,)) if ctx.use_env else None
        lib = _lib.load_train()
        with torch.cuda.device(dev):
            p = _lib.iron_composite_params()
            p.diffuse_albedo, p.specular_albedo, p.specular_roughness = kd_f.data_ptr(), ks_f.data_ptr(), maps[0].data_ptr()
            p.metallic_eta, p.metallic_k, p.dielectric_eta = maps[1].data_ptr(), maps[2].data_ptr(), maps[3].data_ptr()
            p.env_light = last.data_ptr() if ctx.use_env else None
            gi = _lib.iron_composite_grads_in()
            gi.d_rgb, gi.d_specular_rgb, gi.d_metallic_rgb, gi.d_dielectric_rgb = (_lib.ptr(u) for u in ups)
            gi.d_env_light_out = _lib.ptr(g_env)
            d_light = torch.zeros(1, dtype=torch.float32, device=dev)
            d_last = torch.empty_like(last)
            d_nrm, d_vd, d_kd, d_ks = (torch.empty((n, 3), dtype=torch.float32, device=dev) for _ in range(4))
            d_maps = [torch.empty_like(m) for m in maps]
            go = _lib.iron_composite_grads_out()
            go.d_light, go.d_normal, go.d_viewdir = d_light.data_ptr(), d_nrm.data_ptr(), d_vd.data_ptr()
            go.d_distance = None if ctx.use_env else d_last.data_ptr()
            go.d_env_light = d_last.data_ptr() if ctx.use_env else None
            go.d_diffuse_albedo, go.d_specular_albedo = d_kd.data_ptr(), d_ks.data_ptr()
            go.d_specular_roughness, go.d_metallic_eta, go.d_metallic_k, go.d_dielectric_eta = (m.data_ptr() for m in d_maps)
            _lib.check_train(lib.iron_composite_colocated_backward(ctx.light_value, None if ctx.use_env else last.data_ptr(), nrm.data_ptr(),
                                                                   vd.data_ptr(), C.byref(p), t1.data_ptr(), t2.data_ptr(), n, C.byref(gi), C.byref(go),
                                                                   _lib.stream_ptr(dev)))
        s_nrm, s_vd, s_kd, s_ks = ctx.shapes[:4]

        def unvec(g, shape):
            g = g.reshape(sh + [3])
            return (g.sum(dim=-1, keepdim=True) if shape[-1] == 1 else g).reshape(shape)

        d_light_out = d_light.reshape(ctx.light_shape) if (ctx.light_shape is not None and not ctx.use_env) else None
        d_dist = None
        if not ctx.use_env:
            d_dist = d_last.reshape(ctx.shapes[8])
        elif ctx.dist_shape is not None:
            d_dist = torch.zeros(ctx.dist_shape, dtype=torch.float32, device=dev)
        d_env = d_last.reshape(ctx.shapes[8]) if ctx.use_env else None
        return (None, d_light_out, d_dist, d_nrm.reshape(s_nrm), d_vd.reshape(s_vd), unvec(d_kd, s_kd), unvec(d_ks, s_ks)) + tuple(
            m.reshape(s) for m, s in zip(d_maps, ctx.shapes[4:8])) + (d_env,)


class ColocHeadFn(torch.autograd.Function):
    """(diffuse_rgb, specular_rgb, rgb) of the four simple co-located heads (models/renderer_ggx.py:149-395)."""

    @staticmethod
    def forward(ctx, head, light, distance, normal, viewdir, kd, ks, alpha):
        ctx.set_materialize_grads(False)
        ctx.head = head
        ctx.light_shape = light.shape if torch.is_tensor(light) else None
        ctx.light_value = float(light)
        with torch.no_grad():
            out = head._forward_values(ctx.light_value, distance, normal, viewdir, kd, ks, alpha)
        ctx.has_alpha = alpha is not None
        ctx.shapes = (distance.shape, normal.shape, viewdir.shape, kd.shape, ks.shape, None if alpha is None else alpha.shape)
        ctx.save_for_backward(distance.detach(), normal.detach(), viewdir.detach(), kd.detach(), ks.detach(),
                              *([alpha.detach()] if alpha is not None else []))
        return out["diffuse_rgb"], out["specular_rgb"], out["rgb"]

    @staticmethod
    @once_differentiable  # closed-form first-order backward: differentiating it again raises instead of returning zeros
    def backward(ctx, g_diff, g_spec, g_rgb):
        saved = list(ctx.saved_tensors)
        dist, nrm, vd, kd, ks = saved[:5]
        alpha = saved[5] if ctx.has_alpha else None
        head = ctx.head
        nrm = _lib.require_cuda_f32(nrm, "normal").reshape(-1, 3)
        n = nrm.shape[0]
        dev = nrm.device
        sh = list(ctx.shapes[1][:-1])
        dist = _lib.require_cuda_f32(dist, "distance").reshape(-1)
        vd = _lib.require_cuda_f32(vd, "viewdir").reshape(-1, 3)
        kd_f = _lib.require_cuda_f32(kd.expand(sh + [3]), "diffuse_albedo").reshape(-1, 3)
        ks_f = _lib.require_cuda_f32(ks.expand(sh + [3]), "specular_albedo").reshape(-1, 3)
        al = _lib.require_cuda_f32(alpha, "alpha").reshape(-1) if (alpha is not None and head.KIND == 3) else None
        ups = [_opt(g, (-1, 3)) for g in (g_diff, g_spec, g_rgb)]
        with torch.cuda.device(dev):
            d_light = torch.zeros(1, dtype=torch.float32, device=dev)
            d_dist = torch.empty_like(dist)
            d_nrm, d_vd, d_kd, d_ks = (torch.empty((n, 3), dtype=torch.float32, device=dev) for _ in range(4))
            d_al = torch.empty_like(al) if al is not None else None
            _lib.check_train(_lib.load_train().iron_coloc_head_backward(
                head.KIND, ctx.light_value, float(head.eta), float(head.k), dist.data_ptr(), nrm.data_ptr(), vd.data_ptr(), kd_f.data_ptr(),
                ks_f.data_ptr(), _lib.ptr(al), n, _lib.ptr(ups[0]), _lib.ptr(ups[1]), _lib.ptr(ups[2]), d_light.data_ptr(), d_dist.data_ptr(),
                d_nrm.data_ptr(), d_vd.data_ptr(), d_kd.data_ptr(), d_ks.data_ptr(), _lib.ptr(d_al), _lib.stream_ptr(dev)))
        s_dist, s_nrm, s_vd, s_kd, s_ks, s_al = ctx.shapes

        def unvec(g, shape):
            g = g.reshape(sh + [3])
            return (g.sum(dim=-1, keepdim=True) if shape[-1] == 1 else g).reshape(shape)

        d_alpha = None
        if ctx.has_alpha:
            d_alpha = d_al.reshape(s_al) if d_al is not None else torch.zeros(s_al, dtype=torch.float32, device=dev)
        return (None, d_light.reshape(ctx.light_shape) if ctx.light_shape is not None else None, d_dist.reshape(s_dist), d_nrm.reshape(s_nrm),
                d_vd.reshape(s_vd), unvec(d_kd, s_kd), unvec(d_ks, s_ks), d_alpha)


class NeRFFn(torch.autograd.Function):
    """(alpha [n,1], rgb [n,3]) = NeRF(input_pts [n,4], input_views [n,3]); differentiable w.r.t. the parameters (the reference
    feeds sample positions computed without grad, renderer.py:163-172, so the inputs get none)."""

    @staticmethod
    def forward(ctx, net, pts, views, *params):
        ctx.set_materialize_grads(False)
        ctx.net = net
        with torch.no_grad():
            alpha, rgb = net._forward_values(pts, views)
        ctx.save_for_backward(pts.detach(), views.detach())
        return alpha, rgb

    @staticmethod
    @once_differentiable  # closed-form first-order backward: differentiating it again raises instead of returning zeros
    def backward(ctx, d_alpha, d_rgb):
        net = ctx.net
        pts, views = ctx.saved_tensors
        pts = _lib.require_cuda_f32(pts, "input_pts").reshape(-1, net.d_in)
        views = _lib.require_cuda_f32(views, "input_views").reshape(-1, net.d_in_view)
        n = pts.shape[0]
        dev = pts.device
        lib = _lib.load_train()
        with torch.cuda.device(dev):
            arr, keep, grads = _train_layers(net, dev)
            desc = _lib.iron_nerf_train_desc()
            desc.D, desc.W, desc.d_in, desc.d_in_view = net.D, net.W, net.d_in, net.d_in_view
            desc.multires, desc.multires_view = net.multires, net.multires_view
            if len(net.skips) > 1:
                raise _lib.IronError("NeRF backward supports at most one skip layer")
            desc.skip = net.skips[0] if net.skips else -1
            desc.layers = arr
            nbytes = lib.iron_nerf_backward_workspace_bytes(C.byref(desc), n)
            if nbytes == 0:
                raise _lib.IronError("unsupported NeRF shape for the backward pass")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            d_alpha, d_rgb = _opt(d_alpha, (-1,)), _opt(d_rgb, (-1, 3))
            _lib.check_train(lib.iron_nerf_backward(C.byref(desc), pts.data_ptr(), views.data_ptr(), n, _lib.ptr(d_alpha), _lib.ptr(d_rgb),
                                                    ws.data_ptr(), nbytes, _lib.stream_ptr(dev)))
        del keep
        return (None, None, None) + tuple(grads)


class NeusCompositeFn(torch.autograd.Function):
    """The compositing of NeuSRenderer.render_core (models/renderer.py:279-344, background blend :174-178) from the network
    outputs at the section mid points: (sdf [n*m,1], grad [n*m,3], colour [n*m,3], 1/s, background density / colour) ->
    (color [n,3], weights, weight_sum, gradient_error, cdf, inside_sphere, weight_max).  The last three carry no gradient."""

    @staticmethod
    def forward(ctx, sdf, grad, color, inv_s, bg_density, bg_color, dists, pts, dirs, bg_dists, background_rgb, cos_anneal_ratio):
        ctx.set_materialize_grads(False)
        n, m = dists.shape
        dev = dists.device
        f32 = lambda t, name: _lib.require_cuda_f32(t.detach(), name)
        sdf_c, grad_c, color_c = f32(sdf, "sdf").reshape(-1), f32(grad, "gradients").reshape(-1, 3), f32(color, "sampled_color").reshape(-1, 3)
        dists_c, pts_c, dirs_c = f32(dists, "dists"), f32(pts, "pts").reshape(-1, 3), f32(dirs, "dirs").reshape(-1, 3)
        bg = bg_density is not None
        mo = bg_dists.shape[1] if bg else m
        keep = [sdf_c, grad_c, color_c, dists_c, pts_c, dirs_c]
        a = _lib.iron_neus_composite_args()
        a.dists, a.pts, a.dirs, a.sdf, a.grad, a.color = (t.data_ptr() for t in (dists_c, pts_c, dirs_c, sdf_c, grad_c, color_c))
        if bg:
            bgd, bgc, bgdist = f32(bg_density, "density").reshape(-1), f32(bg_color, "background colour").reshape(-1, 3), f32(bg_dists, "bg_dists")
            keep += [bgd, bgc, bgdist]
            a.bg_dists, a.bg_density, a.bg_color = bgdist.data_ptr(), bgd.data_ptr(), bgc.data_ptr()
        bgrgb = f32(background_rgb, "background_rgb").reshape(3) if background_rgb is not None else None
        a.background_rgb = _lib.ptr(bgrgb)
        a.n, a.m, a.mo = n, m, mo
        inv_s_value = float(inv_s)
        a.inv_s, a.cos_anneal_ratio = inv_s_value, float(cos_anneal_ratio)
        out_color = torch.empty((n, 3), dtype=torch.float32, device=dev)
        weights = torch.empty((n, mo), dtype=torch.float32, device=dev)
        cdf = torch.empty((n, m), dtype=torch.float32, device=dev)
        inside = torch.empty((n, m), dtype=torch.float32, device=dev)
        wsum = torch.empty((n, 1), dtype=torch.float32, device=dev)
        wmax = torch.empty((n, 1), dtype=torch.float32, device=dev)
        gacc = torch.zeros(2, dtype=torch.float32, device=dev)
        a.out_color, a.weights, a.cdf, a.inside_sphere = out_color.data_ptr(), weights.data_ptr(), cdf.data_ptr(), inside.data_ptr()
        a.weight_sum, a.weight_max, a.gradient_error_acc = wsum.data_ptr(), wmax.data_ptr(), gacc.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().iron_neus_composite(C.byref(a), _lib.stream_ptr(dev)))
        gerr = gacc[0] / (gacc[1] + 1e-5)
        ctx.meta = (n, m, mo, bg, inv_s_value, float(cos_anneal_ratio), torch.is_tensor(inv_s), tuple(sdf.shape), tuple(grad.shape), tuple(color.shape),
                    None if not bg else tuple(bg_density.shape), None if not bg else tuple(bg_color.shape))
        ctx.save_for_backward(gacc, *(keep + ([bgrgb] if bgrgb is not None else [])))
        ctx.has_bgrgb = bgrgb is not None
        ctx.mark_non_differentiable(cdf, inside, wmax)
        return out_color, weights, wsum, gerr, cdf, inside, wmax

    @staticmethod
    @once_differentiable  # closed-form first-order backward: differentiating it again raises instead of returning zeros
    def backward(ctx, d_color, d_weights, d_wsum, d_gerr, _dc, _di, _dm):
        n, m, mo, bg, inv_s_value, ca, inv_s_is_tensor, s_sdf, s_grad, s_color, s_bgd, s_bgc = ctx.meta
        saved = list(ctx.saved_tensors)
        gacc, sdf_c, grad_c, color_c, dists_c, pts_c, dirs_c = saved[:7]
        k = 7
        bgd = bgc = bgdist = None
        if bg:
            bgd, bgc, bgdist = saved[k:k + 3]
            k += 3
        bgrgb = saved[k] if ctx.has_bgrgb else None
        dev = dists_c.device
        a = _lib.iron_neus_composite_args()
        a.dists, a.pts, a.dirs, a.sdf, a.grad, a.color = (t.data_ptr() for t in (dists_c, pts_c, dirs_c, sdf_c, grad_c, color_c))
        if bg:
            a.bg_dists, a.bg_density, a.bg_color = bgdist.data_ptr(), bgd.data_ptr(), bgc.data_ptr()
        a.background_rgb = _lib.ptr(bgrgb)
        a.n, a.m, a.mo, a.inv_s, a.cos_anneal_ratio = n, m, mo, inv_s_value, ca
        g = _lib.iron_neus_composite_grads()
        ups = [_opt(d_color, (n, 3)), _opt(d_wsum, (n,)), _opt(d_weights, (n, mo)), _opt(d_gerr, (1,))]
        g.d_color, g.d_weight_sum, g.d_weights, g.d_gradient_error = (_lib.ptr(u) for u in ups)
        relax = gacc[1:2].contiguous()
        g.relax_count = relax.data_ptr()
        d_sdf, d_grad, d_col = torch.empty_like(sdf_c), torch.empty_like(grad_c), torch.empty_like(color_c)
        d_inv = torch.zeros(1, dtype=torch.float32, device=dev)
        d_bgd = torch.empty_like(bgd) if bg else None
        d_bgc = torch.empty_like(bgc) if bg else None
        g.d_sdf, g.d_grad, g.d_sample_color, g.d_inv_s = d_sdf.data_ptr(), d_grad.data_ptr(), d_col.data_ptr(), d_inv.data_ptr()
        g.d_bg_density, g.d_bg_color = _lib.ptr(d_bgd), _lib.ptr(d_bgc)
        with torch.cuda.device(dev):
            _lib.check_train(_lib.load_train().iron_neus_composite_backward(C.byref(a), C.byref(g), _lib.stream_ptr(dev)))
        return (d_sdf.reshape(s_sdf), d_grad.reshape(s_grad), d_col.reshape(s_color), d_inv.reshape(()) if inv_s_is_tensor else None,
                d_bgd.reshape(s_bgd) if bg else None, d_bgc.reshape(s_bgc) if bg else None, None, None, None, None, None, None)


def numeric_status(reset: bool = True, device=None) -> bool:
    """True when a backward pass since the last reset met an operand outside the range of its split-fp16 layer products
    (|x| > 65 504 or non-finite: the gradients of that pass are inf / NaN where fp32 autograd has numbers; include/iron_train.h).
    Synchronises the current stream of `device`."""
    lib = _lib.load_train()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    rc = lib.iron_train_numeric_status(1 if reset else 0, _lib.stream_ptr(dev))
    if rc not in (0, -6):
        raise _lib.IronError("iron_train_numeric_status failed (%d)" % rc)
    return rc == -6


def any_requires_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors)


def refuse_grad(what: str, *tensors) -> None:
    """Operators without a backward must not hand detached results to a training graph."""
    flat = []
    for t in tensors:
        flat += list(t.values()) if isinstance(t, dict) else [t]
    if any_requires_grad(*flat):
        raise NotImplementedError("%s has no backward pass in iron_amd (SURVEY 8 row f-2 covers SDFNetwork, RenderingNetwork and "
                                  "GGXColocatedRenderer); call it under torch.no_grad() or on detached inputs" % what)
