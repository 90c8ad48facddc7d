"""CPU-side checks: the C-ABI library builds/loads and exports every symbol include/iron_hip.h declares;
host logic (constructors, camera algebra, tile sharding, records) behaves like the reference's."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from iron_amd import _lib, build
    build.build()
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "iron_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(iron_[a-z_0-9]+)\s*\(", hdr))
    assert {"iron_trace", "iron_shade_ggx", "iron_sdf_forward", "iron_net_create"} <= declared
    for name in sorted(declared):
        assert hasattr(lib, name), "symbol %s is declared but not exported" % name
        assert name in _lib.SYMBOLS, "symbol %s has no ctypes binding" % name
    assert lib.iron_version() == 1
    assert lib.iron_strerror(-2).decode().startswith("unsupported")
    # argument validation happens before any device work: callable without a GPU
    assert lib.iron_sdf_forward(None, None, 4, None, 1, None) == -1
    assert lib.iron_trace_workspace_bytes(1000, None) > 0
    assert lib.iron_shade_workspace_bytes(1000) > 0


def test_training_library_exports_every_declared_symbol():
    """include/iron_train.h <-> libiron_train.so (the backward passes; links rocBLAS, loads without a GPU)."""
    from iron_amd import _lib, build
    build.build()
    lib = _lib.load_train()
    hdr = open(os.path.join(ROOT, "include", "iron_train.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(iron_[a-z_0-9]+)\s*\(", hdr))
    assert {"iron_sdf_backward", "iron_render_backward", "iron_ggx_colocated_backward"} <= declared
    for name in sorted(declared):
        assert hasattr(lib, name), "symbol %s is declared but not exported" % name
        assert name in _lib.TRAIN_SYMBOLS, "symbol %s has no ctypes binding" % name
    # argument validation / workspace sizing happen on the host
    assert lib.iron_sdf_backward(None, None, -1, None, None, None, None, 0, None) == -1
    layers = (_lib.iron_train_layer * 9)()
    dims = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
    for l, (o, i) in zip(layers, dims):
        l.out_dim, l.in_dim = o, i
    d = _lib.iron_sdf_train_desc()
    d.n_linear, d.multires, d.skip_layer, d.layers = 9, 6, 4, layers
    per_point = lib.iron_sdf_backward_workspace_bytes(ctypes.byref(d), 65536) / 65536
    assert 30e3 < per_point < 60e3            # ~37 KB of kept activations per point (+ the two gradient buffers)
    d.skip_layer = 3                          # inconsistent with the layer shapes
    assert lib.iron_sdf_backward_workspace_bytes(ctypes.byref(d), 16) == 0


def test_autograd_wrappers_refuse_cpu_parameters():
    """is_training=True has no eager path either: CPU parameters / tensors raise."""
    from iron_amd import scenes
    nets = scenes.build_networks("S0")
    with pytest.raises(RuntimeError):
        nets["sdf_network"].get_all(torch.zeros(4, 3), is_training=True)
    z = torch.zeros(4, 3)
    with pytest.raises(RuntimeError):
        nets["diffuse_albedo_network"](z, z, z, torch.zeros(4, 256))


def test_cpu_tensors_are_refused_not_computed():
    """No CPU fallback: the product path raises on CPU tensors."""
    from iron_amd import scenes, _lib
    nets = scenes.build_networks("S0")
    with pytest.raises(RuntimeError):
        nets["sdf_network"](torch.zeros(4, 3))
    from iron_amd.raytracer import intersect_sphere
    with pytest.raises(RuntimeError):
        intersect_sphere(torch.zeros(4, 3), torch.ones(4, 3), 1.0)
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    z = torch.zeros(2, 3)
    with pytest.raises(RuntimeError):
        GGXColocatedRenderer()(1.0, z[:, :1], z, z, params={"diffuse_albedo": z, "specular_albedo": z, "specular_roughness": z[:, :1]})


def test_state_dict_layout_matches_reference_checkpoints():
    """Key names / shapes of utils/ckpt_loader.py-style checkpoints: lin{l}.weight_g [out,1], weight_v, bias."""
    from iron_amd import scenes
    nets = scenes.build_networks("S0")
    sd = nets["sdf_network"].state_dict()
    assert list(sd.keys())[:3] == ["lin0.weight_g", "lin0.weight_v", "lin0.bias"] or set(list(sd.keys())[:3]) == {"lin0.bias", "lin0.weight_g", "lin0.weight_v"}
    assert sd["lin0.weight_v"].shape == (256, 39) and sd["lin0.weight_g"].shape == (256, 1)
    assert sd["lin3.weight_v"].shape == (217, 256) and sd["lin4.weight_v"].shape == (256, 256)
    assert sd["lin8.weight_v"].shape == (257, 256)
    assert sum(v.numel() for k, v in sd.items() if "weight_v" in k or "bias" in k) == 526810
    d = nets["diffuse_albedo_network"].state_dict()
    assert d["lin0.weight_v"].shape == (256, 289) and d["lin4.weight_v"].shape == (3, 256)
    s = nets["specular_albedo_network"].state_dict()
    assert s["lin0.weight_v"].shape == (256, 298)
    # round trip
    from iron_amd.fields import SDFNetwork
    m = SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0)
    m.load_state_dict(sd)
    assert torch.equal(m.lin5.weight_v, nets["sdf_network"].lin5.weight_v)


def test_camera_algebra_matches_oracle():
    from iron_amd.raytracer import Camera
    from iron_amd import scenes
    from oracle import iron_ref as R
    K, W2C = scenes.fixture_camera_matrices(512, 512)
    cam = Camera(512, 512, K, W2C)
    oc = R.CameraSpec(512, 512, K, W2C)
    assert torch.equal(cam.K_inv, oc.K_inv) and torch.equal(cam.C2W, oc.C2W)
    assert torch.equal(cam.get_uv(), oc.get_uv())
    c2, img, msk = cam.crop_region(64, 64, ul_corner=(224, 224), image=torch.zeros(512, 512, 3), mask=torch.ones(512, 512))
    assert img.shape == (64, 64, 3) and msk.shape == (64, 64)  # 3-tuple like raytracer.py:351
    assert torch.equal(c2.K, oc.crop(64, 64, (224, 224)).K)
    c3, none = cam.resize(0.25)
    assert none is None and c3.W == 128 and torch.equal(c3.K, oc.scaled(128, 128).K)
    p = torch.rand(10, 3) - 0.5
    uv = cam.project(p)
    assert uv.shape == (10, 2)
    assert torch.allclose(cam.get_camera_origin(), oc.C2W[:3, 3])


def test_tile_sharding_partitions_the_image():
    from iron_amd.sharding import tile_pixels, shard_sizes, global_ray_index, chunks_per_view, RECORD_WIDTH, split_record
    for (H, W, tile, world) in [(800, 800, 32, 8), (100, 70, 32, 3), (64, 64, 32, 2), (33, 65, 16, 4)]:
        allpix = torch.cat([tile_pixels(H, W, tile, world, r) for r in range(world)])
        assert allpix.numel() == H * W
        assert torch.equal(torch.sort(allpix).values, torch.arange(H * W))
        assert sum(shard_sizes(H, W, tile, world)) == H * W
    # interleaving balances the shards (800x800 / 8 ranks: 625 tiles -> 78 or 79 tiles each)
    sizes = shard_sizes(800, 800, 32, 8)
    assert max(sizes) - min(sizes) <= 32 * 32
    # chunk ids never straddle views
    pix = tile_pixels(800, 800, 32, 8, 3)
    cpv = chunks_per_view(800, 800, 50000)
    assert cpv == 13
    for v in (0, 1, 7):
        ch = global_ray_index(pix, v, 800, 800, 50000) // 50000
        assert int(ch.min()) >= v * cpv and int(ch.max()) < (v + 1) * cpv
        assert torch.equal(ch - v * cpv, pix // 50000)
    rec = torch.zeros(4, 4, RECORD_WIDTH)
    rec[..., 0] = 1.0
    d = split_record(rec)
    assert d["convergent_mask"].dtype == torch.bool and d["color"].shape == (4, 4, 3) and d["depth"].shape == (4, 4)


def test_camera_resize_resamples_the_image_by_area():
    """Camera.resize(factor, image) (models/raytracer.py:353-364, cv2.INTER_AREA there): shrinking by an integer ratio is the exact
    box mean, by a fractional ratio the coverage-weighted mean (constant images stay constant, the total is preserved), dtypes and
    array kinds come back as given.  (cv2 is absent from the build container: parity-unpinned against OpenCV itself.)"""
    import numpy as np
    import torch
    from iron_amd.raytracer import Camera
    K = torch.eye(4); K[0, 0] = K[1, 1] = 100.0; K[0, 2], K[1, 2] = 32.0, 24.0
    cam = Camera(64, 48, K, torch.eye(4))
    g = torch.Generator().manual_seed(0)
    img = torch.rand(48, 64, 3, generator=g)
    c2, small = cam.resize(0.25, image=img)
    assert (c2.W, c2.H) == (16, 12) and small.shape == (12, 16, 3)
    want = img.reshape(12, 4, 16, 4, 3).mean(dim=(1, 3))
    assert float((small - want).abs().max()) <= 1e-6
    assert abs(float(c2.K[0, 0]) - 25.0) < 1e-6 and abs(float(c2.K[1, 2]) - 6.0) < 1e-6
    c3, frac = cam.resize(0.4, image=img[..., 0])          # 64 -> 25, 48 -> 19: fractional footprints
    assert frac.shape == (19, 25)
    assert abs(float(frac.mean()) - float(img[..., 0].mean())) <= 2e-3
    _, const = cam.resize(0.4, image=torch.full((48, 64), 0.37))
    assert float((const - 0.37).abs().max()) <= 1e-6
    _, u8 = cam.resize(0.5, image=(img.numpy() * 255).astype(np.uint8))
    assert isinstance(u8, np.ndarray) and u8.dtype == np.uint8 and u8.shape == (24, 32, 3)
    cam_only, none = cam.resize(0.5)
    assert none is None and cam_only.W == 32
