"""Golden for BASELINE.json's full-size configuration C1 (S0, 800x800) -- BUILD CONTAINER ONLY (minutes of CPU).

Runs the real reference's render_camera (fill_holes=False, handle_edges=False) at 800x800 in fp32 and fp64 and stores
what a full-size parity test needs in a small file: the complete hit mask (bit-packed), and colour / normal / distance on
the pixel sub-lattice [::4, ::4] (40 000 pixels) for both precisions, plus the fp32-vs-fp64 floor over ALL pixels.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_800.py [S0|S1]
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

from models.raytracer import RayTracer, render_camera  # noqa: E402  (reference)
from models.renderer_ggx import GGXColocatedRenderer  # noqa: E402

RES, STRIDE = 800, 4
SCENE = sys.argv[1] if len(sys.argv) > 1 else "S0"  # S0: the BASELINE headline scene; S1: the bumpy one (chaotic grazing pixels)


def run(dtype):
    nets = MG.build_reference_networks(SCENE)
    if dtype == torch.float64:
        nets = {k: v.double() for k, v in nets.items()}
        cam32 = MG.fixture_camera(RES, RES)
        cam = MG.Camera64(RES, RES, cam32.K.double(), cam32.W2C.double())
    else:
        cam = MG.fixture_camera(RES, RES)
    renderer = GGXColocatedRenderer(use_cuda=False)  # fp32 tables in both runs, as make_golden.py's fp64 runs
    fn = MG.make_render_fn(nets, renderer, dtype)
    t0 = time.time()
    with torch.no_grad():
        res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=False, handle_edges=False,
                            is_training=False)
    print(dtype, "%.0f s" % (time.time() - t0), "hits", int(res["convergent_mask"].sum()), flush=True)
    return res


def main():
    r32 = run(torch.float32)
    r64 = run(torch.float64)
    m32, m64 = r32["convergent_mask"].numpy(), r64["convergent_mask"].numpy()
    both = m32 & m64
    c32, c64 = r32["color"].numpy().astype(np.float64), r64["color"].numpy()
    floor = float(np.linalg.norm(c32[both] - c64[both]) / np.linalg.norm(c64[both]))
    out = {"mask_bits": np.packbits(m32.reshape(-1)), "mask64_bits": np.packbits(m64.reshape(-1)), "res": np.int64(RES),
           "stride": np.int64(STRIDE)}
    for k in ("color", "normal", "distance"):
        out[k] = r32[k].numpy()[::STRIDE, ::STRIDE].astype(np.float32)
        out[k + "_fp64"] = r64[k].numpy()[::STRIDE, ::STRIDE].astype(np.float64)
    # every pixel, not only the sub-lattice: colour summed over 8x8 tiles (pixels that hit in both runs), fp32 and fp64 run
    TILE = 8
    w = both[..., None].astype(np.float64)
    tiles = lambda img: (img * w).reshape(RES // TILE, TILE, RES // TILE, TILE, 3).sum(axis=(1, 3))
    out["tile"] = np.int64(TILE)
    out["color_tile_sum"] = tiles(c32)
    out["color_tile_sum_fp64"] = tiles(c64)
    out["tile_hits"] = both.reshape(RES // TILE, TILE, RES // TILE, TILE).sum(axis=(1, 3)).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "g12_%s_800.npz" % SCENE), **out)
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path))
    keys = {"n_conv_%s_800" % SCENE: int(m32.sum()), "mask_flips_ref32_ref64_%s_800" % SCENE: int((m32 != m64).sum()),
            "colour_rel_l2_ref32_ref64_%s_800" % SCENE: floor}
    meta.update(keys)
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print(keys)


if __name__ == "__main__":
    main()
