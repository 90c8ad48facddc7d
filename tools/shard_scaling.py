#!/usr/bin/env python3
"""Per-shard device time of the 8-rank tile-sharded render played through on ONE card (iron_amd.sharding.render_emulated) and
the strong-scaling factor it predicts: T(frame) / max_r T(shard r), kernels only (the all-reduce, the gather and rank 0's
assemble come on top and are reported beside it).  Writes gpurun_out/shard_scaling_<res>.json; copy into profiles/ to commit.

    python tools/shard_scaling.py [res ...]          (default: 800 1600)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from iron_amd import scenes  # noqa: E402
from iron_amd.raytracer import Camera, RayTracer, render_camera  # noqa: E402
from iron_amd.renderer_ggx import GGXColocatedRenderer  # noqa: E402
from iron_amd.rendering_func import make_render_fn  # noqa: E402
from iron_amd.sharding import render_emulated  # noqa: E402


def frame_ms(f, reps=3):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [800, 1600]
    tile = int(os.environ.get("IRON_SHARD_TILE", "8"))
    nets = {k: v.cuda() for k, v in scenes.build_networks("S0").items()}
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    sdf = nets["sdf_network"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for res in sizes:
        K, W2C = scenes.fixture_camera_matrices(res, res, yaw_deg=45.0 if res >= 1600 else 0.0)
        cam = Camera(res, res, K.cuda(), W2C.cuda())
        t_frame = frame_ms(lambda: render_camera(cam, sdf, RayTracer(), nets, fn, fill_holes=False, handle_edges=False))
        out = {"res": res, "tile": tile, "frame_ms": t_frame}
        for world in (2, 4, 8):
            render_emulated(world, [cam], sdf, nets, fn, RayTracer, tile=tile)
            runs = [render_emulated(world, [cam], sdf, nets, fn, RayTracer, tile=tile) for _ in range(3)]
            ms = [min(r[1][k] for r in runs) for k in range(world)]
            asm = min(r[2] for r in runs)
            out["n%d" % world] = {"shard_ms": [round(x, 3) for x in ms], "max_ms": max(ms), "mean_ms": sum(ms) / world, "assemble_ms": asm,
                                  "predicted_strong_scaling_kernels_only": t_frame / max(ms),
                                  "predicted_with_rank0_assemble": t_frame / (max(ms) + asm)}
            print("%dx%d world %d: frame %.2f ms; shards max %.2f mean %.2f; assemble %.2f; predicted %.2fx (%.2fx with assemble)" %
                  (res, res, world, t_frame, max(ms), sum(ms) / world, asm, t_frame / max(ms), t_frame / (max(ms) + asm)), flush=True)
        with open(os.path.join(ROOT, "gpurun_out", "shard_scaling_%d.json" % res), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
