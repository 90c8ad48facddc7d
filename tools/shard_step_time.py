#!/usr/bin/env python3
"""What ONE rank of an N-way tile-sharded render spends per step, measured the way a rank runs: its rank-local phases
(trace_begin -> [the chunk-count table as the MAX all-reduce leaves it] -> trace_finish -> shade) back to back, K
steps without a host sync in between.  The per-phase-synchronised figures of render_emulated() add host launch latency to every
phase; this is the steady-state cost.

    python tools/shard_step_time.py [res] [world] [ranks...]     env: IRON_SHARD_TILE, IRON_TRACE_SPLIT
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.set_grad_enabled(False)
from iron_amd import _lib, scenes  # noqa: E402
from iron_amd.raytracer import Camera, RayTracer, render_camera  # noqa: E402
from iron_amd.renderer_ggx import GGXColocatedRenderer  # noqa: E402
from iron_amd.rendering_func import make_render_fn  # noqa: E402
from iron_amd.sharding import ShardedRenderer  # noqa: E402


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ranks = [int(x) for x in sys.argv[3:]] or list(range(world))
    tile = int(os.environ.get("IRON_SHARD_TILE", "8"))
    K_ = int(os.environ.get("IRON_STEPS", "5"))
    dev = torch.device("cuda", 0)
    nets = {k: v.to(dev) for k, v in scenes.build_networks("S0").items()}
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    sdf = nets["sdf_network"]
    K, W2C = scenes.fixture_camera_matrices(res, res)
    cam = Camera(res, res, K.to(dev), W2C.to(dev))

    def wall(f, n):
        for _ in range(4):
            f()
        best = None
        for _ in range(4):    # best of 4 runs: a caching-allocator stall inflates single runs
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                f()
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / n * 1e3
            best = t if best is None else min(best, t)
        return best

    t_frame = wall(lambda: render_camera(cam, sdf, RayTracer(), nets, fn, fill_holes=False, handle_edges=False), K_)
    out = []
    # the chunk-global bisection counts as the MAX all-reduce leaves them (every rank finishes its bisections up to the same counts)
    table = None
    for r in range(world):
        sh0 = ShardedRenderer(sdf, nets, RayTracer(), fn, tile=tile, chunk=50000, world=world, rank=r)
        st0 = sh0.trace_begin([cam])
        table = st0["chunk_iters"].clone() if table is None else torch.maximum(table, st0["chunk_iters"])
        sh0.trace_finish(st0)
    for r in ranks:
        sh = ShardedRenderer(sdf, nets, RayTracer(), fn, tile=tile, chunk=50000, world=world, rank=r)

        def step():
            st = sh.trace_begin([cam])
            st["chunk_iters"].copy_(table)
            st = sh.trace_finish(st)
            return sh.shade(st)
        ms = wall(step, K_)
        _lib.profile_enable(True)
        _lib.profile_read()
        for _ in range(K_):
            step()
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        ksum = sum(v[0] for v in prof.values()) / K_
        out.append(ms)
        print("rank %d/%d: %.2f ms per step (kernel events sum %.2f: %s)" % (r, world, ms, ksum, {k: round(v[0] / K_, 2) for k, v in prof.items() if v[1]}), flush=True)
    print("%dx%d tile %d world %d: frame %.2f ms; rank steps max %.2f mean %.2f -> strong scaling (kernels + host glue, no collectives) %.2fx"
          % (res, res, tile, world, t_frame, max(out), sum(out) / len(out), t_frame / max(out)))


if __name__ == "__main__":
    main()
