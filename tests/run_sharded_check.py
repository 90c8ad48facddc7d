"""Multi-rank rehearsal on ONE GPU (launched by tests/test_gpu_sharded.py with torchrun, gloo backend):
every rank traces + shades its interleaved tiles of 2 views on cuda:0, exchanges the chunk counts and gathers
the records; rank 0 compares the assembled images with the single-process render_camera result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    from iron_amd.sharding import ShardedRenderer

    res_px = int(os.environ.get("IRON_CHECK_RES", "96"))
    nets = {k: v.to(dev) for k, v in scenes.build_networks("S1").items()}
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    cams = []
    for v in range(2):
        K, W2C = scenes.fixture_camera_matrices(res_px, res_px, yaw_deg=45.0 * v)
        cams.append(Camera(res_px, res_px, K.to(dev), W2C.to(dev)))
    # small chunk so that several chunks exist and straddle ranks
    sh = ShardedRenderer(nets["sdf_network"], nets, RayTracer(), fn, tile=16, chunk=2000)
    out = sh.render(cams, collect_stats=True)
    ok = True
    if rank == 0:
        import iron_amd.raytracer as rt
        for v, cam in enumerate(cams):
            # single-process reference with the same chunking
            res = rt.raytrace_camera(cam, nets["sdf_network"], RayTracer(), max_num_rays=2000)
            rt.render_normal_and_color(res, nets["sdf_network"], nets, fn)
            for k in ("convergent_mask", "distance", "points", "sdf", "depth", "color", "normal", "diffuse_albedo",
                      "specular_albedo", "specular_roughness", "diffuse_color", "specular_color"):
                a, b = out[k][v].cpu().numpy(), res[k].cpu().numpy()
                if not np.array_equal(a, b):
                    ok = False
                    print("MISMATCH view %d key %s: %d elements differ, max |d| %g" %
                          (v, k, int((a != b).sum()), float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())))
        print("SHARDED_CHECK", "OK" if ok else "FAIL", "world", world, "hits", int(out["convergent_mask"].sum()))
    # the whole-image passes: hole filling on every rank (all-gather of the trace records), silhouette edges on rank 0
    out2 = sh.render(cams, fill_holes=True, handle_edges=True)
    if rank == 0:
        import iron_amd.raytracer as rt
        ok2 = True
        for v, cam in enumerate(cams):
            res = rt.raytrace_camera(cam, nets["sdf_network"], RayTracer(), max_num_rays=2000, fill_holes=True, detect_edges=True)
            rt.render_normal_and_color(res, nets["sdf_network"], nets, fn)
            if res["edge_mask"].sum() > 0:
                rt.render_edge_pixels(res, cam, nets["sdf_network"], RayTracer(), nets, fn)
            for k in ("convergent_mask", "distance", "points", "depth", "color", "normal", "diffuse_albedo", "specular_roughness",
                      "edge_mask", "uv"):
                a, b = out2[k][v].cpu().numpy(), res[k].cpu().numpy()
                if not np.array_equal(a, b):
                    ok2 = False
                    print("MISMATCH (holes+edges) view %d key %s: %d elements differ" % (v, k, int((a != b).sum())))
        print("SHARDED_EDGES_CHECK", "OK" if ok2 else "FAIL", "edge pixels", int(sum(int(m.sum()) for m in out2["edge_mask"])))
        ok = ok and ok2
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
