"""One warm frame with fill_holes + handle_edges under rocprofv3 --kernel-trace: do the silhouette pass and the hit shading overlap?
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/eo -o eo -- python3 tools/edge_overlap_trace.py
   python3 tools/edge_overlap_trace.py --report gpurun_out/eo"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    torch.set_grad_enabled(False)
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    dev = torch.device("cuda", 0)
    nets = {k: v.to(dev) for k, v in scenes.build_networks("S0").items()}
    K, W2C = scenes.fixture_camera_matrices(800, 800)
    cam = Camera(800, 800, K.to(dev), W2C.to(dev))
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    tr = RayTracer()
    for _ in range(3):
        render_camera(cam, nets["sdf_network"], tr, nets, fn, fill_holes=True, handle_edges=True)
    torch.cuda.synchronize()


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last frame: from the last k_camera_rays / first sphere of the final third
    starts = [i for i, r in enumerate(rows) if "k_sphere" in r["Kernel_Name"]]
    # a frame has 2 sphere launches (main + side rays): take the second-to-last as the frame start
    i0 = starts[-2]
    t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[i0:]:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("iron::", "")[:48]
        print("%-50s q%-3s %9.3f -> %9.3f ms" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        run()
