"""Where the host time of a C3 training step goes: torch.profiler over a few steps of tools/train_step.py's loop (CPU-side operator
table + total device kernel time), to separate launch/sync overhead from kernel time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity


def main(size=512, steps=3):
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    dev = torch.device("cuda", 0)
    nets = {k: v.to(dev) for k, v in scenes.build_networks("S1").items()}
    K, W2C = scenes.fixture_camera_matrices(size, size)
    cam = Camera(size, size, K.to(dev), W2C.to(dev))
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    params = [p for n in nets.values() for p in n.parameters()]
    opt = torch.optim.Adam(params, lr=1e-5)
    target = torch.rand(size, size, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    tracer = RayTracer()

    def step():
        res = render_camera(cam, nets["sdf_network"], tracer, nets, fn, fill_holes=False, handle_edges=True, is_training=True)
        mask = res["convergent_mask"] | res["edge_mask"]
        img_loss = (res["color"][mask] - target[mask]).abs().mean()
        eik_pts = torch.empty(size * size // 2, 3, device=dev).uniform_(-1.0, 1.0)
        eik = ((nets["sdf_network"].gradient(eik_pts).norm(dim=-1) - 1.0) ** 2).mean()
        loss = img_loss + 0.1 * eik
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    print("wall per step %.2f ms" % ((time.perf_counter() - t0) / steps * 1e3))
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False) as prof:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    ka = prof.key_averages()
    dev_total = sum(getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)) for e in ka)
    print("device kernel time per step %.2f ms" % (dev_total / steps / 1e3))
    print(ka.table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))


if __name__ == "__main__":
    main()
