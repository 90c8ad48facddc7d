#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the stage-2 forward render (ray-gen -> sphere trace -> GGX shade ->
pixels) on synthetic 800x800 views (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Both forms start N ranks, one per GPU: without WORLD_SIZE in the environment `--gpus N` (N > 1) makes this process a
launcher -- it never touches the GPU, starts N fresh children of this script with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT set, relays rank 0's JSON line and exits non-zero if any rank does.

Workload (config C1): scene S0 (seed-0 geometric-init SDF, `ggx` material nets), fixture camera
rescaled to 800x800, tracer defaults, fill_holes=False, handle_edges=False, fp32.
  N = 1 : one step = render_camera() of one 800x800 view through the drop-in operator surface.
  N > 1 : rays sharded over the N ranks by interleaved 8x8 tiles; one MAX all-reduce of the per-chunk bisection
          counts and one RCCL gather of the finished pixel records to rank 0 per step.
          --scaling weak (default): one step = N views (fixture pose orbited by k*45 deg, BASELINE config C4's shape),
          every view sharded over all ranks; per-GPU work is fixed -> "scaling": "weak".
          --scaling strong: one step = ONE view sharded over the N ranks (north_star: "rays of one render call shard by
          image tile across the GPUs"); total work is fixed -> "scaling": "strong".
          Whichever mode is timed as `value`, the OTHER mode is timed right after it in the same run and reported in the
          "other_scaling" object, so one SCALE record carries both curves.
  N = 1 also reports "predicted_strong_scaling": the 8 tile-shards of the frame run one after the other on this card
          (iron_amd.sharding.render_emulated) -- T(frame) / max_r T(shard r) -- and "f32_core": the same workload on the
          exact-fp32 MFMA core (a child process with IRON_MLP_CORE=f32).
One JSON line is printed by rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_EVAL = 918016          # one SDF evaluation (sdf only), 459 008 MAC
FLOP_SDF_GRAD = 2 * (524544 + 459008)   # per hit: full forward (257 outs) + input gradient
FLOP_MATERIALS = 2 * 818176             # per hit: the three material MLPs
FLOP_PER_HIT = FLOP_SDF_GRAD + FLOP_MATERIALS
PEAK_FP32_MFMA = 157.3e12       # MI355X_MICROARCH.md: 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA = 2.5e15          # dense f16/bf16 MFMA (the pipe the default "h2" core executes on)
MFMA_FLOP_PER_EVAL_H2 = 3 * 2 * (7 * 256 * 256 + 2 * 8 * 3 * 16 * 32)   # executed: 3 f16 products per MAC, padded shapes
MFMA_FLOP_PER_EVAL_F32 = 7488 * 4096 // 32
SHARD_TILE = 8                  # interleaved tile edge of the N>1 sharding (iron_amd/sharding.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--scene", default="S0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--edges", action="store_true",
                    help="N=1 only: fill_holes=True, handle_edges=True (the reference's validation render, SURVEY row f-1) "
                         "instead of the headline configuration")
    ap.add_argument("--cpu-sample-res", type=int, default=160)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N>1: weak = N views per step, every view tile-sharded over the N ranks (default); strong = one view per step")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras of the default line (other scaling mode, 8-shard emulation, exact-fp32-core child run)")
    ap.add_argument("--workload", choices=["c1", "c2", "c3"], default="c1",
                    help="c1 (default): the BASELINE headline, 800x800 sphere-trace + GGX shade.  c2 / c3 (N=1 only): the other two "
                         "single-GPU BASELINE configurations, reported in their own units with the roofline of their dominant kernel and "
                         "the oracle timed on a bounded sample -- "
                         "c2 = stage-1 NeuS forward, 4096 rays x 128 samples per step; c3 = one stage-2 training step with edge "
                         "sampling at 512x512")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="N>1: only join the process group (IRON_BENCH_BACKEND, default nccl), all-reduce the rank count and print the "
                         "line's launch fields -- the launcher's own test (tests/test_bench_launcher.py runs it on CPU over gloo)")
    return ap.parse_args()


_JSON_OUT = None


def _reserve_stdout():
    """N > 1: rank 0's stdout must carry the JSON line and nothing else, and a communication backend may write notes of its own to
    fd 1 (gloo prints its connection count there).  File descriptor 1 is pointed at stderr for the rest of the run; `emit` writes
    to the original stdout."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        _JSON_OUT = os.fdopen(saved, "w")


def emit(obj) -> None:
    out = _JSON_OUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a torchrun around it: this process becomes the launcher.  It has not touched the GPU
    (importing torch does not; torch.cuda.device_count() does not initialise HIP on this image) and never will: it starts N fresh
    children -- never an exec of a process that holds the device -- one rank per GPU, waits for all of them, relays rank 0's
    stdout (the JSON line) and returns non-zero if any rank failed.  IRON_BENCH_ONE_GPU=1 puts every rank on cuda:0 (rehearsal
    on a one-GPU box, with IRON_BENCH_BACKEND=gloo)."""
    import subprocess
    import threading
    n = a.gpus
    one_gpu = os.environ.get("IRON_BENCH_ONE_GPU", "0") == "1"
    if not a.rendezvous_only and not one_gpu:
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s) (IRON_BENCH_ONE_GPU=1 IRON_BENCH_BACKEND=gloo rehearses "
                             "the %d-rank path on one card)" % (n, have, n))
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IRON_BENCH_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    lines = []

    def pump():
        for line in procs[0].stdout:
            lines.append(line)
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print("[bench] rank %d exited with %d: stopping the other ranks" % (r, code), file=sys.stderr, flush=True)
                for q in alive:          # exactly the processes started above, by handle
                    procs[q].terminate()
        time.sleep(0.05)
    t.join(timeout=5)
    for line in lines:   # stdout carries the JSON line only (a backend's own chatter on rank 0's stdout, e.g. gloo's connection notes, goes to stderr)
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
    sys.stdout.flush()
    if rc == 0 and not any(l.startswith("{") for l in lines):
        print("[bench] rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


def rendezvous_only(a, world, rank):
    """The launch protocol without a render: join the group, count the ranks with an all-reduce, print the launch fields."""
    import torch.distributed as dist
    _reserve_stdout()
    backend = os.environ.get("IRON_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        dev = torch.device("cuda", 0 if os.environ.get("IRON_BENCH_ONE_GPU", "0") == "1" else int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)
        one = torch.ones(1, dtype=torch.int64, device=dev)
    else:
        dist.init_process_group(backend)
        one = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(one)
    if rank == 0:
        emit({"rendezvous_only": True, "n_gpus": a.gpus, "ranks_seen": dist.get_world_size(), "ranks_counted": int(one.item()),
              "backend": backend, "launcher": os.environ.get("IRON_BENCH_LAUNCHER", "torchrun")})
    dist.barrier()
    dist.destroy_process_group()


def _cpu_share():
    """(cores this job may actually run on, details): the scheduler affinity, cut to the cgroup CPU quota when there is one -- the
    GPU box shows all 256 hardware threads of its host in the affinity mask but grants a job a 16-core share (256 torch threads on
    that share run ~20x slower than 16: the first attempt at "all host cores" did not finish)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                                           # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    share = avail if quota is None else max(1, min(avail, int(quota + 0.5)))
    if quota is None and avail > 64:
        share = 16    # no quota visible but a whole-host mask: the documented share of a one-GPU job on this pool
    return share, {"affinity_cores": avail, "cpu_count": os.cpu_count(), "cgroup_quota_cores": quota}


def _host_threads():
    return max(1, int(os.environ.get("IRON_CPU_THREADS", _cpu_share()[0])))


def _c2_cpu_baseline(threads):
    """The oracle's NeuS render (torch-CPU port of models/renderer.py:346-453) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import neus_frames
    from oracle import iron_ref as R
    from oracle import neus_ref as N
    from _util import cpu_sd
    torch.set_num_threads(threads)
    nets = neus_frames.build_networks(cuda=False)
    sc = N.NeusScene(cpu_sd(nets["sdf"]), R.SDFSpec(), cpu_sd(nets["color"]), cpu_sd(nets["nerf"]), nets["deviation"].variance.detach().clone(), n_outside=32)
    n = 2048
    o, d, near, far = neus_frames.rays(n)
    with torch.no_grad():
        N.render(sc, o[:16], d[:16], near[:16], far[:16], cos_anneal_ratio=1.0)
        t0 = time.perf_counter()
        N.render(sc, o, d, near, far, cos_anneal_ratio=1.0)
        dt = time.perf_counter() - t0
    return {"value": n / dt / 1e3, "unit": "krays/s", "cores": threads, "kind": "port",
            "sample": "%d rays x (64 + 4x16 inside + 32 outside) samples of the same scene, torch-CPU oracle (oracle/neus_ref.py), %.1f s" % (n, dt)}


def _c3_cpu_baseline(threads, size=128):
    """The oracle's stage-2 training step (render with edge sampling under autograd + backward, oracle/train_ref.py) on a bounded
    sample: the same scene at size x size (a 512x512 step is (512/size)^2 times the rays)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from iron_amd import scenes
    from oracle import iron_ref as R
    from oracle import train_ref as T
    from _util import cpu_sd, tables
    torch.set_num_threads(threads)
    names = ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")
    nets = scenes.build_networks("S1")
    sd = {k: T.leaf_state(cpu_sd(nets[k])) for k in names}
    mt, md = tables()
    sc = R.Scene(sd["sdf_network"], R.SDFSpec(), {k: (sd[k], R.GGX_SPECS[k]) for k in R.GGX_SPECS}, 32.0, mt, md)
    K, W2C = scenes.fixture_camera_matrices(size, size)
    cam = R.CameraSpec(size, size, K, W2C)
    t0 = time.perf_counter()
    with torch.no_grad():
        tr = R.raytrace_camera(sc, cam, max_num_rays=50000)
        dem = (R.sobel_magnitude(tr["depth"]) > 1e-2) & tr["convergent_mask"]
    res = T.render_camera_edges_train(sc, cam, dem)
    target = torch.rand(size, size, 3, generator=torch.Generator().manual_seed(1))
    m = res["convergent_mask"] | res["edge_mask"]
    (res["color"][m] - target[m]).abs().mean().backward()
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "one training step (trace incl. the mask pre-pass, render with edge sampling under autograd, image loss, backward) of the same "
                      "scene at %dx%d = 1/%d of the rays of the 512x512 step, torch-CPU oracle (oracle/train_ref.py), %.1f s; no eikonal term, no "
                      "optimiser" % (size, size, (512 // size) ** 2, dt),
            "rays_fraction_of_workload": (size / 512.0) ** 2}


def _recorded_traffic(key):
    """HBM bytes per launch of a dominant kernel from the committed PMC passes (tools/pmc_r02.sh -> profiles/hbm_traffic.json), or None."""
    tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        e = json.load(open(tp)).get(key)
    except OSError:
        return None, None
    if not e:
        return None, None
    return e["bytes_per_launch"], ("profiles/hbm_traffic.json [%s]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                   "(gfx950 corrections applied), NOT re-measured in this run; %s" % (key, e.get("source", "")))


def _gemm_roofline():
    """HBM roofline of the backward's dominant kernel, measured here: the layer product Z = X W^T at the shape of one SDF chunk
    (131 072 rows = 65 536 points x {value, tangent}, 256 -> 256) through iron_train_gemm on the current stream, hipEvents around
    20 launches.  Algorithmic bytes per launch = A once in + C once out (the 256 KB weight matrix is L2-resident)."""
    from iron_amd import _lib
    lib = _lib.load_train()
    R_, K_, N_ = 131072, 256, 256
    dev = torch.device("cuda", 0)
    X = torch.randn(R_, K_, device=dev)
    W = torch.randn(N_, K_, device=dev) / 16.0
    out = torch.empty(R_, N_, device=dev)
    nbytes = lib.iron_train_gemm_workspace_bytes(0, R_, N_)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def launch():
        rc = lib.iron_train_gemm(0, 1, R_, N_, K_, X.data_ptr(), K_, W.data_ptr(), K_, 0.0, out.data_ptr(), N_, ws.data_ptr(), nbytes, _lib.stream_ptr(dev))
        assert rc == 0, rc
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record()
    e1.synchronize()
    sec = e0.elapsed_time(e1) / 20 / 1e3
    bytes_alg = 4.0 * R_ * (K_ + N_)
    flop = 2.0 * R_ * K_ * N_
    traffic, traffic_src = _recorded_traffic("c3_gemm_rows_probe")
    return {"bound": "hbm", "kernel": "k_gemm_rows<2> (+ k_gemm_pack_b): layer product Z = X W^T, 131072 x 256 x 256, of iron_sdf_backward", "achieved": bytes_alg / sec / 1e9,
            "peak": 8000.0, "unit": "GB/s", "frac": bytes_alg / sec / 8e12, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": bytes_alg, "avg_launch_ms": sec * 1e3, "tflops_fp32_equivalent": flop / sec / 1e12,
            "note": "bytes = 4 B x rows x (K + N): A read once, C written once; 43 FLOP per byte at K = N = 256 is below the split-fp16 "
                    "ridge (833 TFLOP/s / 8 TB/s = 104), so HBM is the bound of a layer-wise backward"}


def secondary_workload(a):
    """BASELINE configs C2 / C3 through the same entry point (tools/neus_frames.py, tools/train_step.py do the work)."""
    if a.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--workload %s is a single-GPU configuration" % a.workload)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from iron_amd import _lib
    _lib.load()
    base = {"n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "roofline": None, "cpu_baseline": None}
    threads = _host_threads()
    if a.workload == "c2":
        import neus_frames
        def start_profile():
            _lib.profile_enable(True)
            _lib.profile_read()
        # one untimed frame of the same a.steps batches on the SAME networks first (the weight packs, their device buffers and the
        # call workspaces are made on first use: timed, they showed up as 10 or 16 ms per batch depending on the allocator's mood)
        r = neus_frames.run(rays=4096 * a.steps, batches=(4096,), repeats=1, warm=True, before_timed=start_profile)[0]
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        kernels = {k: {"ms_total": ms, "launches": n, "ms_avg": ms / n} for k, (ms, n) in prof.items() if n}
        roof = None
        if "sdf_grad" in kernels:   # get_all at the 128 section mid points of every ray: the dominant kernel of a NeuS batch
            pts = 4096 * 128
            sec = kernels["sdf_grad"]["ms_avg"] / 1e3
            ach = FLOP_SDF_GRAD * pts / sec
            traffic, traffic_src = _recorded_traffic("c2_sdf_grad")
            roof = {"bound": "mfma", "kernel": "k_sdf_getall_rev_h2 (get_all: value + 256 features + reverse-mode gradient at 4096 x 128 points)",
                    "achieved": ach / 1e12, "peak": PEAK_F16_MFMA / 3.0 / 1e12, "unit": "TFLOP/s", "frac": ach / (PEAK_F16_MFMA / 3.0), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "flop_per_unit": FLOP_SDF_GRAD, "units_per_launch": pts, "avg_launch_ms": kernels["sdf_grad"]["ms_avg"],
                    "peak_basis": "dense f16 MFMA 2500 TFLOP/s / 3 products per fp32-accurate MAC",
                    "note": "algorithmic FLOP per point = 2 x (524 544 forward + 459 008 input-gradient) MAC (SURVEY 8d); the kernel is a forward "
                            "pass plus one reverse sweep over transposed weight slots (140 ring steps per 128 points), the sigma' tape of 8 KiB "
                            "per point written to and read back from HBM (not algorithmic bytes: it is the tape autograd keeps in the reference)"}
        base.update({"metric": "krays/s stage-1 NeuS volume render (models/renderer.py), 4096 rays x 128 samples hierarchical",
                     "value": r["krays_per_s"], "unit": "krays/s", "ms_per_step": round(r["ms"] / a.steps, 3),
                     "config": {"workload": "C2: seeded networks of confs/womask_iron.conf (8x256 SDF, 8-layer PE-10 colour net, NeRF "
                                            "background with 32 outside samples), 4096 rays per step, 64 + 4x16 samples, perturb = 0",
                                "mlp_points_per_ray": r["mlp_points_per_ray"]},
                     "roofline": roof, "kernels": kernels})
        if not a.no_cpu_baseline:
            base["cpu_baseline"] = _c2_cpu_baseline(threads)
    else:
        import train_step
        _lib.load_train()
        r = train_step.run(size=512, steps=a.steps, warmup=max(a.warmup, 1))
        base.update({"metric": "training steps/s, stage-2 with backward (autograd through the HIP operators + edge-sample silhouette "
                               "gradient), 512x512", "value": r["steps_per_s"], "unit": "steps/s", "ms_per_step": r["ms_step"],
                     "config": {"workload": "C3: scene S1, render_camera(handle_edges=True, is_training=True), L1 image loss + eikonal "
                                            "term on %d points, backward, Adam on all networks" % r["eikonal_points"],
                                "hits": r["hits"], "edge_pixels": r["edge_pixels"], "ms_forward_render": r["ms_forward_render"],
                                "ms_loss_backward": r["ms_loss_backward"], "ms_adam": r["ms_adam"]},
                     "roofline": _gemm_roofline()})
        if not a.no_cpu_baseline:
            base["cpu_baseline"] = _c3_cpu_baseline(threads)
    print(json.dumps(base))


def cpu_baseline(scene: str, res: int):
    """The oracle (torch-CPU port of the reference path) timed on this host's cores: config C0 -- the 64x64 crop,
    ul = (224, 224), of the 512x512 fixture camera -- always, plus a bounded full view (`res` x `res`, default 160: ~5 s) of the same
    scene, which is the figure `value` quotes because its ray mix (45.9 evaluations per ray, 46 % hits) is the frame's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import iron_ref as R
    from iron_amd import scenes
    from _util import oracle_scene
    # every core this job may use (_cpu_share: affinity cut to the cgroup quota); IRON_CPU_THREADS overrides
    share, host = _cpu_share()
    threads = max(1, int(os.environ.get("IRON_CPU_THREADS", share)))
    torch.set_num_threads(threads)
    print("[bench] cpu_baseline: oracle on %d threads (%s)" % (threads, host), file=sys.stderr, flush=True)
    sc = oracle_scene(scenes.build_networks(scene))
    K512, W2C512 = scenes.fixture_camera_matrices(512, 512)
    c0 = R.CameraSpec(512, 512, K512, W2C512).crop(64, 64, (224, 224))
    R.render_camera(sc, R.CameraSpec(16, 16, *scenes.fixture_camera_matrices(16, 16)))  # warm the thread pool
    t0 = time.perf_counter()
    R.render_camera(sc, c0)
    dt0 = time.perf_counter() - t0
    K, W2C = scenes.fixture_camera_matrices(res, res)
    cam = R.CameraSpec(res, res, K, W2C)
    sc.counter.evals = 0
    t0 = time.perf_counter()
    out = R.render_camera(sc, cam)
    dt = time.perf_counter() - t0
    return {"value": res * res / dt / 1e6, "unit": "Mrays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%s %dx%d full view (fixture camera rescaled), trace+GGX shade, torch-CPU oracle, %.1f s" % (scene, res, res, dt),
            "E_per_ray": sc.counter.evals / (res * res), "H_per_ray": float(out["convergent_mask"].float().mean()),
            "c0": {"value": 64 * 64 / dt0 / 1e6, "unit": "Mrays/s", "seconds": dt0,
                   "sample": "BASELINE config C0: 64x64 crop, ul = (224, 224), of the 512x512 fixture camera (all rays hit: 8 evaluations per ray)"},
            "host": host}


def predicted_strong_scaling(cam, sdf, nets, fn, tracer_factory, frame_ms, steps=3):
    """One-GPU PREDICTION for the multi-GPU target (no collectives in it): for world = 2, 4, 8 every rank's tile shard of the frame is
    run on this card the way the rank runs it -- its rank-local phases of ShardedRenderer (trace_begin -> trace_finish -> shade) back
    to back, `steps` steps without a host sync in between -- and T(frame) / max_r T(rank r step) is the strong-scaling factor the
    kernels and the host glue allow.  What an N-GPU step adds on top: the MAX all-reduce of 13 ints, the gather of 26 floats per
    pixel to rank 0 and rank 0's un-tile (`assemble_ms`, measured here; factor_with_assemble includes it)."""
    from iron_amd.sharding import ShardedRenderer, render_emulated
    out = {"frame_ms": frame_ms, "tile": SHARD_TILE,
           "note": "PREDICTION from one card: T(frame) / max over ranks of the rank's steady-state step (rank-local phases back to back, "
                   "no collectives); factor_with_assemble adds rank 0's un-tile"}
    for world in (2, 4, 8):
        ms = []
        shs = [ShardedRenderer(sdf, nets, tracer_factory(), fn, tile=SHARD_TILE, chunk=50000, world=world, rank=r) for r in range(world)]
        # the chunk-global bisection counts after the MAX all-reduce: every rank runs its second tracer phase up to the SAME counts
        # (one rank's own table would let seven of eight ranks off ~0.4 ms of dependent evaluations)
        table = None
        for sh in shs:
            st = sh.trace_begin([cam])
            table = st["chunk_iters"].clone() if table is None else torch.maximum(table, st["chunk_iters"])
            sh.trace_finish(st)
        for r in range(world):
            sh = shs[r]

            def step():
                st = sh.trace_begin([cam])
                st["chunk_iters"].copy_(table)
                return sh.shade(sh.trace_finish(st))
            for _ in range(4):
                step()
            best = None
            for _ in range(5):   # best of 5 runs of `steps` steps: a caching-allocator stall (new 256 MiB blocks) inflates single runs 2x
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    step()
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / steps * 1e3
                best = t if best is None else min(best, t)
            ms.append(best)
        asm = min(render_emulated(world, [cam], sdf, nets, fn, tracer_factory, tile=SHARD_TILE)[2] for _ in range(2))
        out["n%d" % world] = {"rank_step_ms": [round(x, 3) for x in ms], "max_rank_step_ms": max(ms), "mean_rank_step_ms": sum(ms) / world,
                              "assemble_ms": asm, "factor": frame_ms / max(ms), "factor_with_assemble": frame_ms / (max(ms) + asm)}
    return out


def f32_core_child(a):
    """The same workload on the exact-fp32 MFMA core (IRON_MLP_CORE is read once per process: a child process)."""
    import subprocess
    env = dict(os.environ, IRON_MLP_CORE="f32")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--res", str(a.res), "--scene", a.scene,
           "--no-cpu-baseline", "--no-extras"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "roofline": d["roofline"]}
    except Exception as e:  # the headline line must not depend on the extra
        return {"error": repr(e)}


def main():
    a = parse()
    if a.workload != "c1":
        return secondary_workload(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.rendezvous_only:
        if world < 2:
            raise SystemExit("--rendezvous-only is the N>1 launch check")
        return rendezvous_only(a, world, rank)
    import torch.distributed as dist
    # IRON_BENCH_BACKEND=gloo + IRON_BENCH_ONE_GPU=1: rehearse the N>1 path with all ranks on one card
    backend = os.environ.get("IRON_BENCH_BACKEND", "nccl")
    one_gpu = os.environ.get("IRON_BENCH_ONE_GPU", "0") == "1"
    dev_index = 0 if (world == 1 or one_gpu) else local_rank
    if world > 1:
        _reserve_stdout()
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    from iron_amd import _lib, scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    from iron_amd.sharding import ShardedRenderer
    import iron_amd.raytracer as rt

    _lib.load()  # fail loudly if the HIP library is missing
    nets = {k: v.to(dev) for k, v in scenes.build_networks(a.scene).items()}
    sdf = nets["sdf_network"]
    tracer = RayTracer()
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    n_views = world if a.scaling == "weak" else 1
    cams = []
    for v in range(world):
        K, W2C = scenes.fixture_camera_matrices(a.res, a.res, yaw_deg=45.0 * v)
        cams.append(Camera(a.res, a.res, K.to(dev), W2C.to(dev)))
    all_cams, cams = cams, cams[:n_views]
    sharded = ShardedRenderer(sdf, nets, tracer, fn, tile=SHARD_TILE, chunk=50000) if world > 1 else None

    def step(stats=False):
        if world == 1:
            rt.VERBOSE_MODE = stats
            try:
                return render_camera(cams[0], sdf, tracer, nets, fn, fill_holes=a.edges and not stats, handle_edges=a.edges and not stats,
                                     is_training=False)  # the work counters are read on the plain trace
            finally:
                rt.VERBOSE_MODE = False
        return sharded.render(cams, collect_stats=stats)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if rank == 0:
        print("[bench] warmup done", file=sys.stderr, flush=True)
    _lib.profile_enable(True)
    _lib.profile_read()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    if rank == 0:
        print("[bench] timed region: %.3f s for %d steps" % (dt, a.steps), file=sys.stderr, flush=True)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # the other scaling mode, same run, same protocol (outside the headline's timed region)
    other = None
    if world > 1 and not a.no_extras:
        o_cams = all_cams[:1] if a.scaling == "weak" else all_cams
        for _ in range(max(1, a.warmup)):
            sharded.render(o_cams)
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            sharded.render(o_cams)
        torch.cuda.synchronize()
        barrier()
        odt = time.perf_counter() - t1
        tmax = torch.tensor([odt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        odt = float(tmax.item())
        other = {"scaling": "strong" if a.scaling == "weak" else "weak", "views_per_step": len(o_cams),
                 "value": len(o_cams) * a.res * a.res * a.steps / odt / 1e6, "unit": "Mrays/s", "ms_per_step": odt / a.steps * 1e3,
                 "steps": a.steps}

    # one extra (untimed) step for the work counters
    res = step(stats=True)
    torch.cuda.synchronize()
    st = tracer.last_stats
    cnt = torch.tensor([st["n_evals_ref"], st["n_evals"], st["n_conv"], st["n_evals_sphere"], st["n_sampler"], st["n_bisect"]],
                       dtype=torch.int64, device=coll_dev)
    if world > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    E_ref, E_hip, H, E_sphere, n_sampler, n_bisect = [int(x) for x in cnt.tolist()]

    if rank == 0:
        rays = n_views * a.res * a.res
        ms_per_step = dt / a.steps * 1e3
        value = rays * a.steps / dt / 1e6
        # oracle-counted work for this exact workload, if committed (tools/count_work.py)
        E_oracle = H_oracle = None
        wc_path = os.path.join(ROOT, "tests", "golden", "work_counts.json")
        if world == 1 and os.path.exists(wc_path):
            wc = json.load(open(wc_path)).get("%s_%d" % (a.scene, a.res))
            if wc:
                E_oracle, H_oracle = wc["E"], wc["H"]
        E_alg = E_oracle if E_oracle is not None else E_ref
        H_alg = H_oracle if H_oracle is not None else H
        flop_frame = FLOP_PER_EVAL * E_alg + FLOP_PER_HIT * H_alg
        # per-kernel device time (rank 0's launches, hipEvents on the launch stream inside the timed region)
        alg = {"sphere": FLOP_PER_EVAL * E_sphere / world, "sampler": FLOP_PER_EVAL * n_sampler * tracer.n_steps / world,
               "bisect_a": None, "bisect_b": None, "sdf_grad": FLOP_SDF_GRAD * H / world, "material": FLOP_MATERIALS * H / world / 3.0,
               "ggx": None, "sdf_forward": None}
        kernels = {}
        for k, (ms, n) in prof.items():
            if n:
                kernels[k] = {"ms_total": ms, "launches": n, "ms_avg": ms / n}
        dom = max(kernels, key=lambda k: kernels[k]["ms_total"]) if kernels else None
        roof = None
        if a.edges:
            dom = None  # per-kernel averages mix the full-image launches with the small edge-ray launches: no roofline line
        if dom and alg.get(dom):
            avg_s = kernels[dom]["ms_avg"] / 1e3
            traffic = None
            tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tp):
                traffic = json.load(open(tp)).get(dom, {}).get("bytes_per_launch")
            core = os.environ.get("IRON_MLP_CORE", "h2")
            h2 = core.startswith("h")
            # units one launch processes = the SDF evaluations the kernel EXECUTES (the sampler stops at a ray's first
            # negative block, so it executes fewer than the reference's 128 per ray; those are NOT credited here)
            exec_evals = {"sphere": E_sphere, "sampler": E_hip - E_sphere - n_bisect * 9}.get(dom)
            if exec_evals is not None:
                flop_launch = FLOP_PER_EVAL * exec_evals / world
            else:
                flop_launch = alg[dom]
            ach = flop_launch / avg_s
            # fp32-accurate MACs on the f16 pipe cost three MFMA products each: the pipe's ceiling for this arithmetic is
            # its dense f16 peak / 3.  The exact-fp32 core (IRON_MLP_CORE=f32) is priced against the fp32 MFMA peak.
            peak = PEAK_F16_MFMA / 3.0 if h2 else PEAK_FP32_MFMA
            executed = None
            if exec_evals is not None:
                per = MFMA_FLOP_PER_EVAL_H2 if h2 else MFMA_FLOP_PER_EVAL_F32
                pk = PEAK_F16_MFMA if h2 else PEAK_FP32_MFMA
                ex = per * exec_evals / world / avg_s
                executed = {"mfma_tflops": ex / 1e12, "pipe": "f16 (fp32 = 2 x fp16 split, 3 products)" if h2 else "f32",
                            "pipe_peak": pk / 1e12, "frac_of_pipe_peak": ex / pk}
            roof = {"bound": "mfma", "kernel": dom, "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                    "frac": ach / peak, "traffic": traffic,
                    "traffic_source": ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                       "(gfx950 corrections applied), NOT re-measured in this run") if traffic is not None else None,
                    "core": core,
                    "peak_basis": ("dense f16 MFMA 2500 TFLOP/s / 3 products per fp32-accurate MAC" if h2
                                   else "dense fp32 MFMA 157.3 TFLOP/s"),
                    "vs_fp32_mfma_peak": ach / PEAK_FP32_MFMA, "executed": executed,
                    "algorithmic_flop_per_launch": flop_launch, "units_per_launch": exec_evals / world if exec_evals is not None else None,
                    "flop_per_unit": FLOP_PER_EVAL, "avg_launch_ms": kernels[dom]["ms_avg"],
                    "reference_equivalent_tflops": alg[dom] / avg_s / 1e12,
                    "note": "achieved = 918016 FLOP x SDF evaluations the kernel executes / launch time (hipEvents on the "
                            "launch stream); reference_equivalent also credits the evaluations the reference makes and "
                            "the kernel's early exit skips"}
        out = {
            "metric": "Mrays/s sphere-trace+GGX shade, drv/dragon 800x800 (synthetic S0)", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": a.scaling if world > 1 else "weak", "other_scaling": other, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ranks": {"world_size_seen": dist.get_world_size() if world > 1 else 1, "backend": backend if world > 1 else None,
                      "launcher": (os.environ.get("IRON_BENCH_LAUNCHER", "torchrun") if world > 1 else None),
                      "all_ranks_on_one_gpu": bool(one_gpu and world > 1)},
            "mlp_core": os.environ.get("IRON_MLP_CORE", "h2") + (" (fp32-accurate split-fp16 MFMA, LDS weight ring)" if not os.environ.get("IRON_MLP_CORE", "h2").startswith("f") else " (exact fp32 MFMA)"),
            "config": {"workload": "C1: scene %s (seeded geometric-init SDF 8x256 + ggx material nets), %dx%d full image, "
                                   "sphere-trace + GGX shade, fp32%s" % (a.scene, a.res, a.res, " + hole filling + silhouette edge sampling" if a.edges else ""),
                       "views_per_step": n_views, "rays_per_step": rays,
                       "sharding": "none (render_camera)" if world == 1 else "interleaved %dx%d tiles over %d ranks, RCCL gather" % (SHARD_TILE, SHARD_TILE, world),
                       "tracer": {"sdf_threshold": 5e-5, "sphere_tracing_iters": 16, "n_steps": 128, "chunk": 50000}},
            "roofline": roof,
            "frame": {"E_reference_evals": E_alg, "E_source": "oracle (tests/golden/work_counts.json)" if E_oracle is not None else "device counters",
                      "E_device_ref_equivalent": E_ref, "E_executed": E_hip, "H_hits": H_alg, "H_device": H,
                      "algorithmic_flop_per_step": flop_frame, "achieved_tflops": flop_frame * a.steps / dt / 1e12,
                      "frac_of_fp32_mfma_peak": flop_frame * a.steps / dt / PEAK_FP32_MFMA,
                      "executed_tflops": (FLOP_PER_EVAL * E_hip + FLOP_PER_HIT * H) * a.steps / dt / 1e12},
            "kernels": kernels,
        }
        from iron_amd import build as _build
        man = _build.manifest()
        out["build"] = {"hipcc": man.get("hipcc"), "any_fallback_flags": man.get("any_fallback"),
                        "units": {k: (" ".join(f for f in v["flags"] if not f.startswith("-W")) + (" [FALLBACK %d]" % v["attempt"] if v["fallback"] else ""))
                                  for k, v in man.get("units", {}).items() if k in ("trace.hip", "shade.hip", "h2_kernels.hip")}}
        if world == 1 and not a.no_extras and not a.edges:
            out["predicted_strong_scaling"] = predicted_strong_scaling(cams[0], sdf, nets, fn, RayTracer, ms_per_step)
            out["f32_core"] = f32_core_child(a)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.scene, a.cpu_sample_res)
        else:
            out["cpu_baseline"] = None
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
