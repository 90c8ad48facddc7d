"""Multi-GPU render: rays of a render call sharded by interleaved image tiles, one process per GPU
(torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no distributed code; rays are independent, so the only exchanges are
  1. one MAX all-reduce of the per-chunk bisection counts between the two tracer phases -- the
     reference's rootfind loop count is global to a 50 000-ray chunk (models/raytracer.py:204-217)
     and a chunk's rays live on several ranks, and
  2. one gather of the finished per-pixel buffers to rank 0, followed by a local un-tile permutation.
Tiles are dealt round-robin (tile_id % world) because the cost per ray varies ~15x and is spatially
coherent; weights (5.4 MB) are replicated.  Default tile: 8x8 pixels -- rays are independent and every
kernel works from ray lists, so nothing is gained from larger tiles, and at 800x800 over 8 ranks 32x32 tiles leave the
slowest rank 8 % above the mean where 8x8 tiles leave 0.2 % (tools/shard_step_time.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

# packed per-pixel record gathered to rank 0 (floats)
RECORD = (("convergent_mask", 1), ("depth", 1), ("distance", 1), ("sdf", 1), ("points", 3), ("color", 3),
          ("diffuse_color", 3), ("specular_color", 3), ("diffuse_albedo", 3), ("specular_albedo", 3),
          ("specular_roughness", 1), ("normal", 3))
RECORD_WIDTH = sum(w for _, w in RECORD)


def tile_pixels(H: int, W: int, tile: int, world: int, rank: int) -> torch.Tensor:
    """Flat pixel indices (row-major, int64) of the tiles owned by `rank`: tiles are numbered row-major and
    dealt on a skewed lattice (below); pixels inside a tile stay row-major."""
    ty, tx = (H + tile - 1) // tile, (W + tile - 1) // tile
    ids = torch.arange(ty * tx, dtype=torch.int64)
    # owner of tile (row, col) = (col + skew * row) % world: a diagonal lattice with `world` distinct row phases.  Plain tile_id % world
    # degenerates when the tiles per row share a factor with `world` (100 tiles per row over 8 ranks: only two row phases, i.e. vertical
    # stripes that line up with the silhouette -- one rank 6 % above the others at 800 x 800)
    mine = ids[((ids % tx) + _skew(world) * (ids // tx)) % world == rank]
    r0 = (mine // tx) * tile
    c0 = (mine % tx) * tile
    dy = torch.arange(tile, dtype=torch.int64).view(1, tile, 1)
    dx = torch.arange(tile, dtype=torch.int64).view(1, 1, tile)
    rows = r0.view(-1, 1, 1) + dy
    cols = c0.view(-1, 1, 1) + dx
    ok = (rows < H) & (cols < W)
    return (rows * W + cols)[ok]


def _skew(world: int) -> int:
    """Row-to-row shift of the tile lattice: the integer nearest 0.382 x world (golden-section spacing) that is coprime to `world`."""
    import math
    s = max(1, round(0.382 * world))
    while math.gcd(s, world) != 1:
        s += 1
    return s


def shard_sizes(H: int, W: int, tile: int, world: int) -> List[int]:
    return [int(tile_pixels(H, W, tile, world, r).numel()) for r in range(world)]


def chunks_per_view(H: int, W: int, chunk: int) -> int:
    return (H * W + chunk - 1) // chunk


def global_ray_index(pix: torch.Tensor, view: int, H: int, W: int, chunk: int) -> torch.Tensor:
    """Position of a ray in the concatenated job such that ray_index // chunk is unique per (view, chunk):
    views are spaced by a whole number of chunks so chunks never straddle two views."""
    return pix + view * chunks_per_view(H, W, chunk) * chunk


def _via_host(t: torch.Tensor, group) -> bool:
    """gloo has no device collectives for every op: stage device tensors through the host for it (tests /
    rehearsals on one GPU); RCCL ("nccl") works on device buffers directly."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def reduce_chunk_iters(chunk_iters: torch.Tensor, group=None) -> torch.Tensor:
    """MAX all-reduce of the per-chunk bisection counts (exchange 1), in place."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if _via_host(chunk_iters, group):
            h = chunk_iters.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
            chunk_iters.copy_(h)
        else:
            dist.all_reduce(chunk_iters, op=dist.ReduceOp.MAX, group=group)
    return chunk_iters


def gather_records(local: torch.Tensor, sizes: Sequence[int], group=None, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather per-rank [n_r, C] record buffers on `dst` (exchange 2).  Ranks may own different pixel counts, so
    buffers are padded to the largest shard for the collective and cut back afterwards."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [local]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_max = max(sizes)
    dev = local.device
    if _via_host(local, group):
        local = local.cpu()
    buf = local
    if local.shape[0] != n_max:
        buf = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf[: local.shape[0]] = local
    buf = buf.contiguous()
    if rank == dst:
        outs = [torch.empty_like(buf) for _ in range(world)]
        dist.gather(buf, outs, dst=dst, group=group)
        return [o[: sizes[r]].to(dev) for r, o in enumerate(outs)]
    dist.gather(buf, None, dst=dst, group=group)
    return None


def assemble_views(parts: Sequence[torch.Tensor], pix_lists: Sequence[torch.Tensor], n_views: int, H: int, W: int) -> torch.Tensor:
    """Un-tile: parts[r] is rank r's [n_views * n_r, C] records (view-major), pix_lists[r] its pixel indices.
    Returns [n_views, H, W, C]."""
    Cw = parts[0].shape[1]
    out = torch.zeros((n_views, H * W, Cw), dtype=parts[0].dtype, device=parts[0].device)
    for part, pix in zip(parts, pix_lists):
        n_r = pix.numel()
        p = pix.to(part.device)
        out[:, p] = part.reshape(n_views, n_r, Cw)
    return out.reshape(n_views, H, W, Cw)


def split_record(img: torch.Tensor, layout=None) -> Dict[str, torch.Tensor]:
    """[..., width] -> result dict with the reference's keys / dtypes (layout: RECORD unless given)."""
    out, o = {}, 0
    for k, w in (layout or RECORD):
        v = img[..., o:o + w]
        o += w
        out[k] = (v[..., 0] > 0.5) if k == "convergent_mask" else (v[..., 0] if w == 1 else v)
    return out


def all_gather_records(local: torch.Tensor, sizes: Sequence[int], group=None) -> List[torch.Tensor]:
    """Every rank receives every rank's [n_r, C] buffer (padded to the largest shard for the collective)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [local]
    world = dist.get_world_size(group)
    n_max = max(sizes)
    dev = local.device
    if _via_host(local, group):
        local = local.cpu()
    buf = local
    if local.shape[0] != n_max:
        buf = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf[: local.shape[0]] = local
    buf = buf.contiguous()
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return [o[: sizes[r]].to(dev) for r, o in enumerate(outs)]


# what the fill_holes pass needs of the whole image from every rank (floats per pixel)
TRACE_RECORD = (("convergent_mask", 1), ("depth", 1), ("distance", 1), ("sdf", 1), ("points", 3))
TRACE_WIDTH = sum(w for _, w in TRACE_RECORD)
SHADE_KEYS = ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo", "specular_roughness", "normal")


def _pack(res: Dict[str, torch.Tensor], layout, rows: int) -> torch.Tensor:
    cols = []
    for k, w in layout:
        v = res[k]
        cols.append((v.float() if v.dtype == torch.bool else v).reshape(rows, w))
    return torch.cat(cols, dim=1).contiguous()


class ShardedRenderer:
    """render_camera(is_training=False) for a batch of views with the rays of every view tile-sharded over the ranks of
    `group`; the result dicts arrive on rank 0.  Exchanges per call: the MAX all-reduce of the chunk bisection counts, one
    gather of the finished pixel records, and -- with fill_holes only -- one all-gather of the 7-float trace records,
    because the reference's hole filling (raytracer.py:554-564) is a whole-image pass that rewrites distance and points of
    EVERY pixel before shading: each rank applies it to the assembled image and shades its own tiles from the result.
    Silhouette edge sampling (handle_edges: sobel, surface walk, side rays: a few thousand rays) runs on rank 0 after the
    gather.

    `world` / `rank` override the process group's: shard emulation on one card (tests/test_gpu_shards.py, tools/
    shard_scaling.py drive the phase methods of `world` instances by hand and do the exchanges in memory)."""

    def __init__(self, sdf_network, color_network_dict, raytracer, render_fn, tile: int = 8, chunk: int = 50000,
                 group=None, world: Optional[int] = None, rank: Optional[int] = None):
        self.sdf_network = sdf_network
        self.nets = color_network_dict
        self.tracer = raytracer
        self.render_fn = render_fn
        self.tile = tile
        self.chunk = chunk
        self.group = group
        if world is not None:
            self.world, self.rank = int(world), int(rank or 0)
        else:
            self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if self.world > 1 else 0
        self._pix_cache = {}
        self.last_stats = None

    def _pix(self, H, W, device):
        key = (H, W, str(device))
        if key not in self._pix_cache:
            # every rank's pixel list, resident on the device: assemble_views indexes with them every frame (a host list would be
            # one synchronous host-to-device copy per rank and frame on rank 0)
            lists = [tile_pixels(H, W, self.tile, self.world, r).to(device) for r in range(self.world)]
            mine = lists[self.rank]
            uv = torch.stack(((mine % W).float() + 0.5, (mine // W).float() + 0.5), dim=-1)  # raytracer.py:300-303
            self._pix_cache[key] = (lists, mine, uv)
        return self._pix_cache[key]

    # ---- phases (each is local to the rank; the exchanges sit between them) -------------------------------------------
    @torch.no_grad()
    def trace_begin(self, cameras, collect_stats: bool = False):
        """This rank's rays of every view, sphere intersection, tracer phase 0.  state["chunk_iters"] is the table to
        MAX-reduce over the ranks before trace_finish."""
        from .raytracer import SDFHandle, intersect_sphere
        H, W = cameras[0].H, cameras[0].W
        dev = cameras[0].device
        lists, mine, uv = self._pix(H, W, dev)
        V = len(cameras)
        o_parts, d_parts, nrm_parts, idx_parts = [], [], [], []
        for v, cam in enumerate(cameras):
            ro, rd, rn = cam.get_rays(uv)
            o_parts.append(ro); d_parts.append(rd); nrm_parts.append(rn)
            idx_parts.append(global_ray_index(mine, v, H, W, self.chunk))
        ray_o, ray_d, ray_n = torch.cat(o_parts), torch.cat(d_parts), torch.cat(nrm_parts)
        ray_index = torch.cat(idx_parts).contiguous()
        hit, near, far = intersect_sphere(ray_o, ray_d, 1.0)
        n_chunks = V * chunks_per_view(H, W, self.chunk)
        st = self.tracer.phase_begin(SDFHandle(self.sdf_network), ray_o, ray_d, near, far, hit, ray_index, n_chunks, self.chunk,
                                     collect_stats)
        st.update({"cameras": cameras, "ray_o": ray_o, "ray_d": ray_d, "ray_n": ray_n, "V": V, "H": H, "W": W, "lists": lists,
                   "mine": mine, "n_r": int(mine.numel())})
        return st

    @torch.no_grad()
    def trace_finish(self, st):
        res = self.tracer.phase_finish(st)
        self.last_stats = self.tracer.last_stats
        res["depth"] = res["distance"] / st["ray_n"] * res["convergent_mask"].float()  # raytracer.py:393,552
        res.update({"ray_o": st["ray_o"], "ray_d": st["ray_d"]})
        st["res"] = res
        return st

    def trace_records(self, st) -> torch.Tensor:
        return _pack(st["res"], TRACE_RECORD, st["V"] * st["n_r"])

    @torch.no_grad()
    def fill_holes(self, st, parts: Sequence[torch.Tensor]):
        """parts[r] = rank r's trace records (all ranks, view-major).  Applies the reference's hole filling to each whole
        view and takes this rank's pixels back out of the result."""
        from .raytracer import fill_depth_holes
        V, H, W, mine, n_r = st["V"], st["H"], st["W"], st["mine"], st["n_r"]
        img = assemble_views(parts, st["lists"], V, H, W)
        res = st["res"]
        out = {k: [] for k in ("convergent_mask", "depth", "distance", "points")}
        for v, cam in enumerate(st["cameras"]):
            full = split_record(img[v], TRACE_RECORD)
            full = {k: t.contiguous() for k, t in full.items()}
            ro, rd, rn = cam.get_rays(cam.get_uv())
            full.update({"ray_o": ro, "ray_d": rd, "ray_d_norm": rn})
            fill_depth_holes(full)
            for k in out:
                out[k].append(full[k].reshape(H * W, -1)[mine])
        res["convergent_mask"] = torch.cat(out["convergent_mask"]).reshape(-1)
        res["depth"] = torch.cat(out["depth"]).reshape(-1)
        res["distance"] = torch.cat(out["distance"]).reshape(-1)
        res["points"] = torch.cat(out["points"]).reshape(-1, 3)
        return st

    @torch.no_grad()
    def shade(self, st) -> torch.Tensor:
        """Shade this rank's pixels (fused kernels when render_fn is the GGX one) and pack the per-pixel records."""
        from .raytracer import render_normal_and_color
        render_normal_and_color(st["res"], self.sdf_network, self.nets, self.render_fn, is_training=False)
        return _pack(st["res"], RECORD, st["V"] * st["n_r"])

    @torch.no_grad()
    def assemble(self, parts: Sequence[torch.Tensor], cameras, handle_edges: bool = False):
        """Rank 0: un-tile the gathered records; with handle_edges the silhouette pass of render_camera on every view."""
        from .raytracer import locate_silhouette, render_edge_pixels
        H, W = cameras[0].H, cameras[0].W
        lists = self._pix(H, W, cameras[0].device)[0]
        img = assemble_views(parts, lists, len(cameras), H, W)
        out = split_record(img)
        if not handle_edges:
            return out
        out = {k: v.contiguous() for k, v in out.items()}
        extra = {}
        for v, cam in enumerate(cameras):
            view = {k: t[v] for k, t in out.items()}
            uv = cam.get_uv()
            ro, rd, rn = cam.get_rays(uv)
            view.update({"uv": uv, "ray_o": ro, "ray_d": rd, "ray_d_norm": rn})
            locate_silhouette(view, cam, self.sdf_network, max_num_rays=self.chunk)
            edge = view["edge_mask"]
            for k in SHADE_KEYS:  # an edge pixel leaves the convergent mask BEFORE shading (raytracer.py:586): zeros
                view[k][edge] = 0.0
            if int(edge.sum()) > 0:
                render_edge_pixels(view, cam, self.sdf_network, self.tracer, self.nets, self.render_fn, is_training=False)
            for k in view:
                if k not in out:
                    extra.setdefault(k, []).append(view[k])
            out["convergent_mask"][v] = view["convergent_mask"]
        for k, vs in extra.items():
            out[k] = vs if k.startswith("edge_") else torch.stack(vs)
        return out

    @torch.no_grad()
    def render(self, cameras, collect_stats: bool = False, fill_holes: bool = False, handle_edges: bool = False):
        st = self.trace_begin(cameras, collect_stats)
        reduce_chunk_iters(st["chunk_iters"], self.group)
        st = self.trace_finish(st)
        sizes = [st["V"] * int(l.numel()) for l in st["lists"]]
        if fill_holes:
            st = self.fill_holes(st, all_gather_records(self.trace_records(st), sizes, self.group))
        local = self.shade(st)
        parts = gather_records(local, sizes, self.group, dst=0)
        if parts is None:
            return None
        return self.assemble(parts, cameras, handle_edges)


@torch.no_grad()
def render_emulated(world: int, cameras, sdf_network, color_network_dict, render_fn, raytracer_factory, tile: int = 8,
                    chunk: int = 50000, fill_holes: bool = False, handle_edges: bool = False):
    """The sharded render of `world` ranks played through on ONE device: `world` ShardedRenderer instances run their phase
    methods one after the other and the exchanges (MAX of the chunk tables, all-gather, gather) are done in memory -- the
    same code path as ShardedRenderer.render() minus torch.distributed.  Returns (result dict of rank 0, per-rank device
    milliseconds of the rank-local phases): max_r of those is what an N-GPU step costs in kernels, their ratio to the
    unsharded frame the strong-scaling factor load balance allows (tests/test_gpu_shards.py, bench.py)."""
    rs = [ShardedRenderer(sdf_network, color_network_dict, raytracer_factory(), render_fn, tile=tile, chunk=chunk, world=world, rank=r)
          for r in range(world)]
    ms = [0.0] * world
    for r in rs:   # the pixel lists are per-resolution setup (cached by a long-lived renderer), not part of a step
        r._pix(cameras[0].H, cameras[0].W, cameras[0].device)
    torch.cuda.synchronize(cameras[0].device)

    def timed(r, fn, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a)
        e1.record()
        e1.synchronize()
        ms[r] += e0.elapsed_time(e1)
        return out

    sts = [timed(r, rs[r].trace_begin, cameras) for r in range(world)]
    table = sts[0]["chunk_iters"].clone()
    for st in sts[1:]:
        table = torch.maximum(table, st["chunk_iters"])
    for st in sts:
        st["chunk_iters"].copy_(table)
    sts = [timed(r, rs[r].trace_finish, sts[r]) for r in range(world)]
    if fill_holes:
        parts = [rs[r].trace_records(sts[r]) for r in range(world)]
        sts = [timed(r, rs[r].fill_holes, sts[r], parts) for r in range(world)]
    locals_ = [timed(r, rs[r].shade, sts[r]) for r in range(world)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = rs[0].assemble(locals_, cameras, handle_edges)
    e1.record()
    e1.synchronize()
    return out, ms, e0.elapsed_time(e1)
