"""Multi-GPU render: rays of a render call sharded by interleaved image tiles, one process per GPU
(torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no distributed code; rays are independent, so the only exchanges are
  1. one MAX all-reduce of the per-chunk bisection counts between the two tracer phases -- the
     reference's rootfind loop count is global to a 50 000-ray chunk (models/raytracer.py:204-217)
     and a chunk's rays live on several ranks, and
  2. one gather of the finished per-pixel buffers to rank 0, followed by a local un-tile permutation.
Tiles are dealt round-robin (tile_id % world) because the cost per ray varies ~15x and is spatially
coherent; weights (5.4 MB) are replicated.
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
    dealt round-robin; pixels inside a tile stay row-major."""
    ty, tx = (H + tile - 1) // tile, (W + tile - 1) // tile
    ids = torch.arange(ty * tx, dtype=torch.int64)
    mine = ids[ids % world == rank]
    r0 = (mine // tx) * tile
    c0 = (mine % tx) * tile
    dy = torch.arange(tile, dtype=torch.int64).view(1, tile, 1)
    dx = torch.arange(tile, dtype=torch.int64).view(1, 1, tile)
    rows = r0.view(-1, 1, 1) + dy
    cols = c0.view(-1, 1, 1) + dx
    ok = (rows < H) & (cols < W)
    return (rows * W + cols)[ok]


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


def split_record(img: torch.Tensor) -> Dict[str, torch.Tensor]:
    """[..., RECORD_WIDTH] -> result dict with the reference's keys / dtypes."""
    out, o = {}, 0
    for k, w in RECORD:
        v = img[..., o:o + w]
        o += w
        out[k] = (v[..., 0] > 0.5) if k == "convergent_mask" else (v[..., 0] if w == 1 else v)
    return out


class ShardedRenderer:
    """render_camera (fill_holes=False, handle_edges=False, is_training=False) for a batch of views, with the
    rays of every view tile-sharded over the ranks of `group`.  The result dicts arrive on rank 0."""

    def __init__(self, sdf_network, color_network_dict, raytracer, render_fn, tile: int = 32, chunk: int = 50000,
                 group=None):
        self.sdf_network = sdf_network
        self.nets = color_network_dict
        self.tracer = raytracer
        self.render_fn = render_fn
        self.tile = tile
        self.chunk = chunk
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self._pix_cache = {}
        self.last_stats = None

    def _pix(self, H, W, device):
        key = (H, W, str(device))
        if key not in self._pix_cache:
            lists = [tile_pixels(H, W, self.tile, self.world, r) for r in range(self.world)]
            mine = lists[self.rank].to(device)
            uv = torch.stack(((mine % W).float() + 0.5, (mine // W).float() + 0.5), dim=-1)  # raytracer.py:300-303
            self._pix_cache[key] = (lists, mine, uv)
        return self._pix_cache[key]

    @torch.no_grad()
    def render(self, cameras, collect_stats: bool = False):
        from .raytracer import SDFHandle, intersect_sphere
        H, W = cameras[0].H, cameras[0].W
        dev = cameras[0].device
        lists, mine, uv = self._pix(H, W, dev)
        V = len(cameras)
        n_r = mine.numel()
        # 1. this rank's rays of every view
        o_parts, d_parts, nrm_parts, idx_parts = [], [], [], []
        for v, cam in enumerate(cameras):
            ro, rd, rn = cam.get_rays(uv)
            o_parts.append(ro); d_parts.append(rd); nrm_parts.append(rn)
            idx_parts.append(global_ray_index(mine, v, H, W, self.chunk))
        ray_o, ray_d, ray_n = torch.cat(o_parts), torch.cat(d_parts), torch.cat(nrm_parts)
        ray_index = torch.cat(idx_parts).contiguous()
        hit, near, far = intersect_sphere(ray_o, ray_d, 1.0)
        # 2. tracer phase 0 -> MAX all-reduce of the chunk counts -> phase 1
        n_chunks = V * chunks_per_view(H, W, self.chunk)
        res = self.tracer.forward_phased(SDFHandle(self.sdf_network), ray_o, ray_d, near, far, hit, ray_index, n_chunks,
                                         self.chunk, lambda t: reduce_chunk_iters(t, self.group), collect_stats)
        self.last_stats = self.tracer.last_stats
        res["depth"] = res["distance"] / ray_n * res["convergent_mask"].float()  # raytracer.py:393,552
        res.update({"ray_o": ray_o, "ray_d": ray_d})
        # 3. shade (fused kernels when render_fn is the GGX one)
        from .raytracer import render_normal_and_color
        render_normal_and_color(res, self.sdf_network, self.nets, self.render_fn, is_training=False)
        # 4. pack + gather + un-tile
        cols = []
        for k, w in RECORD:
            v = res[k]
            v = v.float() if k == "convergent_mask" else v
            cols.append(v.reshape(V * n_r, w))
        local = torch.cat(cols, dim=1).contiguous()
        sizes = [V * int(l.numel()) for l in lists]
        parts = gather_records(local, sizes, self.group, dst=0)
        if parts is None:
            return None
        img = assemble_views(parts, lists, V, H, W)
        out = split_record(img)
        return out
