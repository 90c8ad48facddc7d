"""world_size-2 `gloo` test (CPU) of the N>1 host path: tile sharding, the MAX all-reduce of the per-chunk
bisection counts, the padded gather of pixel records and the un-tile permutation on rank 0."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _record_of(pix, view, width):
    """Deterministic fake per-pixel record: channel c of pixel p in view v = p + 1000*c + 0.5*v."""
    c = torch.arange(width, dtype=torch.float32).view(1, width)
    return pix.float().view(-1, 1) + 1000.0 * c + 0.5 * view


def _worker(rank, world, port, H, W, tile, n_views, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iron_amd.sharding import (RECORD_WIDTH, TRACE_WIDTH, all_gather_records, assemble_views, chunks_per_view, gather_records,
                                   global_ray_index, reduce_chunk_iters, tile_pixels)
    lists = [tile_pixels(H, W, tile, world, r) for r in range(world)]
    mine = lists[rank]
    chunk = 1000
    # exchange 1: each rank knows the bisection count only of the chunks its rays fall in
    n_chunks = n_views * chunks_per_view(H, W, chunk)
    iters = torch.zeros(n_chunks, dtype=torch.int32)
    for v in range(n_views):
        ch = global_ray_index(mine, v, H, W, chunk) // chunk
        for c in ch.unique().tolist():
            iters[c] = max(int(iters[c]), 3 + (c + rank) % 5)
    local_before = iters.clone()
    reduce_chunk_iters(iters)
    assert torch.all(iters >= local_before)
    # exchange for fill_holes: every rank receives every rank's trace records and can assemble the whole image
    tl = torch.cat([_record_of(mine, v, TRACE_WIDTH) for v in range(n_views)], dim=0)
    everyone = all_gather_records(tl, [n_views * int(l.numel()) for l in lists])
    full = assemble_views(everyone, lists, n_views, H, W)
    for v in range(n_views):
        assert torch.equal(full[v], _record_of(torch.arange(H * W), v, TRACE_WIDTH).reshape(H, W, TRACE_WIDTH))
    # exchange 2: gather + un-tile
    local = torch.cat([_record_of(mine, v, RECORD_WIDTH) for v in range(n_views)], dim=0)
    sizes = [n_views * int(l.numel()) for l in lists]
    parts = gather_records(local, sizes, dst=0)
    if rank == 0:
        img = assemble_views(parts, lists, n_views, H, W)
        torch.save({"img": img, "iters": iters}, out_path)
    else:
        assert parts is None
        torch.save({"iters": iters}, out_path + ".r1")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W,tile,n_views", [(96, 80, 32, 2), (70, 50, 16, 1)])
def test_two_rank_gather_and_untile(H, W, tile, n_views):
    from iron_amd.sharding import RECORD_WIDTH
    world = 2
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res.pt")
        mp.spawn(_worker, args=(world, port, H, W, tile, n_views, out), nprocs=world, join=True)
        r0 = torch.load(out)
        r1 = torch.load(out + ".r1")
    img = r0["img"]
    assert img.shape == (n_views, H, W, RECORD_WIDTH)
    pix = torch.arange(H * W)
    for v in range(n_views):
        want = _record_of(pix, v, RECORD_WIDTH).reshape(H, W, RECORD_WIDTH)
        assert torch.equal(img[v], want)
    # both ranks end with the same (max-reduced) chunk table
    assert torch.equal(r0["iters"], r1["iters"])
    assert int(r0["iters"].max()) >= 3
