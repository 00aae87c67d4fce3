"""CPU tests of the multi-GPU plumbing: tile interleave maths and the all-gather of compact tile
buffers, with torch.distributed's gloo backend and world_size 2 (the GPU path uses the same code
with backend "nccl" = RCCL)."""
import os
import socket

import numpy as np
import pytest

from cudaraytracing_amd import distributed as D
import cudaraytracing_amd as crt


def _tile(img, rank, world):
    """Inverse of untile_*: what rank `rank` writes with CRT_FLAG_TILED_OUTPUT."""
    h, w, ch = img.shape
    tx, ty = D.tile_grid(w, h)
    lt = D.local_tiles(w, h, world)
    pad = np.zeros((ty * 8, tx * 8, ch), dtype=img.dtype)
    pad[:h, :w] = img
    out = np.zeros((lt * 64, ch), dtype=img.dtype)
    for k in range(lt):
        t = k * world + rank
        if t >= tx * ty:
            break
        y, x = divmod(t, tx)
        out[k * 64:(k + 1) * 64] = pad[y * 8:(y + 1) * 8, x * 8:(x + 1) * 8].reshape(64, ch)
    return out


@pytest.mark.parametrize("w,h,world", [(800, 600, 1), (800, 600, 8), (100, 75, 3), (8, 8, 2), (17, 9, 5)])
def test_untile_inverts_the_shard_layout(w, h, world):
    rng = np.random.default_rng(w * h + world)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    shards = np.stack([_tile(img, r, world) for r in range(world)])
    assert shards.shape[1] == crt.shard_slots(w, h, 0, world)
    assert np.array_equal(D.untile_numpy(shards, w, h), img)
    import torch
    assert np.array_equal(D.untile_torch(torch.from_numpy(shards), w, h).numpy(), img)


def _worker(rank, world, port, w, h, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(99)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    local = torch.from_numpy(_tile(img, rank, world))
    out = D.gather_image(local, w, h, world)
    q.put((rank, bool(np.array_equal(out.numpy(), img))))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_all_gather_reassembles_the_frame():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, 100, 75, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
