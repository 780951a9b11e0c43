"""N > 1 host path on CPU: tile ownership, pack / gather / unpack over torch.distributed (gloo,
world_size 2).  The renderer of each rank is the ORACLE here (CPU test only): what is under test is
the sharding protocol of radiance-ray-tracing_amd/dist.py, whose result must equal the unsharded
frame bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_math():
    import rrt_amd
    from radiance_ray_tracing_amd import dist
    W, H, tw, th = 100, 60, 16, 16
    for world in (1, 2, 3, 8):
        seen = np.zeros(W * H, int)
        for r in range(world):
            px = dist.owned_pixels(r, world, W, H, tw, th)
            seen[px] += 1
            img = np.arange(W * H * 4, dtype=np.float32).reshape(W * H, 4)
            packed = dist.pack_tiles_np(img, r, world, W, H, tw, th)
            assert packed.shape[0] == len(dist.owned_tile_ids(r, world, W, H, tw, th)) * tw * th
            out = np.zeros_like(img)
            dist.unpack_tiles_np(packed, out, r, world, W, H, tw, th)
            assert np.array_equal(out[px], img[px]) and out.sum() == img[px].sum()
        assert (seen == 1).all()
        assert dist.max_owned_tiles(world, W, H, tw, th) >= len(dist.owned_tile_ids(world - 1, world, W, H, tw, th))


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as tdist
    import oracle_bind as ob
    import rrt_amd
    from radiance_ray_tracing_amd import dist, scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, tw, th = 40, 24, 8, 8
    s = scenes.c0_two_boxes(W, H, spp=2, depth=3)
    osc = ob.OracleScene(s)
    px = dist.owned_pixels(rank, world, W, H, tw, th)
    osc.render(nthreads=1, pixels=px)                 # this rank's shard only
    mt = dist.max_owned_tiles(world, W, H, tw, th)
    packed = np.zeros((mt * tw * th, 4), np.float32)
    mine = dist.pack_tiles_np(osc.scratch.reshape(-1, 4), rank, world, W, H, tw, th)
    packed[:mine.shape[0]] = mine
    bufs = dist.gather_to_root(torch.from_numpy(packed), world, 0)
    if rank == 0:
        full = np.zeros((W * H, 4), np.float32)
        for r in range(world):
            dist.unpack_tiles_np(bufs[r].numpy(), full, r, world, W, H, tw, th)
        ref = ob.OracleScene(s)
        ref.render(nthreads=1)
        np.save(out_path, np.stack([full.reshape(-1), ref.scratch]))
    tdist.barrier()
    tdist.destroy_process_group()


def test_gloo_world2_sharded_frame_equals_full(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    full, ref = np.load(out)
    assert np.array_equal(full.view(np.uint32), ref.view(np.uint32))
