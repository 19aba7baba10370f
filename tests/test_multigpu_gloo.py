"""The N > 1 path (interleaved row-block shards + ONE gather to rank 0) with world_size 2 and 3 over gloo on the CPU.
The tiles come from the CPU oracle here (the HIP path needs a GPU); the sharding/gather code is the product's
(raytracer-3_amd/shard.py) — the same object bench.py drives over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tile_rows, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import importlib
    from cases import mode_x_cases, oracle_render, rt3
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = importlib.import_module("raytracer-3_amd.shard")
        case = mode_x_cases()["three_spheres_64x36x16_d8"]
        plist = [rt3.make_params(**dict(case["params"], tile_rows=tile_rows, tile_index=i, tile_count=world)) for i in range(world)]
        g = shard.FrameGatherer(rt3, plist, rank, torch.device("cpu"))
        mine, _ = oracle_render(case, threads=2, tile_rows=tile_rows, tile_index=rank, tile_count=world)
        g.tile[: mine.shape[0]] = torch.from_numpy(mine.view(np.int32))
        frame = g.gather()
        dist.barrier()
        if rank == 0:
            np.save(out_path, frame.numpy().view(np.uint32))
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_rows", [(2, 8), (3, 5), (2, 1)])
def test_sharded_render_gathers_to_the_single_gpu_image(tmp_path, world, tile_rows):
    from cases import mode_x_cases, oracle_render
    out_path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), tile_rows, out_path), nprocs=world, join=True)
    whole, _ = oracle_render(mode_x_cases()["three_spheres_64x36x16_d8"])
    assert np.array_equal(np.load(out_path), whole)
