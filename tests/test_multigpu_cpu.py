"""The N > 1 path on CPU: world_size-2 gloo, samples sharded exactly as bench.py / the GPU path
shard them (pathed_amd/parallel.py), one reduce(SUM) to rank 0.  The oracle stands in for the
per-rank renderer here (no GPU in this container); what is under test is the sharding and the
reduce, which are backend-independent."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathed_amd import parallel

WIDTH = HEIGHT = 16
SPP = 6
SEED = 9


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_path):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib
    from pathed_amd.scene import LoadedScene

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    scene = LoadedScene("scenes/cornell.json", WIDTH, HEIGHT)
    oracle = oracle_lib.OracleScene(scene.desc)
    begin, count = parallel.strong_range(rank, world_size, 0, SPP)
    image, _ = oracle.render(WIDTH, HEIGHT, SEED, begin, count, 0, 5, threads=1)
    tensor = torch.from_numpy(image)
    parallel.reduce_to_root(tensor, root=0)
    if rank == 0:
        np.save(out_path, tensor.numpy())
    dist.destroy_process_group()


def test_strong_and_weak_ranges_partition_the_samples():
    for world in (1, 2, 3, 4, 8):
        covered = []
        for rank in range(world):
            begin, count = parallel.strong_range(rank, world, 10, 37)
            covered.extend(range(begin, begin + count))
        assert covered == list(range(10, 47))
        spans = [parallel.weak_range(rank, 64, first=5) for rank in range(world)]
        assert [s[0] for s in spans] == [5 + 64 * r for r in range(world)]


def test_two_rank_sharded_render_equals_single_process(tmp_path):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib
    from pathed_amd.scene import LoadedScene

    out_path = str(tmp_path / "reduced.npy")
    mp.spawn(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    reduced = np.load(out_path)

    scene = LoadedScene("scenes/cornell.json", WIDTH, HEIGHT)
    oracle = oracle_lib.OracleScene(scene.desc)
    single, _ = oracle.render(WIDTH, HEIGHT, SEED, 0, SPP, 0, 5, threads=1)
    # only the fp32 summation order differs (relL2 <= 1e-5)
    rel = np.linalg.norm(reduced - single) / np.linalg.norm(single)
    assert rel < 1e-6
