"""spp sharding over two ranks (gloo, CPU): the reduced image equals the single-rank image.

The per-rank renderer here is the CPU checker (there is no GPU in this tier); what is under test is the sharding
arithmetic, the reduce and the normalisation of pathtracer_amd.distributed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_api
from pathtracer_amd import distributed, scenes


def test_sample_ranges_partition_the_samples():
    for spp in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                b, c = distributed.sample_range(spp, r, world)
                got += list(range(b, b + c))
            assert got == list(range(spp))
    with pytest.raises(ValueError):
        distributed.sample_range(8, 2, 2)


def _worker(rank, world, port, spp, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.cornell(res=(24, 24), spheres=False)
    o = oracle_api.OracleScene(sc)

    def render_accumulate(acc, begin, count):
        fb = o.render(spp, seed=9, threads=2, sample_begin=begin, sample_count=count)  # mean over spp of the block
        acc += torch.from_numpy(fb.reshape(-1) * spp)

    acc = torch.zeros(24 * 24 * 3)
    img = distributed.render_sharded(render_accumulate, acc, spp, rank, world)
    if rank == 0:
        np.save(out, img.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    spp = 7  # odd on purpose: ranks get 4 and 3 samples
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, port, spp, out), nprocs=2, join=True)
    got = np.load(out).reshape(24, 24, 3)
    sc = scenes.cornell(res=(24, 24), spheres=False)
    want = oracle_api.OracleScene(sc).render(spp, seed=9, threads=4)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)
