"""The samples-per-pixel split across ranks (pathtracer_amd/distributed.py) with two
gloo processes on CPU.  The device renderer is replaced by a stand-in that fills the
bound sum tensor from the CPU oracle, so what is exercised is the partition of the
sample index space, the single sum-reduce and the resolve on rank 0."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd import SceneLoader, override_scene_text
from pathtracer_amd.distributed import sample_range, split_render_step
from util import ROOT, scene_text

SPP = 3


class OracleRenderer:
    def __init__(self, scene, sums):
        self.scene, self.sums, self.result = scene, sums, None

    def render_range(self, j0, j1, clear=True):
        from oracle import oracle_py as orc
        s, _, _ = orc.render(self.scene, semantics=1, seeding=1, bounce_cap=8, spp_begin=j0, spp_end=j1, threads=2)
        t = torch.from_numpy(s.reshape(-1))
        if clear:
            self.sums.zero_()
        self.sums += t

    def resolve(self, total_spp):
        g = self.scene.config.gamma
        self.result = (self.sums / total_spp).pow(1.0 / g)


def make_scene():
    return SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), size="24 24", spp=str(SPP)))


def worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = make_scene()
    sums = torch.zeros(scene.width * scene.height * 3, dtype=torch.float32)
    r = OracleRenderer(scene, sums)
    split_render_step(r, sums, rank, world, SPP, dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "sums.npy"), sums.numpy())
        np.save(os.path.join(out_dir, "result.npy"), r.result.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sample_ranges_tile_the_index_space():
    for world in (1, 2, 4, 8):
        ranges = [sample_range(r, world, 512) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == 512 * world
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    with pytest.raises(ValueError):
        sample_range(2, 2, 4)


def test_two_rank_split_equals_single_render(tmp_path):
    from oracle import oracle_py as orc
    world = 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sums = np.load(tmp_path / "sums.npy")
    scene = make_scene()
    scene.samples_per_pixel = world * SPP
    whole, _, _ = orc.render(scene, semantics=1, seeding=1, bounce_cap=8)
    # same samples, different summation order across the two halves
    assert np.allclose(sums, whole.reshape(-1), rtol=1e-5, atol=1e-6)
    res = np.load(tmp_path / "result.npy")
    assert np.allclose(res, (whole.reshape(-1) / (world * SPP)) ** (1.0 / scene.config.gamma), rtol=1e-5, atol=1e-6)
