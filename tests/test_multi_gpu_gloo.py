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


def worker(rank, world, port, out_dir, ordered=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = make_scene()
    sums = torch.zeros(scene.width * scene.height * 3, dtype=torch.float32)
    r = OracleRenderer(scene, sums)
    split_render_step(r, sums, rank, world, SPP, dist, ordered=ordered)
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


@pytest.mark.parametrize("world", [2, 3])
def test_rank_ordered_reduce_is_bit_exact(tmp_path, world):
    """ordered=True: rank 0 ends up with ((s0 + s1) + s2)..., bit for bit what rendering the same
    sample ranges one after the other and adding the partial sums in rank order gives."""
    from oracle import oracle_py as orc
    port = 31500 + os.getpid() % 2000 + world
    mp.spawn(worker, args=(world, port, str(tmp_path), True), nprocs=world, join=True)
    sums = np.load(tmp_path / "sums.npy")
    scene = make_scene()
    want = None
    for g in range(world):
        j0, j1 = sample_range(g, world, SPP)
        part, _, _ = orc.render(scene, semantics=1, seeding=1, bounce_cap=8, spp_begin=j0, spp_end=j1, threads=2)
        part = part.reshape(-1)
        want = part.copy() if want is None else want + part
    assert np.array_equal(sums.view(np.uint32), want.view(np.uint32))


def test_bench_launcher_starts_the_ranks_it_reports(tmp_path):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE) starts N ranks of itself; with
    --launcher-selftest they only join a gloo group on the CPU and count themselves."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    for n in (2, 3):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launcher-selftest"], env=env,
                           capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
        assert len(lines) == 1
        out = json.loads(lines[0])
        assert out["ranks_joined"] == n and out["n_gpus"] == 0
    # a rank that fails makes the launcher fail: without a GPU the real benchmark refuses to run
    if not torch.cuda.is_available():
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env,
                           capture_output=True, timeout=300)
        assert p.returncode != 0 and not p.stdout.decode().strip()
    # a rank that dies before the rendezvous: the launcher ends the job at once instead of waiting with the others
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=dict(env, BENCH_SELFTEST_FAIL_RANK="1"),
                       capture_output=True, timeout=120)
    assert p.returncode != 0 and not p.stdout.decode().strip() and time.time() - t0 < 60
    # --gpus must agree with the launcher's WORLD_SIZE
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launcher-selftest"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, timeout=120)
    assert p.returncode != 0 and b"WORLD_SIZE=2" in p.stderr
