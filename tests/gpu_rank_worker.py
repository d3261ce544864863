"""One rank of test_gpu_configs.py::test_ranks_sharing_one_gpu: the real renderer on cuda:0, its share of the
samples, the combine through a gloo group on host copies.  argv: rank world port out_dir ordered(0/1) spp_per_rank"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text  # noqa: E402
from pathtracer_amd.distributed import split_render_step  # noqa: E402
from util import scene_text  # noqa: E402

rank, world, port, out_dir, ordered, spp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5] == "1", int(sys.argv[6])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), size="96 80", spp=str(spp * world), maxdepth="6"))
r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1})
sums = torch.zeros(80 * 96 * 3, dtype=torch.float32, device="cuda")
r.bind_sum_buffer(sums.data_ptr(), sums.numel() * 4)
r.set_stream(torch.cuda.current_stream().cuda_stream)
split_render_step(r, sums, rank, world, spp, dist, ordered=ordered, stage_on_host=True)
r.synchronize()
if rank == 0:
    np.save(os.path.join(out_dir, "result.npy"), r.result())
    np.save(os.path.join(out_dir, "sums.npy"), sums.cpu().numpy())
dist.barrier()
r.close()
dist.destroy_process_group()
