"""Samples-per-pixel split across GPUs (one process per GPU).

The path shards over samples: rank g of G renders the global sample indices
[g*spp, (g+1)*spp) of EVERY pixel (sample 0, the pixel-centre sample, belongs to
rank 0 only) with counter-based seeds, into its own pre-gamma radiance sums.  The
only exchange is one sum-reduce of those W*H*3 floats to rank 0 (RCCL over xGMI on
GPUs; gloo in the CPU tests), after which rank 0 averages and gamma-corrects,
because the reference averages and applies gamma after its sample loop
(PinholeCamera.cu:70-71)."""


def sample_range(rank: int, world: int, spp_per_rank: int):
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * spp_per_rank, (rank + 1) * spp_per_rank


def split_render_step(renderer, sums, rank: int, world: int, spp_per_rank: int, dist=None):
    """One whole-job render.  `renderer` accumulates into `sums` (a tensor it was
    bound to); `dist` is torch.distributed (needed when world > 1)."""
    j0, j1 = sample_range(rank, world, spp_per_rank)
    renderer.render_range(j0, j1, clear=True)
    if world > 1:
        dist.reduce(sums, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        renderer.resolve(world * spp_per_rank)
