"""Samples-per-pixel split across GPUs (one process per GPU).

The path shards over samples: rank g of G renders the global sample indices
[g*spp, (g+1)*spp) of EVERY pixel (sample 0, the pixel-centre sample, belongs to
rank 0 only) with counter-based seeds, into its own pre-gamma radiance sums.  The
only exchange is one combine of those W*H*3 floats on rank 0 (RCCL over xGMI on
GPUs; gloo in the CPU tests), after which rank 0 averages and gamma-corrects,
because the reference averages and applies gamma after its sample loop
(PinholeCamera.cu:70-71).

Two ways to combine:
  * one sum-reduce to rank 0 (default): a single collective; the order in which the
    ranks' floats are added is the library's, so the last bits may differ from run
    to run and with the number of ranks;
  * ordered=True: the ranks' buffers are gathered on rank 0 and added there in rank
    order, ((s0 + s1) + s2) + ... -- bit-reproducible, and equal bit for bit to
    rendering the same sample ranges one after the other on one GPU and adding the
    partial sums in that order.  Costs G buffers on rank 0 (8 x 99.5 MB at 4K).
"""


def sample_range(rank: int, world: int, spp_per_rank: int):
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * spp_per_rank, (rank + 1) * spp_per_rank


def ordered_reduce(sums, rank: int, world: int, dist):
    """sums (rank 0) <- ((sums_0 + sums_1) + sums_2) + ... in rank order; other ranks keep theirs."""
    import torch
    parts = [torch.empty_like(sums) for _ in range(world)] if rank == 0 else None
    dist.gather(sums, gather_list=parts, dst=0)
    if rank == 0:
        acc = parts[0]
        for p in parts[1:]:
            acc += p
        sums.copy_(acc)


def split_render_step(renderer, sums, rank: int, world: int, spp_per_rank: int, dist=None, ordered: bool = False,
                      stage_on_host: bool = False, phase_times: dict = None):
    """One whole-job render.  `renderer` accumulates into `sums` (a tensor it was
    bound to, rendering on the stream torch uses -- Renderer.set_stream -- so that the
    collective is ordered after the render); `dist` is torch.distributed (needed when
    world > 1).  `stage_on_host`: the combine runs on host copies of the sums, for
    process groups that cannot move device memory (gloo) -- which is how several ranks
    that share ONE GPU rehearse the whole path (RCCL refuses a device that appears twice).
    `phase_times` (a dict with "render_s" and "reduce_s"): this rank's host time in the two halves of the step is
    added to it -- the render half ends with a stream synchronise (the render call has waited for all but its last
    few kernels anyway), the other half (combine + resolve) with a synchronise of the device."""
    import time
    j0, j1 = sample_range(rank, world, spp_per_rank)
    t0 = time.perf_counter()
    renderer.render_range(j0, j1, clear=True)
    if phase_times is not None:
        renderer.synchronize()
        t1 = time.perf_counter()
        phase_times["render_s"] += t1 - t0
    if world > 1:
        buf = sums.cpu() if stage_on_host else sums  # .cpu() waits for the render on torch's stream
        if ordered:
            ordered_reduce(buf, rank, world, dist)
        else:
            dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM)
        if stage_on_host and rank == 0:
            sums.copy_(buf)
    if rank == 0:
        renderer.resolve(world * spp_per_rank)
    if phase_times is not None:
        renderer.synchronize()
        if world > 1 and not stage_on_host:
            import torch
            torch.cuda.synchronize()  # the collective may run on a stream of its own
        phase_times["reduce_s"] += time.perf_counter() - t1
