#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload dragon|cornell|cornell_ggx]

A step = one full render of the workload (every pixel x every sample) through the
C ABI: generate -> (extend -> surface -> shadow -> gather)* -> accumulate, plus, for N > 1, one
RCCL reduce of the pre-gamma radiance sums to rank 0 and the resolve there.  Inputs
(scene, BVH) are resident in HBM before the timed region.  With N > 1 every rank
renders its own `spp` samples of every pixel (global sample indices
[rank*spp, (rank+1)*spp), counter-based seeds), i.e. weak scaling.

Prints ONE JSON line: Mrays/s (extension segments = paths x bounces, whole job), the
roofline object of the dominant kernel class (extension-ray traversal) and the
CPU-oracle baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[0]: spheres + quad lights, `direct`, 640x480, 16 spp (scene 2/sphere.test: the
    # shipped scene with quad lights and the direct integrator; 75 stratified shadow rays per path)
    "sphere": dict(scene="scene2/sphere.test", over=dict(size="640 480", spp="16"),
                   name="scene 2 sphere.test (1 sphere, 2 triangles, 3 quad lights, 25 stratified samples each) direct lighting, 640x480, 16 spp"),
    # BASELINE.json configs[3]: Stanford dragon GGX, 1080p, 256 spp (BVH-traversal bound)
    "dragon": dict(scene="scene4/dragon.test", over=dict(size="1920 1080", spp="256"),
                   name="Stanford dragon (100006 tris, 2 quad lights) GGX BRDF sampling + NEE + RR, maxdepth 5, 1920x1080, 256 spp"),
    # configs[1]: Cornell box Phong, NEE + Russian roulette, 512x512, 1024 spp
    "cornell": dict(scene="scene3/cornellRR.test", over=dict(size="512 512", spp="1024"),
                    name="Cornell box Phong, NEE + Russian roulette, maxdepth -1, 512x512, 1024 spp"),
    # configs[2] without the (NaN-producing) MIS weights: Cornell, GGX sphere, BRDF importance sampling + NEE
    "cornell_ggx": dict(scene="scene4/cornellBRDF.test", over=dict(size="512 512", spp="4096"), ggx_sphere=True,
                        name="Cornell box, GGX sphere (roughness 0.25), BRDF importance sampling + NEE + RR, 512x512, 4096 spp"),
}

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def build_scene(workload, spp_override=None, size_override=None):
    from pathtracer_amd import SceneLoader, override_scene_text
    w = WORKLOADS[workload]
    text = open(os.path.join(ROOT, "scenes", w["scene"])).read()
    over = dict(w["over"])
    if spp_override:
        over["spp"] = str(spp_override)
    if size_override:
        over["size"] = size_override
    text = override_scene_text(text, **over)
    if w.get("ggx_sphere"):
        text = text.replace("#sphere\n", "#sphere\nbrdf ggx\nroughness 0.25\n", 1)
    return SceneLoader().loadText(text)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline config)")
    ap.add_argument("--size", default="", help='override "W H" (invalidates the headline config)')
    ap.add_argument("--semantics", type=int, default=1, help="1 = intended (by-reference helpers), 0 = literal")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--batch", type=int, default=None, help="PTC_OPT_BATCH_PATHS")
    ap.add_argument("--builder", type=int, default=None, help="PTC_OPT_BVH_BUILDER: 0 LBVH, 1 PLOC")
    ap.add_argument("--tune", type=lambda v: int(v, 0), default=None, help="PTC_OPT_TUNE_TRAVERSAL value (experiments)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pathtracer_amd import Renderer, _ffi
    from pathtracer_amd.distributed import split_render_step

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    scene = build_scene(args.workload, args.spp or None, args.size or None)
    spp = scene.samples_per_pixel
    W, H = scene.width, scene.height

    opts = {_ffi.OPT_SEMANTICS: args.semantics, _ffi.OPT_SEEDING: 1}
    if args.tune is not None:
        opts[_ffi.OPT_TUNE_TRAVERSAL] = args.tune
    if args.batch is not None:
        opts[_ffi.OPT_BATCH_PATHS] = args.batch
    if args.builder is not None:
        opts[_ffi.OPT_BVH_BUILDER] = args.builder
    r = Renderer(scene, device=local_rank, options=opts)
    sums = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    r.bind_sum_buffer(sums.data_ptr(), sums.numel() * 4)
    r.set_stream(torch.cuda.current_stream().cuda_stream)

    def step():
        # this rank's share of the global sample index space, then one RCCL reduce
        split_render_step(r, sums, rank, world, spp, dist)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # traversal statistics for the algorithmic-bytes figure (outside the timed region)
    r.set_option(_ffi.OPT_COUNT_TRAVERSAL, 1)
    r.reset_stats()
    r.render_range(0, min(spp, 2), clear=True)
    r.synchronize()
    st = r.stats()
    nodes_per_ray = st.nodes_visited / max(st.extension_rays, 1)
    prims_per_ray = st.prims_tested / max(st.extension_rays, 1)
    snodes_per_ray = st.shadow_nodes_visited / max(st.shadow_rays, 1)
    sprims_per_ray = st.shadow_prims_tested / max(st.shadow_rays, 1)
    r.set_option(_ffi.OPT_COUNT_TRAVERSAL, 0)

    for _ in range(args.warmup):
        step()
    sync()
    r.set_option(_ffi.OPT_PROFILE, 1)
    r.reset_stats()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    st = r.stats()
    r.set_option(_ffi.OPT_PROFILE, 0)

    tvals = torch.tensor([dt], dtype=torch.float64, device="cuda")
    counts = torch.tensor([st.extension_rays, st.shadow_rays, st.paths], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tvals, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dt = float(tvals[0])
    ext_rays, shadow_rays, paths = (float(x) for x in counts)

    if rank == 0:
        # Extension-ray traversal (k_trace_deferred<false>; k_extend_static for tiny hierarchies)
        # is the kernel class the roofline is quoted on.  Algorithmic bytes per extension ray: the
        # queue entry (4 B) and the ray (2 x float4 = 32 B) read, the hit (t + ref = 8 B)
        # written, 64 B per BVH node visited and 48 B per triangle tested.  The shadow-ray
        # kernel is reported the same way (32 B ray record read, 12 B written when occluded).
        b_ext = 44.0 + 64.0 * nodes_per_ray + 48.0 * prims_per_ray
        b_sh = 44.0 + 64.0 * snodes_per_ray + 48.0 * sprims_per_ray
        extend_s = st.extend_ms * 1e-3
        achieved = (b_ext * st.extension_rays / extend_s / 1e9) if extend_s > 0 else 0.0
        launches = max(int(st.extend_launches), 1)
        traffic = None
        tpath = os.path.join(ROOT, "pmc_traffic.json")
        if os.path.exists(tpath) and not (args.spp or args.size):
            t = json.load(open(tpath))
            if t.get("workload") == args.workload and "extend" in t:
                traffic = round(t["extend"]["fetch_bytes_per_launch"] + t["extend"]["write_bytes_per_launch"], 1)
        roofline = {
            "bound": "hbm", "kernel": "extension-ray traversal (k_trace_deferred<false> over four-wide nodes; k_extend_static for hierarchies of at most 64 nodes)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": round(b_ext * st.extension_rays / launches, 1),
            "bytes_per_ray": round(b_ext, 1), "nodes_per_ray": round(nodes_per_ray, 2), "prims_per_ray": round(prims_per_ray, 2),
            "avg_launch_ms": round(st.extend_ms / launches, 4), "launches": int(st.extend_launches),
            "rays_per_launch": round(st.extension_rays / launches, 1),
            "kernel_ms": {"extend": round(st.extend_ms, 2), "shadow": round(st.connect_ms, 2), "surface+gather": round(st.shade_ms, 2), "other": round(st.other_ms, 2)},
            "shadow": {"bytes_per_ray": round(b_sh, 1), "nodes_per_ray": round(snodes_per_ray, 2), "prims_per_ray": round(sprims_per_ray, 2),
                       "rays": int(st.shadow_rays)},
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(args, scene)
        out = {
            "metric": "Mrays/s (paths x bounces)", "value": round(ext_rays / dt / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload]["name"] + (" [overridden]" if (args.spp or args.size) else ""),
                       "width": W, "height": H, "spp_per_gpu": spp, "semantics": "intended" if args.semantics else "literal",
                       "seeding": "counter", "parallelism": f"spp-split x{world}"},
            "total_mrays_per_s": round((ext_rays + shadow_rays) / dt / 1e6, 2),
            "shadow_mrays_per_s": round(shadow_rays / dt / 1e6, 2),
            "bounces_per_path": round(ext_rays / max(paths, 1), 3),
            "bvh": {"nodes": int(st.bvh_nodes), "max_depth": int(st.bvh_max_depth), "build_ms": round(st.build_ms, 2)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, scene):
    """The CPU oracle on a bounded, strided sample of the same workload."""
    from oracle import oracle_py as orc
    threads = min(16, os.cpu_count() or 1)
    stride, spp = 8, 1
    # grow the sample until it takes a meaningful share of the time budget
    best = None
    spent = 0.0
    while True:
        saved = scene.samples_per_pixel
        scene.samples_per_pixel = spp
        try:
            _, _, st = orc.render(scene, semantics=args.semantics, seeding=1, use_bvh=1, threads=threads, stride=stride, spp_begin=0, spp_end=spp)
        finally:
            scene.samples_per_pixel = saved
        spent += st.seconds
        best = (st, stride, spp)
        if st.seconds >= args.cpu_seconds / 4 or spent >= args.cpu_seconds or (stride == 1 and spp >= 16):
            break
        if stride > 1:
            stride //= 2
        else:
            spp *= 2
    st, stride, spp = best
    return {"value": round(st.extension_rays / st.seconds / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"every {stride}th pixel in x and y, {spp} spp, same scene and options; {st.extension_rays} extension + {st.shadow_rays} shadow rays in {st.seconds:.1f} s (incl. CPU BVH build)"}


if __name__ == "__main__":
    main()
