#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload dragon|dragon_4k|cornell|cornell_ggx|sphere]
                    [--strong] [--semantics 0|1] [--ordered-reduce]

A step = one full render of the workload (every pixel x every sample of this rank's share) through
the C ABI: generate -> (extend -> surface -> shadow -> gather)* -> accumulate, plus, for N > 1, one
RCCL reduce of the pre-gamma radiance sums to rank 0 and the resolve there.  Inputs (scene, BVH) are
resident in HBM before the timed region.

N > 1: one process per GPU.  Started by a launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) this file is one rank; started plainly with --gpus N it starts N fresh ranks of itself
before anything here touches a GPU, relays rank 0's line and fails if any rank fails.  Weak scaling
(default): every rank renders the workload's own samples per pixel, global sample indices
[rank*spp, (rank+1)*spp), counter-based seeds.  --strong: the workload's samples are divided among the
ranks.

Prints ONE JSON line: Mrays/s (extension segments = paths x bounces, whole job), the roofline object
of the dominant kernel class (extension-ray traversal) and the CPU-oracle baseline.  The default single-GPU
run (workload dragon) then measures the other half of the metric's config, the Cornell box with the GGX sphere
(512x512, 4096 spp), and attaches it as "secondary" with the roofline object of ITS dominant kernel, k_surface.
"""
import argparse
import json
import os
import socket
import subprocess
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
    # BASELINE.json configs[4]: the same scene at 3840x2160, 4096 spp over 8 GPUs = 512 spp per GPU (weak
    # scaling reaches the configuration at N = 8; --strong --spp 4096 divides it at any N)
    "dragon_4k": dict(scene="scene4/dragon.test", over=dict(size="3840 2160", spp="512"),
                      name="Stanford dragon (100006 tris, 2 quad lights) GGX BRDF sampling + NEE + RR, maxdepth 5, 3840x2160, 512 spp per GPU (4096 spp on 8 GPUs)"),
    # configs[1]: Cornell box Phong, NEE + Russian roulette, 512x512, 1024 spp
    "cornell": dict(scene="scene3/cornellRR.test", over=dict(size="512 512", spp="1024"),
                    name="Cornell box Phong, NEE + Russian roulette, maxdepth -1, 512x512, 1024 spp"),
    # configs[2] without the (NaN-producing) MIS weights: Cornell, GGX sphere, BRDF importance sampling + NEE
    "cornell_ggx": dict(scene="scene4/cornellBRDF.test", over=dict(size="512 512", spp="4096"), ggx_sphere=True,
                        name="Cornell box, GGX sphere (roughness 0.25), BRDF importance sampling + NEE + RR, 512x512, 4096 spp"),
}

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak
CUS, SIMDS_PER_CU = 256, 4
VALU_CYCLES_PER_WAVE_INST = 2.0   # a wave64 VALU instruction occupies a 32-lane SIMD for 2 cycles


def scene_text_for(workload, spp_override=None, size_override=None):
    from pathtracer_amd import override_scene_text
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
    return text


def build_scene(workload, spp_override=None, size_override=None):
    from pathtracer_amd import SceneLoader
    return SceneLoader().loadText(scene_text_for(workload, spp_override, size_override))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline config)")
    ap.add_argument("--size", default="", help='override "W H" (invalidates the headline config)')
    ap.add_argument("--semantics", type=int, default=1, help="1 = intended (by-reference helpers), 0 = literal")
    ap.add_argument("--strong", action="store_true", help="divide the workload's samples among the ranks (default: every rank renders all of them)")
    ap.add_argument("--ordered-reduce", action="store_true", help="gather the ranks' sums and add them in rank order on rank 0 (deterministic) instead of one sum-reduce")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="do not measure the Cornell GGX workload after the dragon (default single-GPU run)")
    ap.add_argument("--secondary-steps", type=int, default=5)
    ap.add_argument("--profile-min-paths", type=int, default=65536,
                    help="kernels of bounces with fewer live paths are not bracketed by HIP events (1 = every kernel)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--batch", type=int, default=None, help="PTC_OPT_BATCH_PATHS")
    ap.add_argument("--builder", type=int, default=None, help="PTC_OPT_BVH_BUILDER")
    ap.add_argument("--tune", type=lambda v: int(v, 0), default=None, help="PTC_OPT_TUNE_TRAVERSAL value (experiments)")
    ap.add_argument("--sort-rays", type=int, default=None, help="PTC_OPT_SORT_RAYS (experiments)")
    ap.add_argument("--opt", action="append", default=[], metavar="ID=VALUE", help="any other PTC_OPT_* by number (experiments), e.g. --opt 11=0")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: every rank renders on GPU 0 and the collectives run through gloo "
                         "on host copies (RCCL refuses a device that appears twice); the line then reports n_gpus 1 and the rank count")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="ranks only join a gloo group on the CPU and count themselves (tests the launcher without GPUs)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------ launcher --
def launch_ranks(n, argv):
    """Start n fresh ranks of this file (one per GPU), relay rank 0's stdout, fail if any rank fails.
    Nothing in this process touches a GPU (torch is not even imported)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Watch all ranks: one that fails ends the job at once (the others may be blocked in the rendezvous waiting for it).
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + 1800
    failed = None
    while failed is None and time.time() < deadline:
        states = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in states):
            failed = states
        elif all(rc == 0 for rc in states):
            break
        else:
            time.sleep(0.1)
    if failed is not None or any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        sys.stderr.write(f"bench.py: ranks exited with {[p.returncode for p in procs]}\n")
        return 1
    reader.join(timeout=10)
    # rank 0's result line goes to stdout; whatever else a library printed there (gloo announces its connections) to stderr
    for line in b"".join(chunks).decode(errors="replace").splitlines():
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return 0


def launcher_selftest(rank, world):
    if os.environ.get("BENCH_SELFTEST_FAIL_RANK") == str(rank):  # a rank that dies before the rendezvous
        return 3
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    joined = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(joined)
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "ranks_joined": int(joined[0]), "backend": "gloo", "n_gpus": 0}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


# -------------------------------------------------------------------------------- one rank --
def quoted_counters(workload, semantics, overridden):
    """Per-launch hardware-counter figures of this workload's kernels, as committed by tools/pmc_summary.py from
    rocprofv3 --pmc passes of this same command (they cannot be collected inside a normal run) -> (entry, stale).
    Every entry carries the hash of the device sources it was measured on; when the sources have changed since,
    `stale` is true and the caller must not derive a roofline fraction from the entry."""
    from pathtracer_amd.srchash import device_source_hash
    path = os.path.join(ROOT, "pmc_counters.json")
    if overridden or not os.path.exists(path):
        return None, False
    try:
        data = json.load(open(path))
    except ValueError:
        return None, False
    e = data.get(f"{workload}/semantics{semantics}")
    if e is None:
        return None, False
    return e, e.get("code_hash") != device_source_hash()


def valu_roofline(e, units_per_launch, avg_launch_s, per_unit, q, stale):
    """The VALU-issue and HBM fractions of one kernel class from its quoted counter entry `e` (per work item: ray or
    hit) and this run's live launch size and duration."""
    clock_ghz = e["clock_ghz"]
    insts = e[f"valu_insts_per_{per_unit}"]
    valu_rate = insts * units_per_launch / avg_launch_s          # wave instructions per second
    valu_peak = CUS * SIMDS_PER_CU * clock_ghz * 1e9 / VALU_CYCLES_PER_WAVE_INST
    hbm_bytes = e[f"hbm_bytes_per_{per_unit}"] * units_per_launch
    out = {
        "bound": "valu", "achieved": round(valu_rate / 1e9, 1), "peak": round(valu_peak / 1e9, 1), "unit": "G wave-instructions/s",
        "frac": round(valu_rate / valu_peak, 4),
        "peak_note": f"{CUS} CUs x {SIMDS_PER_CU} SIMDs x {clock_ghz:.3f} GHz (GRBM_GUI_ACTIVE / 8 / kernel time in the counter pass) / {VALU_CYCLES_PER_WAVE_INST:.0f} cycles per wave64 instruction",
        f"valu_insts_per_{per_unit}": round(insts, 1), "active_lane_frac": round(e["active_lane_frac"], 3),
        "salu_per_valu": round(e["salu_per_valu"], 3),
        "traffic": round(hbm_bytes, 1),
        "traffic_note": f"HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) per {per_unit} from the committed counter passes x this run's {per_unit}s per launch; quoted, not measured in this run",
        "hbm_GBs": round(hbm_bytes / avg_launch_s / 1e9, 1), "hbm_peak_GBs": HBM_PEAK_GBS, "hbm_frac": round(hbm_bytes / avg_launch_s / 1e9 / HBM_PEAK_GBS, 4),
        "counters_from": q.get("source", "pmc_counters.json"), "counters_code_hash": q.get("code_hash"), "counters_stale": False,
    }
    if stale:
        # the kernels have changed since the counter passes: the per-ray figures may no longer hold
        for k in ("achieved", "frac", "traffic", "hbm_GBs", "hbm_frac"):
            out[k] = None
        out["counters_stale"] = True
        out["note"] = "the device sources differ from the ones the committed counter passes ran on (tools/pmc_passes.sh has to be run again): no fraction is derived"
    return out


def measure(args, workload, steps, warmup, dist, rank, world, local_rank, shared, cdev, with_cpu_baseline):
    """Warm up, time `steps` renders of `workload` (this rank's share + the reduce) and return rank 0's result object
    (None on the other ranks)."""
    import torch
    from pathtracer_amd import Renderer, _ffi
    from pathtracer_amd.distributed import split_render_step
    secondary = workload != args.workload
    spp_over, size_over = (None, None) if secondary else (args.spp or None, args.size or None)
    scene = build_scene(workload, spp_over, size_over)
    spp_job = scene.samples_per_pixel
    if args.strong:
        if spp_job % world:
            raise SystemExit(f"--strong: {spp_job} samples do not divide among {world} ranks")
        spp = spp_job // world
    else:
        spp = spp_job
    W, H = scene.width, scene.height

    opts = {_ffi.OPT_SEMANTICS: args.semantics, _ffi.OPT_SEEDING: 1}
    if not secondary:
        if args.tune is not None:
            opts[_ffi.OPT_TUNE_TRAVERSAL] = args.tune
        if args.batch is not None:
            opts[_ffi.OPT_BATCH_PATHS] = args.batch
        if args.sort_rays is not None:
            opts[_ffi.OPT_SORT_RAYS] = args.sort_rays
    if args.builder is not None:
        opts[_ffi.OPT_BVH_BUILDER] = args.builder
    for kv in args.opt:
        k, v = kv.split("=")
        opts[int(k, 0)] = int(v, 0)
    r = Renderer(scene, device=local_rank, options=opts)
    sums = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    r.bind_sum_buffer(sums.data_ptr(), sums.numel() * 4)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    phase = {"render_s": 0.0, "reduce_s": 0.0}

    def step():
        # this rank's share of the global sample index space, then one RCCL reduce
        split_render_step(r, sums, rank, world, spp, dist, ordered=args.ordered_reduce, stage_on_host=shared, phase_times=phase)

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

    for _ in range(warmup):
        step()
    sync()
    # HIP events around every kernel of the bounces with at least PROFILE_MIN_PATHS live paths; the long tail of
    # sub-millisecond launches (the Cornell boxes run hundreds of bounces) is not instrumented: two event records per
    # kernel cost more there than the kernels
    r.set_option(_ffi.OPT_PROFILE, args.profile_min_paths)
    r.reset_stats()
    phase["render_s"] = phase["reduce_s"] = 0.0
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    st = r.stats()
    r.set_option(_ffi.OPT_PROFILE, 0)

    tvals = torch.tensor([dt, phase["render_s"], phase["reduce_s"]], dtype=torch.float64, device=cdev)
    tmin = torch.tensor([phase["render_s"]], dtype=torch.float64, device=cdev)
    counts = torch.tensor([st.extension_rays, st.shadow_rays, st.paths], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tvals, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dt = float(tvals[0])
    ext_rays, shadow_rays, paths = (float(x) for x in counts)

    out = None
    if rank == 0:
        overridden = bool(spp_over or size_over)
        # Extension-ray traversal (k_trace_deferred<false>; k_extend_static for tiny hierarchies) is the
        # kernel class the roofline is quoted on (rank 0's launches).  Times: HIP events on the launch
        # stream, live.  Instruction and byte counts per ray: hardware counters of the same command,
        # committed in pmc_counters.json (they need rocprofv3 --pmc passes of their own).
        launches = max(int(st.extend_timed_launches), 1)
        extend_s = st.extend_ms * 1e-3
        rays_per_launch = st.extend_timed_rays / launches
        avg_launch_s = extend_s / launches if extend_s > 0 else 0.0
        # algorithmic bytes per extension ray: the ray (2 x float4 = 32 B) read, the hit (t + ref = 8 B) written,
        # 4 B for the queue slot, 64 B per BVH node visited and 48 B per triangle tested -- served mostly by L1/L2
        b_ext = 44.0 + 64.0 * nodes_per_ray + 48.0 * prims_per_ray
        b_sh = 44.0 + 64.0 * snodes_per_ray + 48.0 * sprims_per_ray
        on_chip_gbs = (b_ext * st.extend_timed_rays / extend_s / 1e9) if extend_s > 0 else 0.0
        q, stale = quoted_counters(workload, args.semantics, overridden)
        surf_launches = max(int(st.surface_timed_launches), 1)
        hits_per_launch = st.surface_timed_hits / surf_launches
        surf_avg_s = st.surface_ms * 1e-3 / surf_launches
        kernel_ms = {"extend": round(st.extend_ms, 2), "shadow": round(st.connect_ms, 2), "surface": round(st.surface_ms, 2), "gather": round(st.gather_ms, 2),
                     "other": round(st.other_ms, 2)}
        roofline = {
            "kernel": "extension-ray traversal (k_trace_deferred<false> over four-wide nodes; k_extend_static / k_extend for hierarchies of at most 6 / 64 nodes)",
            "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": int(st.extend_timed_launches), "rays_per_launch": round(rays_per_launch, 1),
            "timed": f"launches of bounces with at least {args.profile_min_paths} live paths: {int(st.extend_timed_launches)} of {int(st.extend_launches)} launches, "
                     f"{st.extend_timed_rays / max(st.extension_rays, 1):.4f} of the extension rays",
            "nodes_per_ray": round(nodes_per_ray, 2), "prims_per_ray": round(prims_per_ray, 2),
            "algorithmic_bytes_per_ray": round(b_ext, 1), "algorithmic_bytes_per_launch": round(b_ext * rays_per_launch, 1),
            "on_chip_GBs": round(on_chip_gbs, 1),
            "on_chip_note": "algorithmic bytes / kernel time; node and triangle records are served by L1/L2, so this is NOT a fraction of the HBM peak",
            "kernel_ms": kernel_ms,
            "shadow": {"algorithmic_bytes_per_ray": round(b_sh, 1), "nodes_per_ray": round(snodes_per_ray, 2), "prims_per_ray": round(sprims_per_ray, 2),
                       "rays": int(st.shadow_rays)},
            "surface": {"hits": int(st.surface_timed_hits), "launches": int(st.surface_timed_launches), "hits_per_launch": round(hits_per_launch, 1), "avg_launch_ms": round(surf_avg_s * 1e3, 4)},
        }
        none = {"bound": "valu", "achieved": None, "peak": None, "unit": "G wave-instructions/s", "frac": None, "traffic": None,
                "note": "no committed counter passes for this workload / override: only event-timed figures"}
        roofline.update(valu_roofline(q["extend"], rays_per_launch, avg_launch_s, "ray", q, stale) if q and "extend" in q and avg_launch_s > 0 else none)
        # the surface kernel (closest-hit programs): the longest kernel of the Cornell boxes
        surf = dict(roofline["surface"], kernel="k_surface: light samples, BRDF sampling, throughput, roulette, queues of the next bounce")
        surf.update(valu_roofline(q["surface"], hits_per_launch, surf_avg_s, "hit", q, stale) if q and "surface" in q and surf_avg_s > 0 else none)
        roofline["surface"] = surf
        cpu = None
        if with_cpu_baseline:
            cpu = cpu_baseline(args, scene)
        host_ms = round(st.build_host_ms, 2)
        out = {
            "metric": "Mrays/s (paths x bounces)", "value": round(ext_rays / dt / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": 1 if shared else world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "the reference's own scene file (scenes/" + WORKLOADS[workload]["scene"] + ") at the config's size and sample count; nothing synthetic, no weights",
            "config": {"workload": WORKLOADS[workload]["name"] + (" [overridden]" if overridden else ""),
                       "width": W, "height": H, "spp_per_gpu": spp, "spp_total": spp * world, "semantics": "intended" if args.semantics else "literal",
                       "seeding": "counter", "parallelism": f"spp-split x{world}" + (" (rehearsal: all ranks on one GPU, gloo)" if shared else ""), "reduce": "rank-ordered gather + add" if args.ordered_reduce else "one sum-reduce to rank 0"},
            "total_mrays_per_s": round((ext_rays + shadow_rays) / dt / 1e6, 2),
            "shadow_mrays_per_s": round(shadow_rays / dt / 1e6, 2),
            "bounces_per_path": round(ext_rays / max(paths, 1), 3),
            # per step: the slowest and the fastest rank's render part and the slowest rank's reduce part (host clocks around
            # the two halves of a step, each ended by a stream synchronise)
            "phases_ms": {"render_max": round(float(tvals[1]) / steps * 1e3, 3), "render_min": round(float(tmin[0]) / steps * 1e3, 3),
                          "reduce_max": round(float(tvals[2]) / steps * 1e3, 3)},
            "batch": {"paths": int(st.batch_paths), "path_state_GB": round(st.path_state_bytes / 1e9, 2)},
            "bvh": {"builder": 3 if args.builder is None else args.builder, "nodes": int(st.bvh_nodes), "max_depth": int(st.bvh_max_depth), "build_ms": round(st.build_ms, 2),
                    "build_host_ms": host_ms, "built_on": "gpu" if host_ms == 0 else "gpu + host passes (build_host_ms of build_ms is host code)"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
    r.close()
    del sums
    torch.cuda.empty_cache()
    return out


def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launcher_selftest:
        return launcher_selftest(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    shared = args.share_gpu and world > 1
    if shared:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    cdev = "cpu" if shared else "cuda"  # where the small collectives' tensors live
    joined = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        j = torch.ones(1, dtype=torch.int64, device=cdev)
        dist.all_reduce(j)  # the ranks that really joined the group
        joined = int(j[0])
        if joined != world:
            raise SystemExit(f"{joined} of {world} ranks joined")

    out = measure(args, args.workload, args.steps, args.warmup, dist, rank, world, local_rank, shared, cdev,
                  with_cpu_baseline=not args.no_cpu_baseline and world == 1)
    # BASELINE.json quotes its metric on "Cornell-GGX + dragon": the default single-GPU run measures the other half too
    second = None
    if args.workload == "dragon" and world == 1 and not args.no_secondary and not (args.spp or args.size):
        second = measure(args, "cornell_ggx", max(args.secondary_steps, 1), 1, dist, rank, world, local_rank, shared, cdev, with_cpu_baseline=False)
    if rank == 0:
        if not shared:
            out["n_gpus"] = joined
        if second is not None:
            out["secondary"] = {"workload": second["config"]["workload"], "value": second["value"], "unit": second["unit"], "ms_per_step": second["ms_per_step"],
                                "steps": second["steps"], "warmup": second["warmup"], "total_mrays_per_s": second["total_mrays_per_s"],
                                "bounces_per_path": second["bounces_per_path"], "config": second["config"], "batch": second["batch"], "bvh": second["bvh"],
                                "roofline": dict(second["roofline"]["surface"], extend=dict((k, v) for k, v in second["roofline"].items() if k not in ("surface", "shadow")),
                                                 kernel_ms=second["roofline"]["kernel_ms"])}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def cpu_baseline(args, scene):
    """The CPU oracle on a bounded, strided sample of the same workload, on every core this job may use."""
    from oracle import oracle_py as orc
    nproc = os.cpu_count() or 1
    # the box exports its per-job CPU share (16 for one GPU) as OMP_NUM_THREADS; nproc counts the whole machine
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = min(nproc, share) if share > 0 else nproc
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
    return {"value": round(st.extension_rays / st.seconds / 1e6, 3), "unit": "Mrays/s", "cores": threads, "nproc": nproc, "kind": "port",
            "sample": ("every pixel" if stride == 1 else f"every {stride}th pixel in x and y") + f", {spp} spp, same scene and options; {st.extension_rays} extension + {st.shadow_rays} shadow rays in {st.seconds:.1f} s (incl. CPU BVH build); "
                      f"{threads} threads = this job's CPU share (OMP_NUM_THREADS) of the machine's {nproc} hardware threads"}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
