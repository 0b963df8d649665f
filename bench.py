#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its headline config.

  metric   Msamples/s (W x H x spp / s), 720p masterchief scene @ 5000 spp, 4 bounces
           (BASELINE.json configs[3]; main.cpp:175-180).
  step     one full render of that frame: every rank renders its interleaved 8x8 tiles
           (srtRenderTiles), one gather of the tile buffers to rank 0 (RCCL, N>1 only),
           rank 0 resolves to RGBA8 (srtResolveTiles).  Scene and camera are resident in HBM
           before the timed region; the output stays on the device.
  N GPUs   python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...
           strong scaling: the frame is fixed, tiles shard across ranks.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).  The CPU oracle is used
only for the `cpu_baseline` leg (a reported baseline on a bounded sample, never the thing timed
as `value`)."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)

DATA_NOTE = {
    "masterchief": "synthetic (assets/masterchief2 mesh + seeded procedural iron textures)",
    "iron": "synthetic (seeded procedural iron textures)",
    "spheres": "synthetic (procedural scene, no assets)",
    "sphere_field": "synthetic (procedural scene placed with the reference's generator, no assets)",
}

WORKLOADS = {
    # name: (scene, W, H, spp, maxBounce)
    "masterchief_720p_5000spp": ("masterchief", 1280, 720, 5000, 4),   # configs[3], the headline
    "iron_720p_5000spp": ("iron", 1280, 720, 5000, 4),                 # configs[2]
    "spheres_720p_1024spp": ("spheres", 1280, 720, 1024, 8),           # configs[1]
    "spheres_240p_64spp": ("spheres", 426, 240, 64, 8),                # configs[0]
    "masterchief_1080p_8192spp": ("masterchief", 1920, 1080, 8192, 4), # configs[4]
    "sphere_field_720p_1024spp": ("sphere_field", 1280, 720, 1024, 8),  # not a BASELINE config: main.cpp:92-122
}


def gather_tiles(local, rank, world):
    """One gather of equal-sized tile buffers to rank 0.  local: (numLocalTiles, 64, 4).
    Returns (world, numLocalTiles, 64, 4) on rank 0, None elsewhere.  Backend-agnostic (nccl = RCCL
    on the GPU box, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local.unsqueeze(0)
    if dist.get_backend() == "gloo" and local.is_cuda:
        # rehearsal only (several ranks sharing one GPU, SRT_BENCH_BACKEND=gloo): gloo gathers host tensors
        host = gather_tiles(local.cpu(), rank, world)
        return host.to(local.device) if rank == 0 else None
    if rank == 0:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, [out[i] for i in range(world)], dst=0)
        return out
    dist.gather(local, None, dst=0)
    return None


def algorithmic_bytes(stats, width, height):
    """32 B per BVH node visit, 48 B per triangle test, 16 B per sphere test,
    24 B of triangle UVs per shaded triangle hit, 4 B per texel fetch, 16 B per pixel written once."""
    return (32 * stats["nodeVisits"] + 48 * stats["triTests"] + 16 * stats["sphereTests"]
            + 24 * stats["shadedTriHits"] + 4 * stats["texelFetches"] + 16 * width * height)


def cpu_baseline(sb, cam_params, abi, width, height, spp_full, max_bounce, seed, budget_s=20.0):
    """The CPU oracle (a port of the reference's loop) on this box's host cores, on a bounded sample
    of the same frame: all 1280x720 pixels at a reduced spp sized for ~budget_s of CPU work."""
    import oracle.oracle_py as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # the box's CPU share for one GPU
    osc = O.OracleScene(sb)
    cam = O.make_camera(cam_params)
    # single thread: what the reference does (it has no threads), a few rows worth
    rows = (height // 2 - 4, height // 2 + 4)
    p = abi.default_render_params(width, height, 2, max_bounce, seed=seed)
    t = time.time()
    osc.render(cam, p, O.RNG_COUNTER, threads=1, rows=rows, want_rgba=False, want_stats=False)
    single = (rows[1] - rows[0]) * width * 2 / (time.time() - t) / 1e6
    # calibrate, then the bounded sample
    p = abi.default_render_params(width, height, 1, max_bounce, seed=seed)
    t = time.time()
    osc.render(cam, p, O.RNG_COUNTER, threads=cores, want_rgba=False, want_stats=False)
    dt1 = time.time() - t
    spp = int(max(1, min(spp_full, budget_s / max(dt1, 1e-3))))
    p = abi.default_render_params(width, height, spp, max_bounce, seed=seed)
    t = time.time()
    osc.render(cam, p, O.RNG_COUNTER, threads=cores, want_rgba=False, want_stats=False)
    dt = time.time() - t
    return {"value": round(width * height * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%dx%d all pixels at %d spp of %d (%.1f s, OpenMP over rows, counter RNG, g++ -O2)" % (
                width, height, spp, spp_full, dt),
            "single_thread_value": round(single, 4),
            "single_thread_sample": "rows %d..%d at 2 spp" % rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="masterchief_720p_5000spp", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (0 = the workload's)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--spp-chunks", type=int, default=0,
                    help="work items per pixel: samples of a pixel are summed in index order inside a chunk and the "
                         "chunk sums in chunk order (1 = the reference's single running sum, main.cpp:217; "
                         "0 = the library default, ~32 samples per item)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--save-png", default="")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback for the hot path)")
    backend = os.environ.get("SRT_BENCH_BACKEND", "nccl")  # nccl = RCCL; gloo only to rehearse N ranks on one GPU
    if backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    srt = importlib.import_module("sexy-raytracer_amd")
    abi, dev = srt.abi, srt.device()
    scene_name, W, H, spp, max_bounce = WORKLOADS[args.workload]
    if args.spp > 0:
        spp = args.spp

    ctx = dev.Context(local_rank)
    sb = srt.scenes.SCENES[scene_name]()
    ctx.upload_scene(sb)  # scene resident in HBM before the timed region
    n_tris = int(sb.desc().numTriangles)
    scene_footprint = "BVH %.0f KB, %d triangles %.0f KB, %d spheres, textures %.1f MB" % (
        len(ctx.bvh(0)) * 32 / 1e3, n_tris, n_tris * 112 / 1e3, len(sb.spheres), len(sb.texels) / 1e6)
    cam_params = abi.default_camera_params()
    ctx.set_camera(dev.make_camera(cam_params))

    nloc = dev.num_local_tiles(W, H, world)
    local = torch.zeros((nloc, 64, 4), dtype=torch.float32, device="cuda")
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") if rank == 0 else None
    chunks = max(1, min(args.spp_chunks, spp)) if args.spp_chunks > 0 else dev.default_spp_chunks(spp)
    params = abi.default_render_params(W, H, spp, max_bounce, seed=args.seed, tile_first=rank, tile_stride=world,
                                       spp_chunks=chunks if args.spp_chunks > 0 else 0)  # 0 = library plan
    stream = torch.cuda.current_stream().cuda_stream
    kernel_ms = []

    def step(record):
        ctx.render_tiles(params, local.data_ptr(), stream)
        if record:
            kernel_ms.append(ctx.last_kernel_ms())  # HIP events on the launch stream, around the render kernel
        gathered = gather_tiles(local, rank, world)
        if rank == 0:
            ctx.resolve_tiles(params, gathered.data_ptr(), rgba.data_ptr(), None, stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        red_dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([sum(kernel_ms) / max(1, len(kernel_ms))], dtype=torch.float64, device=red_dev)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        avg_kernel_ms = float(k.item())
    else:
        avg_kernel_ms = sum(kernel_ms) / max(1, len(kernel_ms))

    # algorithmic bytes per sample: the counting variant of the same kernel on the same scene and
    # seed at a reduced spp (outside the timed region); the traversal is identical sample for sample.
    count_spp = min(spp, 8)
    cparams = abi.default_render_params(W, H, count_spp, max_bounce, seed=args.seed, tile_first=rank, tile_stride=world,
                                        count_stats=1)
    ctx.render_tiles(cparams, local.data_ptr(), stream)
    torch.cuda.synchronize()
    st = ctx.stats()
    if world > 1:
        keys = sorted(st)
        v = torch.tensor([st[k] for k in keys], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(v)
        st = dict(zip(keys, [int(x) for x in v.tolist()]))
    if rank == 0:
        samples_counted = W * H * count_spp
        bytes_per_sample = (algorithmic_bytes(st, W, H) - 16 * W * H) / samples_counted
        samples_per_launch = W * H * spp  # whole frame; per rank it is 1/world of this
        bytes_per_launch = (bytes_per_sample * samples_per_launch + 16 * W * H) / world
        achieved = bytes_per_launch / (avg_kernel_ms * 1e-3) / 1e9
        total_samples = W * H * spp * args.steps
        value = total_samples / elapsed / 1e6
        line = {
            "metric": "Msamples/s (WxHxspp/s), 720p masterchief @5k spp" if args.workload == "masterchief_720p_5000spp"
                      else "Msamples/s (WxHxspp/s), " + args.workload,
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": DATA_NOTE.get(scene_name, "synthetic"),
            "config": {"workload": args.workload, "scene": scene_name, "width": W, "height": H, "spp": spp,
                       "max_bounce": max_bounce, "seed": args.seed, "spp_chunks": chunks, "traversal": "faithful (bvh.h order)",
                       "parallelism": "tiles8x8 interleaved over %d rank(s), 1 gather" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel": "srt_render_kernel<false,false,true>", "kernel_ms_avg": round(avg_kernel_ms, 3),
                         "algorithmic_bytes_per_sample": round(bytes_per_sample, 2),
                         "rays_per_sample": round(st["rays"] / st["samples"], 4),
                         "node_visits_per_ray": round(st["nodeVisits"] / st["rays"], 3),
                         "prim_tests_per_ray": round((st["triTests"] + st["sphereTests"]) / st["rays"], 3),
                         "note": "scene is cache-resident (%s): achieved = algorithmic bytes / kernel time, "
                                 "not HBM traffic" % scene_footprint},
        }
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                tf = json.load(open(traffic_file))
                if tf.get("workload") == args.workload and tf.get("spp") == spp and world == 1:
                    line["roofline"]["traffic"] = tf["hbm_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = tf.get("source", "profiles/traffic.json")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sb, cam_params, abi, W, H, spp, max_bounce, args.seed, args.cpu_budget)
        else:
            line["cpu_baseline"] = None
        if args.save_png:
            from PIL import Image
            Image.fromarray(rgba.cpu().numpy()).save(args.save_png)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
