#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its headline config.

  metric   Msamples/s (W x H x spp / s), 720p masterchief scene @ 5000 spp, 4 bounces
           (BASELINE.json configs[3]; main.cpp:175-180).
  step     one full render of that frame: every rank renders its interleaved 8x8 tiles
           (srtRenderTiles), one gather of the tile buffers to rank 0 (N>1 only: srtGatherTiles, the
           library's own ncclGather call over RCCL, on the render's stream), rank 0 resolves to RGBA8
           (srtResolveTiles).  Scene and camera are resident in HBM
           before the timed region; the output stays on the device.
  N GPUs   python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...
           strong scaling: the frame is fixed, tiles shard across ranks.

Prints ONE JSON line on rank 0.  `roofline` names the roof that binds the workload:
  * the BASELINE configs are cache-resident (BVH 129 KB): bound = "valu-issue".  achieved = VALU
    wave-instructions per second (hardware counter SQ_INSTS_VALU of one launch / the kernel time measured
    live with HIP events), peak = CUs x 4 SIMDs x clock / 2 cycles per wave64 instruction
    (MI355X_MICROARCH.md: SIMD-32, 2 cycles), lane_utilisation = SQ_THREAD_CYCLES_VALU /
    (64 x SQ_ACTIVE_INST_VALU), frac = issue fraction x lane utilisation = useful lane-operations over
    the chip's lane-operation peak.  The algorithmic-bytes figure of SURVEY 8(d) is kept as
    `algorithmic_GBps` (it exceeds the HBM peak because the bytes come from L1/L2) and is never `frac`.
  * the synthetic soups (1-10 M triangles, far larger than the caches) are HBM-bound: bound = "hbm",
    achieved = algorithmic bytes / kernel time, frac = achieved / 8 TB/s, and the measured HBM side
    next to it (`traffic` bytes per launch, `hbm_measured_GBps`, `hbm_measured_frac`).
Counters are collected IN THIS RUN by child processes under `rocprofv3 --pmc` (separate passes for
FETCH_SIZE, WRITE_SIZE and the SQ group, never combined with tracing) before the parent touches the
GPU; if that is not possible (N > 1, already running under a profiler, rocprofv3 missing or failing)
the values recorded under profiles/ by an earlier run of the same kernel source are used and
`counters_source` says so.

The CPU oracle is used only for the `cpu_baseline` leg (a reported baseline on a bounded sample,
never the thing timed as `value`)."""
import argparse
import csv
import glob
import hashlib
import importlib
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
VALU_CYCLES_PER_WAVE_INSTR = 2.0  # wave64 on a SIMD-32 (MI355X_MICROARCH.md "v_fma_f32 (wave64) 2 cyc")
SIMDS_PER_CU = 4

DATA_NOTE = {
    "masterchief": "synthetic (assets/masterchief2 mesh + seeded procedural iron textures)",
    "army": "synthetic (five copies of the assets/masterchief2 mesh)",
    "iron": "synthetic (seeded procedural iron textures)",
    "spheres": "synthetic (procedural scene, no assets)",
    "sphere_field": "synthetic (procedural scene placed with the reference's generator, no assets)",
    "soup": "synthetic (seeded uniform triangle soup, seed 7)",
}

# name: (scene, W, H, spp, maxBounce, tree builder, traversal, binding roof)
WORKLOADS = {
    "masterchief_720p_5000spp": ("masterchief", 1280, 720, 5000, 4, "reference", "faithful", "valu-issue"),   # configs[3], the headline
    "iron_720p_5000spp": ("iron", 1280, 720, 5000, 4, "reference", "faithful", "valu-issue"),                 # configs[2]
    "spheres_720p_1024spp": ("spheres", 1280, 720, 1024, 8, "reference", "faithful", "valu-issue"),           # configs[1]
    "spheres_240p_64spp": ("spheres", 426, 240, 64, 8, "reference", "faithful", "valu-issue"),                # configs[0]
    "masterchief_1080p_8192spp": ("masterchief", 1920, 1080, 8192, 4, "reference", "faithful", "valu-issue"), # configs[4]
    "sphere_field_720p_1024spp": ("sphere_field", 1280, 720, 1024, 8, "reference", "faithful", "valu-issue"),  # main.cpp:92-122
    # a tree that is cache-resident but does not fit a CU's LDS (66 k nodes, 2 MB): the 256-thread kernel's regime, bound by
    # the vector-memory address unit (what treelets are for)
    "soup_50k_720p_64spp": ("soup:50000", 1280, 720, 64, 4, "reference", "faithful", "valu-issue"),
    "soup_200k_720p_64spp": ("soup:200000", 1280, 720, 64, 4, "reference", "faithful", "valu-issue"),
    # (five copies of the mesh: 16.7 k nodes; a 16 k-triangle soup: the 256-thread kernel on trees just beyond a CU's LDS)
    "army_720p_1024spp": ("army", 1280, 720, 1024, 4, "reference", "faithful", "valu-issue"),
    "soup_16k_720p_64spp": ("soup:16000", 1280, 720, 64, 4, "reference", "faithful", "valu-issue"),
    # HBM-bound points (SURVEY 8d "Synthetic"): the parity path (reference tree, bvh.h order) and the
    # fast mode (device-built PLOC tree, closest-hit traversal)
    "soup_1m_720p_16spp": ("soup:1000000", 1280, 720, 16, 4, "reference", "faithful", "hbm"),
    "soup_4m_720p_16spp": ("soup:4000000", 1280, 720, 16, 4, "reference", "faithful", "hbm"),
    "soup_10m_720p_16spp": ("soup:10000000", 1280, 720, 16, 4, "reference", "faithful", "hbm"),
    "soup_1m_ploc_closest_720p_16spp": ("soup:1000000", 1280, 720, 16, 4, "ploc", "closest", "hbm"),
    "soup_4m_ploc_closest_720p_16spp": ("soup:4000000", 1280, 720, 16, 4, "ploc", "closest", "hbm"),
    "soup_10m_ploc_closest_720p_16spp": ("soup:10000000", 1280, 720, 16, 4, "ploc", "closest", "hbm"),
}

PMC_PASSES = [
    ("fetch", ["FETCH_SIZE", "GRBM_GUI_ACTIVE"]),
    ("write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"]),
    ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
            "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_BUSY_CYCLES"]),
    # the vector-memory address unit (one TA per CU, in front of the vector L1): how busy the path of the node loads is
    ("ta", ["TA_BUSY_avr", "TA_BUFFER_READ_WAVEFRONTS_sum"]),
    ("ta_stall", ["TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"]),
]
PMC_OPTIONAL = ("ta", "ta_stall")  # a pass the profiler refuses on this box is left out, the others still count


def build_scene(srt, scene_name, builder):
    abi = srt.abi
    if scene_name.startswith("soup:"):
        n = int(scene_name.split(":")[1])
        b = {"reference": abi.SRT_BUILDER_REFERENCE, "lbvh": abi.SRT_BUILDER_LBVH, "ploc": abi.SRT_BUILDER_PLOC}[builder]
        return srt.scenes.scene_soup(n, seed=7, extent=6.0, size=max(0.01, 0.08 * (100000.0 / n) ** (1.0 / 3.0)), builder=b)
    return srt.scenes.SCENES[scene_name]()


def gather_tiles(local, rank, world):
    """torch.distributed form of the one gather, used by the CPU tests (gloo) and by rehearsals
    of N ranks on one GPU; on RCCL the bench calls the library's own srtGatherTiles instead (main()).
    local: (numLocalTiles, 64, 4).  Returns (world, numLocalTiles, 64, 4) on rank 0, None elsewhere."""
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sb, cam_params, abi, width, height, spp_full, max_bounce, seed, budget_s=20.0):
    """The CPU oracle (a port of the reference's loop) on this box's host cores, on a bounded sample
    of the same frame: all pixels at a reduced spp sized for ~budget_s of CPU work."""
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
            "sample": "%dx%d all pixels at %d spp of %d (%.1f s, OpenMP over rows, counter RNG)" % (
                width, height, spp, spp_full, dt),
            "cpu_model": cpu_model(), "compiler": "g++ -O2 -ffp-contract=off (oracle/Makefile)",
            "single_thread_value": round(single, 4),
            "single_thread_sample": "rows %d..%d at 2 spp" % rows}


# ---------------------------------------------------------------------------------------- counters
def kernel_source_id():
    """Identifies the kernel build the counters belong to: hash of the kernel sources and build flags."""
    h = hashlib.sha256()
    for rel in ("sexy-raytracer_amd/csrc/srt_kernels.hip", "sexy-raytracer_amd/csrc/srt_wavefront.hip", "sexy-raytracer_amd/csrc/srt_path.h",
                "sexy-raytracer_amd/csrc/srt_device.h", "sexy-raytracer_amd/csrc/Makefile"):
        try:
            h.update(open(os.path.join(ROOT, rel), "rb").read())
        except OSError:
            pass
    return h.hexdigest()[:16]


def pmc_file(workload):
    return os.path.join(ROOT, "profiles", "pmc_%s.json" % workload)


def under_profiler():
    env = os.environ
    return any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in env) or "rocprof" in env.get("LD_PRELOAD", "")


def collect_pmc(args, workload, spp, timeout_s):
    """Runs this script as `--pmc-child` under `rocprofv3 --pmc <group>` once per counter group and
    returns {counter: value per render-kernel launch}.  Must be called before this process touches the
    GPU (the children are separate processes; nothing is exec'ed from a GPU-initialised one)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        raise RuntimeError("rocprofv3 not found")
    env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_"))}
    env.pop("LD_PRELOAD", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["TMPDIR"] = "/tmp"
    out = {}
    tmp = tempfile.mkdtemp(prefix="srt_pmc_", dir="/tmp")
    try:
        for name, counters in PMC_PASSES:
            d = os.path.join(tmp, name)
            cmd = [rocprof, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", sys.executable,
                                                  os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", workload,
                                                  "--spp", str(spp), "--seed", str(args.seed), "--spp-chunks", str(args.spp_chunks)]
            # its own process group: on a timeout the whole group goes (rocprofv3's python child holds the GPU), and is
            # waited for, before this process touches the GPU
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                out_text, _ = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.communicate()
                raise RuntimeError("rocprofv3 pass '%s' timed out after %.0f s (process group killed)" % (name, timeout_s))
            r = subprocess.CompletedProcess(cmd, proc.returncode, out_text)
            if r.returncode != 0:
                if name in PMC_OPTIONAL:
                    sys.stderr.write("bench.py: optional counter pass '%s' failed, left out\n" % name)
                    continue
                raise RuntimeError("rocprofv3 pass '%s' failed (rc %d): %s" % (name, r.returncode, r.stdout.decode(errors="replace")[-400:]))
            # the child launches the render kernel twice (a 1-spp warm-up, then the frame): the LAST dispatch counts
            per_dispatch = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if "srt_render_" not in row["Kernel_Name"]:
                        continue
                    disp = int(row.get("Dispatch_Id", 0) or 0)
                    vals = per_dispatch.setdefault(disp, {})
                    vals[row["Counter_Name"]] = vals.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    if name == "fetch" and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                        # the profiled dispatch's own duration: what GRBM_GUI_ACTIVE of the same dispatch is divided by
                        vals["_kernel_ms"] = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6
            if not per_dispatch:
                if name in PMC_OPTIONAL:
                    continue
                raise RuntimeError("rocprofv3 pass '%s' recorded no render-kernel dispatch" % name)
            out.update(per_dispatch[max(per_dispatch)])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def pmc_child(args):
    """One launch of the workload's render kernel (after one untimed warm-up launch at 1 spp), for the
    profiler wrapped around this process to count."""
    import torch
    srt = importlib.import_module("sexy-raytracer_amd")
    abi, dev = srt.abi, srt.device()
    scene_name, W, H, spp, max_bounce, builder, traversal, _ = WORKLOADS[args.workload]
    if args.spp > 0:
        spp = args.spp
    trav = abi.SRT_TRAVERSE_CLOSEST if traversal == "closest" else abi.SRT_TRAVERSE_FAITHFUL
    ctx = dev.Context(0)
    ctx.upload_scene(build_scene(srt, scene_name, builder))
    ctx.set_camera(dev.make_camera(abi.default_camera_params()))
    local = torch.zeros((dev.num_local_tiles(W, H, 1), 64, 4), dtype=torch.float32, device="cuda")
    chunks = max(1, min(args.spp_chunks, spp)) if args.spp_chunks > 0 else 0
    warm = abi.default_render_params(W, H, 1, max_bounce, seed=args.seed, spp_chunks=1, traversal=trav)
    ctx.render_tiles(warm, local.data_ptr(), None)  # code object loaded, LDS tree path exercised: not the dispatch that counts
    torch.cuda.synchronize()
    p = abi.default_render_params(W, H, spp, max_bounce, seed=args.seed, spp_chunks=chunks, traversal=trav)
    ctx.render_tiles(p, local.data_ptr(), None)
    torch.cuda.synchronize()
    ctx.close()


def roofline_block(bound, pmc, pmc_source, avg_kernel_ms, alg_bytes_per_launch, bytes_per_sample, st, info, scene_footprint,
                   kernel_name):
    alg_gbps = alg_bytes_per_launch / (avg_kernel_ms * 1e-3) / 1e9
    common = {"kernel": kernel_name, "kernel_ms_avg": round(avg_kernel_ms, 3),
              "algorithmic_bytes_per_sample": round(bytes_per_sample, 2), "algorithmic_GBps": round(alg_gbps, 1),
              "rays_per_sample": round(st["rays"] / max(1, st["samples"]), 4),
              "node_visits_per_ray": round(st["nodeVisits"] / max(1, st["rays"]), 3),
              "prim_tests_per_ray": round((st["triTests"] + st["sphereTests"]) / max(1, st["rays"]), 3),
              "counters_source": pmc_source, "scene_footprint": scene_footprint}
    traffic = None
    if pmc and "FETCH_SIZE" in pmc:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  gfx950 tallies a 128-B read request at 64 B
        # (MI355X_MICROARCH.md, HBM): the streaming-read correction is x2; 32-B node gathers are narrower
        # than that, so both figures are kept.
        fetch, write = pmc["FETCH_SIZE"] * 1024.0, pmc.get("WRITE_SIZE", 0.0) * 1024.0
        traffic = int(2 * fetch + write)
        common["traffic_uncorrected"] = int(fetch + write)
        common["traffic_note"] = ("FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B) + WRITE_SIZE; "
                                  "traffic_uncorrected = FETCH_SIZE + WRITE_SIZE as read")
    if bound == "hbm":
        # achieved = the HBM bytes the counters saw move (FETCH_SIZE + WRITE_SIZE as read: random 32-64-byte gathers are not
        # the 128-byte streaming requests the x2 correction is for, profiles/r02/fetch_size_calibration_*) over the kernel
        # time; frac = achieved / peak, never above 1.  The algorithmic figure (SURVEY 8d bytes, part of them served by the
        # caches) stays next to it and is not a fraction of the HBM roof.
        r = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": traffic,
             "algorithmic_over_peak": round(alg_gbps / HBM_PEAK_GBS, 4)}
        if traffic is not None:
            lo = common["traffic_uncorrected"] / (avg_kernel_ms * 1e-3) / 1e9
            hi = traffic / (avg_kernel_ms * 1e-3) / 1e9
            r["achieved"] = round(lo, 1)
            r["frac"] = round(lo / HBM_PEAK_GBS, 4)
            r["hbm_measured_GBps"] = round(lo, 1)
            r["hbm_measured_frac"] = round(lo / HBM_PEAK_GBS, 4)
            r["hbm_measured_GBps_x2"] = round(hi, 1)         # with the streaming-read doubling: an upper bound
        r.update(common)
        return r
    # cache-resident: VALU issue x lane utilisation
    r = {"bound": "valu-issue", "achieved": None, "peak": None, "unit": "G wave-instr/s", "frac": None, "traffic": traffic}
    if pmc and "SQ_INSTS_VALU" in pmc:
        clock_ghz = info["clock_mhz"] / 1e3
        if "GRBM_GUI_ACTIVE" in pmc:  # summed over the 8 XCDs
            clock_ghz = pmc["GRBM_GUI_ACTIVE"] / 8.0 / (pmc.get("_kernel_ms", avg_kernel_ms) * 1e-3) / 1e9
            if not (1.0 < clock_ghz < 3.0):
                clock_ghz = info["clock_mhz"] / 1e3
        peak = info["cus"] * SIMDS_PER_CU * clock_ghz / VALU_CYCLES_PER_WAVE_INSTR
        achieved = pmc["SQ_INSTS_VALU"] / (avg_kernel_ms * 1e-3) / 1e9
        lane = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
        r.update({"achieved": round(achieved, 2), "peak": round(peak, 2), "issue_frac": round(achieved / peak, 4),
                  "lane_utilisation": round(lane, 4), "frac": round(achieved / peak * lane, 4),
                  "clock_ghz": round(clock_ghz, 3),
                  "valu_wave_instr_per_sample": round(pmc["SQ_INSTS_VALU"] / max(1, pmc.get("_samples", 1)), 2),
                  "wait_frac": round(pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_ANY" in pmc else None,
                  "salu_per_valu": round(pmc["SQ_INSTS_SALU"] / pmc["SQ_INSTS_VALU"], 4) if "SQ_INSTS_SALU" in pmc else None})
    if pmc and "SQ_INSTS_VMEM_RD" in pmc:
        # The other unit this kernel leans on: every node visit is two divergent 16-byte loads per lane through the
        # CU's vector L1.  tools/ubench_tcp.hip (profiles/r02/ubench_tcp.txt) prices such a wave-instruction at
        # 16 + 0.45 x (cache lines touched) cycles of that path; here is what the frame leaves per instruction.
        ghz = r.get("clock_ghz") or info["clock_mhz"] / 1e3  # the clock measured above when the SQ pass has it
        cu_cycles = info["cus"] * (pmc.get("_kernel_ms", avg_kernel_ms) * 1e-3) * ghz * 1e9
        l1 = {"vmem_rd_wave_instr": int(pmc["SQ_INSTS_VMEM_RD"]),
              "cu_cycles_per_vmem_rd_instr": round(cu_cycles / pmc["SQ_INSTS_VMEM_RD"], 2),
              "ubench_cycles_per_divergent_dwordx4": {"64 lanes": 38.7, "38 lanes": 29.3, "16 lanes": 22.2,
                                                       "source": "profiles/r02/ubench_tcp.txt, 16 KB table (L1 hits)"}}
        if "TA_BUSY_avr" in pmc:
            l1["ta_busy_frac"] = round(pmc["TA_BUSY_avr"] / (cu_cycles / info["cus"]), 4)
        if "TA_ADDR_STALLED_BY_TC_CYCLES_sum" in pmc:
            l1["ta_addr_stalled_by_l1_frac"] = round(pmc["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / cu_cycles, 4)
        if "TA_DATA_STALLED_BY_TC_CYCLES_sum" in pmc:
            l1["ta_data_stalled_by_l1_frac"] = round(pmc["TA_DATA_STALLED_BY_TC_CYCLES_sum"] / cu_cycles, 4)
        r["l1_path"] = l1
    if traffic is not None:
        r["hbm_measured_frac"] = round(traffic / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    r["note"] = ("frac = VALU issue fraction x lane utilisation (useful lane-operations / chip lane-operation peak); "
                 "the scene is cache-resident, algorithmic_GBps is served by L1/L2 and is not an HBM fraction")
    r.update(common)
    return r


def hbm_point(args):
    """The BASELINE scenes are cache-resident, so the headline line says nothing about HBM.  This adds one HBM-bound point
    to the same JSON line, measured in the same run: a seeded 4 M-triangle soup (far beyond the 256 MB of cache), tree
    built on the device, closest-hit traversal -- the fast mode for such scenes, not a parity path -- rendered by a child
    run of this script (own counters, own HIP-event timing) after this process has released the GPU's queues."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", args.hbm_point_workload, "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--seed", str(args.seed)]
    if args.no_pmc:
        cmd.append("--no-pmc")
    try:
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, start_new_session=True)
        rec = json.loads(r.stdout.decode().strip().split("\n")[-1])
        roof = rec["roofline"]
        return {"workload": rec["config"]["workload"], "value": rec["value"], "unit": rec["unit"], "kernel_ms_avg": roof["kernel_ms_avg"],
                "tree": rec["config"]["tree"], "traversal": rec["config"]["traversal"],
                "roofline": {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_uncorrected", "hbm_measured_GBps_x2",
                                                      "algorithmic_GBps", "algorithmic_over_peak", "node_visits_per_ray", "counters_source", "scene_footprint")}}
    except Exception as e:  # the headline line stands without it
        return {"error": "%s: %s" % (type(e).__name__, e)}


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
                         "0 = the library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-pmc", action="store_true", help="do not collect hardware counters in this run (use profiles/)")
    ap.add_argument("--pmc-timeout", type=float, default=240.0)
    ap.add_argument("--save-pmc", action="store_true", help="record the counters of this run under profiles/pmc_<workload>.json")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--save-png", default="")
    ap.add_argument("--no-hbm-point", action="store_true",
                    help="headline workload only: do not add the HBM-bound point (4 M-triangle soup, device-built tree, "
                         "closest-hit traversal) measured in a child run of this script")
    ap.add_argument("--hbm-point-workload", default="soup_4m_ploc_closest_720p_16spp")
    args = ap.parse_args()

    # before anything initialises HIP/HSA in this process: dmabuf IPC for RCCL
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.pmc_child:
        return pmc_child(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    scene_name, W, H, spp, max_bounce, builder, traversal, bound = WORKLOADS[args.workload]
    if args.spp > 0:
        spp = args.spp

    # ---- hardware counters of one launch of this workload, collected by child processes before this
    # process initialises the GPU
    pmc, pmc_source = None, "none"
    ksrc = kernel_source_id()
    if world == 1 and not args.no_pmc and not under_profiler():
        try:
            t = time.time()
            pmc = collect_pmc(args, args.workload, spp, args.pmc_timeout)
            pmc_source = "in-run: rocprofv3 --pmc, %d separate passes, one launch each (%.0f s)" % (len(PMC_PASSES), time.time() - t)
        except Exception as e:  # fall back to the recorded counters, and say so
            sys.stderr.write("bench.py: in-run counter collection failed: %s\n" % e)
            pmc = None
    if pmc is None:
        try:
            rec = json.load(open(pmc_file(args.workload)))
            if rec.get("spp") == spp and world == 1:
                pmc = rec["counters"]
                same = rec.get("kernel_source_id") == ksrc
                pmc_source = "from %s (recorded %s; kernel source %s this build)" % (
                    os.path.relpath(pmc_file(args.workload), ROOT), rec.get("recorded", "?"), "matches" if same else "DIFFERS from")
        except (OSError, ValueError, KeyError):
            pmc = None

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback for the hot path)")
    backend = os.environ.get("SRT_BENCH_BACKEND", "nccl")  # nccl = RCCL; gloo only to rehearse N ranks on one GPU
    if backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    srt = importlib.import_module("sexy-raytracer_amd")
    abi, dev = srt.abi, srt.device()
    trav = abi.SRT_TRAVERSE_CLOSEST if traversal == "closest" else abi.SRT_TRAVERSE_FAITHFUL

    ctx = dev.Context(local_rank)
    info = ctx.device_info()
    t_build = time.time()
    sb = build_scene(srt, scene_name, builder)
    ctx.upload_scene(sb)  # scene resident in HBM before the timed region
    t_build = time.time() - t_build
    n_tris = int(sb.desc().numTriangles)
    n_nodes = len(ctx.bvh(0)) if n_tris < 2000000 else 2 * n_tris
    scene_footprint = "BVH %.0f KB, %d triangles %.0f KB, %d spheres, textures %.1f MB" % (
        n_nodes * 32 / 1e3, n_tris, n_tris * 112 / 1e3, len(sb.spheres), len(sb.texels) / 1e6)
    cam_params = abi.default_camera_params()
    ctx.set_camera(dev.make_camera(cam_params))

    nloc = dev.num_local_tiles(W, H, world)
    local = torch.zeros((nloc, 64, 4), dtype=torch.float32, device="cuda")
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") if rank == 0 else None
    chunks = dev.plan_spp_chunks(W, H, spp, max(1, min(args.spp_chunks, spp)) if args.spp_chunks > 0 else 0)  # what the render will use
    params = abi.default_render_params(W, H, spp, max_bounce, seed=args.seed, tile_first=rank, tile_stride=world,
                                       spp_chunks=chunks if args.spp_chunks > 0 else 0, traversal=trav)  # 0 = library plan
    stream = torch.cuda.current_stream().cuda_stream
    kernel_ms = []
    native_gather = world > 1 and backend == "nccl"
    gathered_buf = None
    native_note = ""
    if native_gather:
        # the C-ABI's own communicator: rank 0 makes the id, torch.distributed (already up for the barrier
        # and the timing reduction) only carries its 128 bytes to the other ranks
        box = [dev.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ok = 1
        try:
            ctx.comm_init(box[0], world, rank)
        except Exception as e:  # every rank must take the same path: agree below
            ok, native_note = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            native_gather = False  # fall back to torch.distributed's gather (same bytes, same order) and say so
            native_note = native_note or "srtCommInit failed on another rank"
            sys.stderr.write("bench.py rank %d: native RCCL communicator unavailable (%s); using dist.gather\n" % (rank, native_note))
        elif rank == 0:
            gathered_buf = torch.empty((world, nloc, 64, 4), dtype=torch.float32, device="cuda")

    launch = {"lds_tree": False, "lds_tree_mode": 0, "workgroups": 0, "threads": 0, "lds_bytes": 0}

    def step(record):
        ctx.render_tiles(params, local.data_ptr(), stream)
        if record:
            kernel_ms.append(ctx.last_kernel_ms())  # HIP events on the launch stream, around the render kernel
            launch.update(ctx.launch_info())        # which variant of the render kernel these launches were
        if native_gather:
            ctx.gather_tiles(params, local.data_ptr(), gathered_buf.data_ptr() if rank == 0 else None, stream)
            gathered = gathered_buf
        else:
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
                                        count_stats=1, traversal=trav)
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
        if pmc is not None:
            pmc = dict(pmc)
            pmc["_samples"] = samples_per_launch
        total_samples = W * H * spp * args.steps
        value = total_samples / elapsed / 1e6
        mode = launch.get("lds_tree_mode")
        # (the bench scenes have one root; the hybrid form's single-root instance serves trees of up to 2^20 nodes: srt_launch_render_wf)
        kernel_name = ("srt_render_wf_kernel<%s,false,true>" % ("true" if n_nodes <= (1 << 20) else "false") if mode == 4 else
                       "srt_render_wf_kernel<true,false,false>" if mode == 3 else
                       "srt_render_kernel<%s,false,true,%s,%s>" % ("true" if traversal == "closest" else "false",
                                                                   "true" if launch["lds_tree"] else "false",
                                                                   "true" if launch.get("lds_tree_mode") == 2 else "false"))
        line = {
            "metric": "Msamples/s (WxHxspp/s), 720p masterchief @5k spp" if args.workload == "masterchief_720p_5000spp"
                      else "Msamples/s (WxHxspp/s), " + args.workload,
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": DATA_NOTE.get(scene_name.split(":")[0], "synthetic"),
            "config": {"workload": args.workload, "scene": scene_name, "width": W, "height": H, "spp": spp,
                       "max_bounce": max_bounce, "seed": args.seed, "spp_chunks": chunks,
                       "tree": builder, "traversal": "faithful (bvh.h order)" if traversal == "faithful" else "closest hit (not the parity path)",
                       "scene_build_upload_s": round(t_build, 2),
                       "parallelism": "tiles8x8 interleaved over %d rank(s), 1 gather (%s)" % (
                           world, "srtGatherTiles: ncclGather" if native_gather else ("none" if world == 1 else "torch.distributed " + backend
                                                                                     + (" (native communicator unavailable)" if native_note else "")))},
            "device": info,
            "launch": {"workgroups": launch["workgroups"], "threads_per_workgroup": launch["threads"], "lds_bytes_per_workgroup": launch["lds_bytes"],
                       "node_records": ("the tree's top (largest boxes) in every CU's LDS, the rest through the vector L1 / L2" if mode == 4 else
                                        "LDS-resident (whole node array in every CU's LDS)" if launch["lds_tree"] else "through the vector L1 / L2 / HBM"),
                       "kernel_form": ("path pool: lanes traverse, full waves shade contexts from per-class LDS rings (srt_wavefront.hip)"
                                       + (", hybrid form" if mode == 4 else "") if mode in (3, 4) else "step scheduler: one path per lane (srt_kernels.hip)"),
                       "attenuation_stacks": "global memory" if mode in (1, 3, 4) else "LDS"},
            "roofline": roofline_block(bound, pmc, pmc_source, avg_kernel_ms, bytes_per_launch, bytes_per_sample, st, info,
                                       scene_footprint, kernel_name),
        }
        if args.save_pmc and pmc is not None and pmc_source.startswith("in-run"):
            rec = {"workload": args.workload, "spp": spp, "spp_chunks": chunks, "kernel_source_id": ksrc,
                   "recorded": time.strftime("%Y-%m-%d"), "device": info, "kernel_ms_at_recording": round(avg_kernel_ms, 3),
                   "counters": {k: v for k, v in pmc.items() if not k.startswith("_")},
                   "how": "bench.py --save-pmc: rocprofv3 --pmc, separate passes %s" % [c for _, c in PMC_PASSES]}
            os.makedirs(os.path.dirname(pmc_file(args.workload)), exist_ok=True)
            json.dump(rec, open(pmc_file(args.workload), "w"), indent=1)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sb, cam_params, abi, W, H, spp, max_bounce, args.seed, args.cpu_budget)
        else:
            line["cpu_baseline"] = None
        if args.workload == "masterchief_720p_5000spp" and world == 1 and not args.no_hbm_point and not under_profiler():
            line["hbm_point"] = hbm_point(args)
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
