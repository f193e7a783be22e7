#!/usr/bin/env python3
"""bench.py -- frames/s of the MaskFusion::processFrame hot path on MI355X (driver contract: see task statement).

Workloads (BASELINE.json `configs`):
  --config 1  (default, the metric's) synthetic 640x480 RGB-D stream S1 (600 frames, Kinect-like noise), one
              background model (-static), geometric ICP (icpWeight = 100) + surfel fusion, precomputed empty masks.
  --config 2s S2 on ONE GPU: the same room with 8 rigid moving objects and their instance masks; background + 8 object
              models, trackAllModels, geometric ICP -- the multi-model path (global projection, label stage, per-model fusion)
              with the Gauss-Newton loops of all 9 models batched into one launch per iteration.
  --config 3  configs[3], ONE scene sharded BY MODEL over the N ranks (maskfusion_amd/sharded.py): S2's 8 objects, rank 0
              owns the background + the label stage, object models live on the other ranks; per frame: frame broadcast, all-reduce(MIN)
              of the projection keys, gather of per-model state, broadcast of the label image + background pose + control record.
              STRONG scaling: value = scene frames/s.  (N = 1: every model on rank 0 through the same three phases.)
  --config 4  the per-GPU share of configs[4]: 1280x960 stream, NUM_GSURFELS = 32M.
A "step" = one processFrame over one frame whose rgb / depth (/ mask) already sit in HBM.  The timed region is `--steps` steps,
repeated as a whole until it has lasted at least `--min-seconds` (a 6 ms region says little); `steps` in the JSON line is the
number of steps actually timed.  N > 1 (weak scaling): every rank owns one surfel model (the reference's per-model independence,
SURVEY.md 8e) and tracks / fuses it against the frame rank 0 broadcasts over RCCL each step; value = model-frames of all ranks
/ max-over-ranks time.

The JSON line also carries
  roofline       -- the dominant kernel (the ICP Gauss-Newton iteration): algorithmic bytes per launch / average launch duration
                    measured here with HIP events on the library's stream, against the 8 TB/s HBM peak; `traffic` = HBM bytes
                    per launch from the newest committed PMC profile (profiles/r*_pmc.json) when it was taken on this kernel;
                    `levels` = the same interval per rocprofv3 kernel name (level 0 / coarse levels); `measured_ceiling` = the same
                    figure over the bandwidth a device-to-device copy reaches on this box;
  roofline_frame -- the same for the whole frame: (741 P + 192 N) algorithmic bytes / frame time;
  host_input     -- frames/s through mf_process_frame (host pointers: 2.15 MB of H2D per frame + one sync), beside `value`;
  cpu_baseline   -- the oracle (CPU restatement, oracle/; a -O3 -march=native timing build) on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

CONFIGS = {
    "1": dict(W=640, H=480, f=528.0, surfels=9437184, n_objects=0, frames=600,
              workload="configs[1]: synthetic 640x480 RGB-D stream (S1: 600 frames, Kinect-like noise), 1 background model per GPU, "
                       "icpWeight=100 (geometric ICP 4/5/10 iterations) + surfel fusion, empty masks"),
    "2s": dict(W=640, H=480, f=528.0, surfels=9437184, n_objects=8, frames=120,
               workload="S2 on one GPU (SURVEY.md 8d): synthetic 640x480 RGB-D stream, 8 rigid moving objects with instance masks, "
                        "background + 8 object models, trackAllModels, icpWeight=100, global projection + label stage + per-model fusion"),
    "3": dict(W=640, H=480, f=528.0, surfels=9437184, n_objects=8, frames=120,
              workload="configs[3]: S2 synthetic 640x480 RGB-D stream with 8 rigid moving objects, ONE scene sharded by model over the ranks "
                       "(rank 0: background + label stage; objects on the other ranks), trackAllModels, icpWeight=100"),
    "4": dict(W=1280, H=960, f=1056.0, surfels=32 * 1024 * 1024, osurfels=4 * 1024 * 1024, n_objects=4, frames=40, s3=True,
              workload="configs[4] on one GPU (SURVEY.md 8d S3): synthetic 1280x960 RGB-D stream with 4 instance-masked objects, MASKFUSION_NUM_GSURFELS=32M / "
                       "NUM_OSURFELS=4M, every map pre-filled to >= 80 % of its capacity (26.5 M background surfels, 3.4 M per object model: generated on "
                       "the scene's surfaces and loaded with Model.uploadMap, maskfusion_amd/stress.py), icpWeight=100, global projection + label stage + "
                       "per-model fusion"),
    "4n": dict(W=1280, H=960, f=1056.0, surfels=32 * 1024 * 1024, n_objects=0, frames=200,
               workload="1280x960 stream, natural map: synthetic 1280x960 RGB-D stream (S3 scaling of S1), 1 background model, NUM_GSURFELS=32M budget "
                        "(the map grows to its natural size, ~1.4 M surfels), icpWeight=100 + surfel fusion, empty masks"),
}


def _stream(cfg):
    from maskfusion_amd import synth
    if cfg.get("s3"):        # configs[4]'s scene: maskfusion_amd/stress.py
        from maskfusion_amd import stress
        return stress.stream(cfg["n_objects"])
    return synth.Stream(W=cfg["W"], H=cfg["H"], fx=cfg["f"], fy=cfg["f"], cx=cfg["W"] / 2.0, cy=cfg["H"] / 2.0, noise=True,
                        n_objects=cfg["n_objects"], seed=1234)


def _render(args):
    cfg, k = args
    return _stream(cfg).frame(k)


def gen_frames(cfg, n, workers=0, cache=None):
    """n frames of the synthetic stream, ray-cast in parallel on the host cores (must run before CUDA is initialised: fork).
    workers = 1: no fork at all (under rocprofv3 --pmc the tool has initialised HSA before Python starts; forked workers hang at exit).
    cache: directory the rendered frames are kept in / taken from (profiler passes over the full 600-frame stream re-use the frames of the
    timing run instead of ray-casting them again without fork)."""
    st = _stream(cfg)
    tag = None
    if cache:
        tag = os.path.join(cache, f"frames_{cfg['W']}x{cfg['H']}_o{cfg['n_objects']}{'_s3' if cfg.get('s3') else ''}_n{n}")
        if os.path.exists(tag + "_rgb.npy"):
            rgb, depth, mask = (np.load(tag + s + ".npy", mmap_mode="r") for s in ("_rgb", "_depth", "_mask"))
            return st, [(np.ascontiguousarray(rgb[k]), np.ascontiguousarray(depth[k]), np.ascontiguousarray(mask[k])) for k in range(n)]
    workers = workers or max(1, min(48, (os.cpu_count() or 1) - 2, n))
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            frames = pool.map(_render, [(cfg, k) for k in range(n)], chunksize=max(1, n // (4 * workers)))
    else:
        frames = [st.frame(k) for k in range(n)]
    if tag:
        os.makedirs(cache, exist_ok=True)
        for s, j in (("_rgb", 0), ("_depth", 1), ("_mask", 2)):
            np.save(tag + s + ".npy", np.stack([f[j] for f in frames]))
    return st, frames


def pingpong(n_frames, steps):
    """0,1,..,n-1,n-2,..,1,0,1,.. so that consecutive frames always differ by one camera step."""
    idx, k, d = [], 0, 1
    for _ in range(steps):
        idx.append(k)
        if k + d < 0 or k + d >= n_frames:
            d = -d
        k += d
    return idx


def cpu_baseline(cfg, frames, max_seconds=25.0):
    """Oracle (port) frames/s on the host cores, bounded sample, from a separate -O3 -march=native build (oracle/mfo.py:
    build_fast; the parity build keeps -O2 -ffp-contract=off).  After 10 warm-up frames (the map has its working size, the pages are
    touched) the OpenMP thread count is swept on 10 frames per candidate -- the oracle's parallel regions are short per-kernel loops, more
    threads are not always faster -- and the reported value is the SUSTAINED rate of a further run at the chosen count."""
    import ctypes
    from oracle import mfo
    W, H, F = cfg["W"], cfg["H"], cfg["f"]
    fast = mfo.use_fast_build()
    ncpu = os.cpu_count() or 1
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    o = mfo.Oracle(W, H, F, F, W / 2.0, H / 2.0, icpWeight=100.0, capacity=(1 << 20) * (W * H // 307200), so3=0)
    o.process_frame(frames[0][0], frames[0][1])  # init frame, untimed
    order = pingpong(len(frames), 4000)[1:]
    pos = 0

    def run(n, limit):
        nonlocal pos
        t0 = time.time()
        done = 0
        while done < n and time.time() - t0 < limit:
            k = order[pos]; pos += 1
            o.process_frame(frames[k][0], frames[k][1])
            done += 1
        return done, time.time() - t0

    t_all = time.time()
    threads = ncpu
    sweep = {}
    if gomp is not None and ncpu > 4:
        gomp.omp_set_num_threads(min(ncpu, 32))
        run(10, 5.0)                                                   # warm-up
        best = None
        for cand in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)} | {ncpu}):
            gomp.omp_set_num_threads(cand)
            n, dt = run(10, 3.0)
            sweep[cand] = round(n / dt, 2)
            if best is None or n / dt > best[0]:
                best = (n / dt, cand)
        threads = best[1]
        gomp.omp_set_num_threads(threads)
    else:
        run(10, 5.0)
    n, dt = run(60, max(3.0, max_seconds - (time.time() - t_all)))
    o.close()
    return {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frames of the same {W}x{H} synthetic stream, sustained, after 1 init frame + 10 warm-up frames + the sweep; OpenMP oracle, "
                      f"{'-O3 -march=native build' if fast else 'parity build (-O2, no -march)'}, {threads} of {ncpu} hardware threads "
                      f"(sweep, 10 frames each, frames/s by thread count: {sweep})"}


def measured_copy_ceiling(dev, mib=1024, reps=10):
    """HBM bandwidth this box actually delivers to a plain streaming kernel: a device-to-device copy of `mib` MiB (torch's copy kernel),
    bytes read + bytes written over the HIP-event time of `reps` copies after two warm-up copies."""
    try:
        import torch
        n = mib * (1 << 20) // 4
        a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
        b = torch.empty_like(a)
        for _ in range(2):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        del a, b
        return {"GB/s": 2.0 * n * 4 / (ms * 1e-3) / 1e9, "what": f"device-to-device copy of {mib} MiB (read + write bytes), {reps} repetitions"}
    except Exception as e:   # a measurement aid: never the reason a bench line is missing
        print(f"[bench] copy ceiling not measured: {e}", file=sys.stderr)
        return None


def _newest_first(paths):
    """profiles/rNN<letter>_* are the intermediate measurements of round NN in order, profiles/rNN_* its final summary: newest first"""
    import re

    def key(path):
        m = re.match(r"r(\d+)([a-z0-9]*)_", os.path.basename(path))
        return (int(m.group(1)), m.group(2) or "~") if m else (-1, "")
    return sorted(paths, key=key, reverse=True)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/r*_pmc.json: separate rocprofv3 --pmc passes,
    gfx950 corrections applied when the file was written; see profiles/README.md).  None when there is no such record."""
    import glob
    for path in _newest_first(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json"))):
        try:
            doc = json.load(open(path))
            rec = doc["kernels"][kernel]
            return {"bytes_per_launch": rec["fetch_bytes"] + rec["write_bytes"], "fetch_bytes": rec["fetch_bytes"], "write_bytes": rec["write_bytes"],
                    "source": f"profiles/{os.path.basename(path)} (tree {doc.get('tree', '?')}; {rec.get('note', '')})"}
        except Exception:
            continue
    return None


def ranks_seen(world, local_rank):
    """who took part: the process group's size AND an all-gathered list of the devices the ranks actually sit on"""
    import torch
    import torch.distributed as dist
    props = torch.cuda.get_device_properties(local_rank)
    me = {"rank": int(os.environ.get("RANK", "0")), "local_rank": local_rank, "device": torch.cuda.current_device(), "name": props.name,
          "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", ""))}
    if world > 1 and dist.is_initialized():
        seen = [None] * world
        dist.all_gather_object(seen, me)
        return {"world_size": dist.get_world_size(), "devices": seen}
    return {"world_size": 1, "devices": [me]}


def sharded_scene(args, cfg, rank, local_rank, world, st, frames, steps, min_seconds):
    """The model-sharded scene (configs[3]; maskfusion_amd/sharded.py) on the already initialised process group: every rank calls
    process_frame for every frame (SPMD); rank 0 holds the inputs.  Returns the result record on rank 0 (None elsewhere)."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    from maskfusion_amd import MaskFusion, sharded
    from maskfusion_amd import dist as mfd
    W, H, F = cfg["W"], cfg["H"], cfg["f"]
    mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, device=local_rank, enableMultipleModels=True,
                    numGSurfels=cfg["surfels"] if rank == 0 else 1 << 20, numOSurfels=1 << 20, trackAllModels=True, modelSpawnOffset=2,
                    initConfidenceGlobal=10.0, initConfidenceObject=0.01)
    for k, v in (("mfThreshold", 0.3), ("mfWeightDistance", 150.0), ("mfWeightConvexity", 2.8), ("mfMorphEdgeIterations", 0),
                 ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", 0.004)):
        mf.setParam(k, v)
    if world == 1 or rank > 0:
        mf.preallocateModels(cfg["n_objects"] if world == 1 else max(2, (cfg["n_objects"] + world - 2) // (world - 1)))   # as single_context_scene: no hipMalloc at spawn time
    scfg = sharded.default_cfg(trackAllModels=True, modelSpawnOffset=2, depthCutoff=3.0)
    smf = sharded.ShardedMaskFusion(mf, dev, scfg)
    cls = [0] + [41 + i for i in range(cfg["n_objects"])]
    order = pingpong(len(frames) if frames else cfg["frames"], 1 << 16)
    cursor = [0]

    def run(n):
        for _ in range(n):
            k = order[cursor[0] % len(order)]
            cursor[0] += 1
            if rank == 0:
                smf.process_frame(frames[k][0], frames[k][1], frames[k][2], cls, 1.0, cursor[0])
            else:
                smf.process_frame(None, None, None, cls, 1.0, cursor[0])

    def barrier():
        mf.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    run(args.warmup)
    total_steps, total_dt, reps = 0, 0.0, 0
    while True:
        reps += 1
        barrier()
        t0 = time.perf_counter()
        run(steps)
        barrier()
        dt = mfd.max_over_ranks(time.perf_counter() - t0, dev)
        total_steps += steps
        total_dt += dt
        if total_dt >= min_seconds or reps >= args.max_reps:
            break
    local = [(m.getID(), m.lastCount()) for i, m in enumerate(mf.getModels()) if rank == 0 or i > 0]
    counts = torch.zeros(world, 2, dtype=torch.float32, device=dev)
    counts[rank, 0], counts[rank, 1] = len(local), sum(c for _, c in local)
    if world > 1:
        dist.all_reduce(counts)
    out = None
    if rank == 0:
        drift = float(np.linalg.norm(mf.getCurrPose()[:3, 3] - st.gt_pose(order[(cursor[0] - 1) % len(order)])[:3, 3]))
        per_rank = counts.cpu().numpy()
        out = {"metric": f"frames/sec ({W}x{H} RGB-D, one scene: background + {int(per_rank[:, 0].sum()) - 1} object models sharded by model over {world} GPU(s))",
               "value": total_steps / total_dt, "unit": "frames/s", "n_gpus": world, "steps": total_steps, "steps_requested": steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * total_dt / total_steps, "timed_seconds": total_dt, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": cfg["workload"], "frames_in_hbm": 0, "models": int(per_rank[:, 0].sum()), "surfels": int(per_rank[:, 1].sum()),
                          "models_per_rank": [int(x) for x in per_rank[:, 0]], "pose_drift_vs_gt_m": drift,
                          "parallelism": f"model-sharded x{world} (RCCL: frame broadcast, key all-reduce(MIN), state gather, label broadcast)",
                          "note": "host-pointer frames (the reference's FrameData boundary): H2D + broadcast per frame are inside the timed region"},
               "roofline": None, "roofline_frame": None, "host_input": None, "cpu_baseline": None}
    mf.close()
    return out


def single_context_scene(args, cfg, local_rank, st, frames, steps, min_seconds):
    """the same scene with EVERY model in one context on one GPU, through the same host-pointer boundary: the 1-GPU figure the sharded
    scene's speed-up is quoted against"""
    from maskfusion_amd import MaskFusion
    W, H, F = cfg["W"], cfg["H"], cfg["f"]
    mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, device=local_rank, enableMultipleModels=True,
                    numGSurfels=cfg["surfels"], numOSurfels=1 << 20, trackAllModels=True, modelSpawnOffset=2, initConfidenceGlobal=10.0,
                    initConfidenceObject=0.01)
    for k, v in (("mfThreshold", 0.3), ("mfWeightDistance", 150.0), ("mfWeightConvexity", 2.8), ("mfMorphEdgeIterations", 0),
                 ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", 0.004)):
        mf.setParam(k, v)
    mf.preallocateModels(cfg["n_objects"])
    cls = [0] + [41 + i for i in range(cfg["n_objects"])]
    order = pingpong(len(frames), 1 << 16)
    pos = 0
    for _ in range(args.warmup):
        k = order[pos]; pos += 1
        mf.processFrame(frames[k][0], frames[k][1], mask=frames[k][2], classIDs=cls)
    total_steps, total_dt = 0, 0.0
    while total_dt < min_seconds:
        mf.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            k = order[pos]; pos += 1
            mf.processFrame(frames[k][0], frames[k][1], mask=frames[k][2], classIDs=cls)
        mf.sync()
        total_dt += time.perf_counter() - t0
        total_steps += steps
    n_models = len(mf.getModels())
    mf.close()
    return {"value": total_steps / total_dt, "unit": "frames/s", "ms_per_step": 1e3 * total_dt / total_steps, "models": n_models,
            "note": "one context holding every model on rank 0's GPU, mf_process_frame with host pointers (the boundary the sharded form uses)"}


def run_sharded(args, cfg, rank, local_rank, world, st, frames):
    """--config 3: the model-sharded scene as the bench's own line."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    seen = ranks_seen(world, local_rank)
    out = sharded_scene(args, cfg, rank, local_rank, world, st, frames, args.steps, args.min_seconds)
    if rank == 0:
        out["ranks_seen"] = seen
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def rocprof_rows(names, pattern="r*kernel_stats.csv"):
    """per-kernel rows {name: {"us": average duration, "calls": n, "source": file}} from the newest committed rocprofv3 --kernel-trace --stats
    summary under profiles/ that holds them (the cross-check the roofline entries name; None when there is none)"""
    import csv
    import glob
    out = {}
    for path in _newest_first(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        if "_c4_" in os.path.basename(path) and "_c4_" not in pattern:      # (the configs[4] summaries are asked for by name)
            continue
        try:
            rows = list(csv.DictReader(open(path)))
        except Exception:
            continue
        for r in rows:
            nm = r.get("Name", "")
            for want in names:
                if (f"::{want}(" in nm or f"::{want}<" in nm or nm.startswith(want)) and want not in out:   # the function of that name, not one whose name contains it
                    try:
                        out[want] = {"us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"]), "kernel": nm, "source": "profiles/" + os.path.basename(path)}
                    except Exception:
                        pass
        if len(out) == len(names):
            break
    return out or None


def reference_default_variant(cfg, local_rank, d_rgb, d_depth, order, seconds=1.5):
    """The configuration the reference ships with (GUI/Tools/GUI.h:189,195: icpWeight 20 -> photometric term on, SO(3) pre-alignment on) on
    the same resident stream: SURVEY.md 8a rows a8-a10.  A Gauss-Newton iteration is two launches there (k_rgbd_iter: ICP normal equations +
    photometric correspondences; k_rgb_step: photometric normal equations), preceded by <= 10 SO(3) iterations on the 160x120 level.
    Reported beside `value` (a `variants` entry), never as `value`."""
    from maskfusion_amd import MaskFusion
    W, H, F = cfg["W"], cfg["H"], cfg["f"]
    P = W * H
    mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=20.0, so3=True, device=local_rank, enableMultipleModels=False, numGSurfels=cfg["surfels"])
    pos = 0
    for _ in range(60):
        k = order[pos]; pos += 1
        mf.processFrameDevice(d_rgb[k].data_ptr(), d_depth[k].data_ptr())
    mf.sync()
    steps, dt = 0, 0.0
    while dt < seconds:
        t0 = time.perf_counter()
        for _ in range(200):
            k = order[pos % len(order)]; pos += 1
            mf.processFrameDevice(d_rgb[k].data_ptr(), d_depth[k].data_ptr())
        mf.sync()
        dt += time.perf_counter() - t0
        steps += 200
    mf.enableTimings(True)
    acc, n = {}, 60
    so3_its = 0.0
    for _ in range(n):
        k = order[pos % len(order)]; pos += 1
        mf.processFrameDevice(d_rgb[k].data_ptr(), d_depth[k].data_ptr())
        for kx, v in mf.timings().items():
            acc[kx] = acc.get(kx, 0.0) + v
        so3_its += mf.trackStats(0)["so3Iterations"]
    mf.enableTimings(False)
    stages = {kx: v / n for kx, v in acc.items()}
    count = mf.getBackgroundModel().lastCount()
    mf.close()
    # SURVEY.md 8d contract bytes per pixel and iteration: ICP 48 B, computeRgbResidual 30 B, rgbStep 32 B; (10 + 5/4 + 4/16) P pixel-iterations
    px_it = (10 + 5 / 4.0 + 4 / 16.0) * P
    loop_bytes = px_it * (48 + 30 + 32)
    t_loop = stages["icpIterations"] * 1e-3
    kern = rocprof_rows(["k_rgbd_iter", "k_rgb_step", "k_so3_iter"])
    levels = None
    if kern:
        per_launch = {"k_rgbd_iter": px_it * 78 / 19, "k_rgb_step": px_it * 32 / 19, "k_so3_iter": 2.0 * (P / 16)}
        levels = {nm: dict(r, bytes=per_launch[nm], frac=per_launch[nm] / (r["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS) for nm, r in kern.items()}
    return {"workload": "configs[1]'s stream with the reference's GUI defaults: icpWeight=20 (photometric term on), SO(3) pre-alignment on",
            "value": steps / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "surfels": count,
            "so3_iterations_per_frame": so3_its / n,
            "roofline": {"bound": "hbm", "kernel": "k_rgbd_iter + k_rgb_step (19 iterations x 2 launches per frame)",
                         "algorithmic_bytes_per_iteration": loop_bytes / 19, "us_per_iteration": t_loop * 1e6 / 19,
                         "achieved": loop_bytes / t_loop / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": loop_bytes / t_loop / 1e9 / HBM_PEAK_GBS,
                         "note": "HIP events on the library's stream around the 38 launches of the loop; contract bytes 48 + 30 + 32 B per pixel "
                                 "and iteration (SURVEY.md 8d); `levels`: per kernel from the committed rocprofv3 summary named in each entry",
                         "levels": levels, "stage_ms": stages}}

# ------------------------------------------------------------------------------------------------------------------------------------
# configs[4] as specified: 1280x960, 32M / 4M surfel budgets pre-filled to >= 80 %, 4 objects -- the workload on which
# the surfel passes (Core/Model/Model.cpp:466-772: predictIndices, fuse, clean, combinedPredict) are bandwidth-relevant.
# ------------------------------------------------------------------------------------------------------------------------------------
C4_LEAD_IN = 10          # frames the lead-in may take (maskfusion_amd/stress.py: the fourth object spawns in frame 8)


class RoomMapJob:
    """The 26.5 M-surfel background map takes ~6 s of numpy: it is generated by a forked child (before CUDA exists in this process) into
    /dev/shm while the parent goes on, and mapped when the scenario needs it."""

    def __init__(self, n_objects=4, fork=True):
        import tempfile
        self.n_objects = n_objects
        self.pid = None
        if not fork:       # profiler runs (--gen-workers 1): no forked children under rocprofv3
            return
        self.path = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir(), f"mf_bench_room_{os.getpid()}.npy")
        self.pid = os.fork()
        if self.pid == 0:
            try:
                np.save(self.path, self.generate())
            finally:
                os._exit(0)

    def generate(self):
        from maskfusion_amd import stress, synth
        return synth.dense_room_map(stress.stream(self.n_objects).scene, int(1.005 * 0.8 * stress.surfel_capacity(stress.NUM_GSURFELS)), last_time=0.0,
                                    furniture_above=self.n_objects)

    def result(self):
        if self.pid is None:
            return self.generate()
        os.waitpid(self.pid, 0)
        m = np.load(self.path)
        os.unlink(self.path)
        return m


# SURVEY.md 8d contract bytes of the surfel half per PASS (mf_get_pass_timings: the passes as the driver run itself times them, HIP events around each
# pass's launches): one read-modify-write (update + clean) 96 B/surfel and two projections of 48 B/surfel each per model and frame (+ GlobalProjection's
# 48 B/surfel); image-side outputs 52 P per index map, 38 P for the prediction.  N: surfels of the model(s) the pass serves, V: surfels of the runs
# the pass actually VISITS (the buffer is kept as runs of <= 512 surfels, Surfels::box: the projection passes visit the runs k_cull lists, Model::clean
# in place those k_cull_clean lists) -- `visited_bytes` is what a roofline fraction is taken over, `contract_bytes` what the reference's dataflow moves.
C4_PASS_BYTES = {
    "bgGlobalProjection": lambda N, V, P, n: (48.0 * N, 48.0 * V),
    "bgIndexMap": lambda N, V, P, n: (48.0 * N + 52.0 * P, 48.0 * V + 52.0 * P),
    "bgFuseData": lambda N, V, P, n: (76.0 * P, 76.0 * P),                         # the frame (rgb 3 + raw / filtered depth 8 + mask 1), the index map's 52 P, candidate records 48 B x P / 4
    "bgFuseUpdate": lambda N, V, P, n: (48.0 * N, 0.0),                            # update.vert: the reference copies the buffer; here only merged surfels move
    "bgIndexMap2": lambda N, V, P, n: (48.0 * N + 52.0 * P, 48.0 * V + 52.0 * P),
    "bgClean": lambda N, V, P, n: (96.0 * N, 48.0 * V),                            # THE read-modify-write of the frame; in place: the listed runs are read, what changes is written
    "bgAppend": lambda N, V, P, n: (0.0, P / 4 * 48.0),
    "bgPredict": lambda N, V, P, n: (48.0 * N + 38.0 * P, 48.0 * V + 38.0 * P),
    "objGlobalProjection": lambda N, V, P, n: (48.0 * N, 48.0 * V),
    "objFuseClean": lambda N, V, P, n: (192.0 * N + 104.0 * P * n, 192.0 * V + 104.0 * P * n),
    "objPredict": lambda N, V, P, n: (48.0 * N + 38.0 * P * n, 48.0 * V + 38.0 * P * n),
}
C4_BG_LISTS = {"bgGlobalProjection": "visible", "bgIndexMap": "visible", "bgIndexMap2": "visible", "bgPredict": "visible", "bgClean": "clean"}


def config4_scene(local_rank, frames, room_job, seconds=2.0, frames_per_rep=30, stages_frames=10, params=(), max_reps=64, tracked=True):
    """The dense configs[4] scenario on one GPU.  One repetition = a fresh context taken through the lead-in and the map uploads of
    maskfusion_amd/stress.py (untimed), 4 untimed frames, then `frames_per_rep` timed frames over device-resident inputs (ping-ponged)
    between two synchronisations.  tracked (S3 as SURVEY.md 8d defines it: "S2 with 4 objects", and S2 tracks every model): after the lead-in every
    object model is made non-static -- each frame then runs FIVE Gauss-Newton loops (the batched kernels) before its surfel passes.  The timed
    window is short on purpose: the scene is not stationary -- over hundreds of frames Model::clean's mask-disagreement decay
    (copy_unstable.vert:139-156) thins the object maps out and the label stage spawns further models -- and the figure is meant for maps that ARE
    >= 80 % full.  Repetitions are added until `seconds` of timed frames have run.  The last repetition is followed by an instrumented pass of
    `stages_frames` frames IN THIS RUN: stage timings (mf_get_timings) and the surfel passes one by one (mf_get_pass_timings), each with its SURVEY.md 8d
    contract bytes and the bytes of the runs it visited."""
    import torch
    from maskfusion_amd import stress
    dev = torch.device("cuda", local_rank)
    st = stress.stream(4)
    cls = [0] + [41 + i for i in range(4)]
    room = room_job.result() if room_job is not None else None
    d = None
    steps, dt, reps, t_setup = 0, 0.0, 0, 0.0
    fills_start = None
    while True:
        reps += 1
        mf = stress.make_context(local_rank)
        for key, val in params:
            mf.setParam(key, val)
        t0 = time.perf_counter()
        k0, loaded = stress.lead_in(mf, st, frames, cls, n_objects=4, max_frames=C4_LEAD_IN, room_map=room,
                                    log=(lambda m: print("[bench c4] " + m, file=sys.stderr)) if reps == 1 else None)
        if tracked:
            stress.track_objects(mf)
        mf.sync()
        t_setup += time.perf_counter() - t0
        if d is None:
            d = [tuple(torch.from_numpy(x).to(dev) for x in f) for f in frames[k0:]]
            order = pingpong(len(d), 1 << 16)
        mf.setMaskClassIDs(cls)
        pos = 0

        def step():
            nonlocal pos
            r, dd, m = d[order[pos % len(order)]]
            pos += 1
            mf.processFrameDevice(r.data_ptr(), dd.data_ptr(), m.data_ptr(), timestamp=k0 + pos)

        for _ in range(4):
            step()
        mf.sync()
        if fills_start is None:
            fills_start = [m.lastCount() for m in mf.getModels()]
        t0 = time.perf_counter()
        for _ in range(frames_per_rep):
            step()
        mf.sync()
        dt += time.perf_counter() - t0
        steps += frames_per_rep
        if dt >= seconds or reps >= max_reps:
            break
        mf.close()
    models_timed = len(mf.getModels())
    compactions = mf.getParam("densifyCount")
    mf.enableTimings(True)
    mf.setParam("passTimings", 1)
    acc, pacc, lists = {}, {}, {"visible": 0.0, "clean": 0.0, "table": 0.0}
    for _ in range(stages_frames):
        step()
        for kx, v in mf.timings().items():
            acc[kx] = acc.get(kx, 0.0) + v
        for kx, v in mf.passTimings().items():
            pacc[kx] = pacc.get(kx, 0.0) + v
        lists["visible"] += mf.getParam("visibleRuns"); lists["clean"] += mf.getParam("cleanRuns"); lists["table"] += mf.getParam("backgroundRuns")
    mf.enableTimings(False)
    mf.setParam("passTimings", 0)
    stages = {kx: v / stages_frames for kx, v in acc.items()}
    passes = {kx: v / stages_frames for kx, v in pacc.items()}
    lists = {kx: v / stages_frames for kx, v in lists.items()}
    models = mf.getModels()
    counts = [m.lastCount() for m in models]
    ids = [m.getID() for m in models]
    n_nonstatic = sum(1 for m in models[1:] if m.isNonstatic())
    drift = float(np.linalg.norm(mf.getCurrPose()[:3, 3] - st.gt_pose(k0 + order[(pos - 1) % len(order)])[:3, 3]))
    caps = [stress.surfel_capacity(stress.NUM_GSURFELS)] + [stress.surfel_capacity(stress.NUM_OSURFELS)] * (len(counts) - 1)
    mf.close()
    del room
    P = stress.W * stress.H
    ms = 1e3 * dt / steps
    N_bg, N_all = fills_start[0], sum(fills_start)
    n_tracked = 1 + (len(fills_start) - 1 if tracked else 0)
    frame_bytes = 741.0 * P * n_tracked + 192.0 * N_all            # every tracked model's odometry + the surfel half of every model
    rows = {}
    N_obj, n_obj = sum(counts[1:]), len(counts) - 1
    for name, ms_p in passes.items():
        if name not in C4_PASS_BYTES or ms_p <= 0.0:
            continue
        bg = name.startswith("bg")
        N = counts[0] if bg else N_obj
        V = min(N, 512.0 * lists[C4_BG_LISTS[name]]) if name in C4_BG_LISTS else N      # object models: every run is visited
        contract, visited = C4_PASS_BYTES[name](N, V, P, n_obj)
        rows[name] = {"ms": ms_p, "contract_bytes": contract, "visited_bytes": visited, "achieved_GBps": visited / (ms_p * 1e-3) / 1e9,
                      "frac": visited / (ms_p * 1e-3) / 1e9 / HBM_PEAK_GBS, "contract_GBps": contract / (ms_p * 1e-3) / 1e9}
    return {"workload": CONFIGS["4"]["workload"] + ("; the four object models TRACKED (non-static: five Gauss-Newton loops per frame)" if tracked else
                                                    "; the objects stand and follow the camera (static)"),
            "value": steps / dt, "unit": "frames/s", "ms_per_step": ms, "steps": steps, "repetitions": reps,
            "frames_per_repetition": frames_per_rep, "models": len(counts), "models_through_the_timed_window": models_timed, "model_ids": ids,
            "tracked_models": 1 + n_nonstatic, "surfels_at_start": fills_start, "surfels_at_end": counts,
            "fill_at_start": [c / cap for c, cap in zip(fills_start, caps)], "fill_at_end": [c / cap for c, cap in zip(counts, caps)],
            "pose_drift_vs_gt_m": drift, "setup_seconds": t_setup, "lead_in_frames": k0, "stage_ms": stages, "compactions": compactions,
            "roofline_frame": {"bound": "hbm", "algorithmic_bytes": frame_bytes, "ms": ms, "achieved": frame_bytes / (ms * 1e-3) / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frame_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "note": "(741 P per tracked model (%d) + 192 N over every model's surfels at the start of the timed window) bytes per "
                                       "frame, SURVEY.md 8d" % n_tracked},
            "runs": {"background_table": lists["table"], "visible": lists["visible"], "clean_in_place": lists["clean"],
                     "note": "runs of <= 512 surfels (Surfels::box): of the background's table / on the visibility list of its projection passes / "
                             "visited by its in-place clean pass; averages over the instrumented frames"},
            "roofline_passes": {"source": "mf_get_pass_timings in THIS run: HIP events around each pass's launches, mean of %d frames" % stages_frames,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "passes": rows,
                                "note": "frac = visited_bytes / time / peak: the bytes of the surfels in the runs the pass visits (48 B each; the passes "
                                        "of the object models visit every run) + its image-side bytes; contract_bytes: SURVEY.md 8d's figure for the "
                                        "reference's dataflow, which streams every buffer whole"}}


def small_variant(cfg_key, local_rank, frames, n_timed, warm):
    """frames/s of another configuration of the table through a fresh context, device-resident frames, ping-ponged: the figures the default
    line carries beside `value` so that the driver times them too (`variants.multi_model_2s`: S2 on one GPU, background + object models,
    batched Gauss-Newton loop; `variants.vga_1280x960`: the 1280x960 stream with its natural map)."""
    import torch
    from maskfusion_amd import MaskFusion
    cfg = CONFIGS[cfg_key]
    W, H, F = cfg["W"], cfg["H"], cfg["f"]
    dev = torch.device("cuda", local_rank)
    multi = cfg["n_objects"] > 0
    if multi:
        mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, device=local_rank, enableMultipleModels=True, numGSurfels=cfg["surfels"],
                        numOSurfels=1 << 20, trackAllModels=True, modelSpawnOffset=2, initConfidenceGlobal=10.0, initConfidenceObject=0.01)
        for k, v in (("mfThreshold", 0.3), ("mfWeightDistance", 150.0), ("mfWeightConvexity", 2.8), ("mfMorphEdgeIterations", 0),
                     ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", 0.004)):
            mf.setParam(k, v)
        mf.preallocateModels(cfg["n_objects"])
        mf.setMaskClassIDs([0] + [41 + i for i in range(cfg["n_objects"])])
    else:
        mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, device=local_rank, enableMultipleModels=False, numGSurfels=cfg["surfels"])
    d = [tuple(torch.from_numpy(x).to(dev) for x in f) for f in frames]
    order = pingpong(len(d), 1 << 16)
    pos = 0

    def step():
        nonlocal pos
        r, dd, m = d[order[pos % len(order)]]
        pos += 1
        mf.processFrameDevice(r.data_ptr(), dd.data_ptr(), m.data_ptr() if multi else 0)

    for _ in range(warm):
        step()
    mf.sync()
    t0 = time.perf_counter()
    for _ in range(n_timed):
        step()
    mf.sync()
    dt = time.perf_counter() - t0
    counts = [m.lastCount() for m in mf.getModels()]
    mf.close()
    return {"workload": cfg["workload"], "value": n_timed / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / n_timed, "steps": n_timed, "warmup": warm,
            "models": len(counts), "surfels": sum(counts)}


def run_config4(args, local_rank, frames, room_job):
    """--config 4 as the bench's own line"""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    res = config4_scene(local_rank, frames, room_job, seconds=args.min_seconds, frames_per_rep=max(10, min(args.steps, 60)), tracked=not args.static_objects,
                        params=[tuple((kv.partition("=")[0], float(kv.partition("=")[2]))) for kv in args.param])
    out = {"metric": f"frames/sec (1280x960 RGB-D, background + {res['models'] - 1} object models, 32M / 4M surfel budgets filled >= 80 %, ICP + surfel fusion)",
           "value": res["value"], "unit": "frames/s", "n_gpus": 1, "steps": res["steps"], "steps_requested": args.steps, "warmup": 4,
           "ms_per_step": res["ms_per_step"], "timed_seconds": res["steps"] * res["ms_per_step"] * 1e-3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": res["workload"], "frames_in_hbm": len(frames) - res["lead_in_frames"], "models": res["models"],
                      "surfels": res["surfels_at_start"], "fill": res["fill_at_start"], "surfels_at_end": res["surfels_at_end"],
                      "repetitions": res["repetitions"], "frames_per_repetition": res["frames_per_repetition"],
                      "pose_drift_vs_gt_m": res["pose_drift_vs_gt_m"], "parallelism": "one context, one GPU",
                      **({"params": {kv.partition("=")[0]: float(kv.partition("=")[2]) for kv in args.param}} if args.param else {})},
           "roofline": None, "roofline_frame": res["roofline_frame"], "roofline_passes": res["roofline_passes"], "runs": res["runs"], "stage_ms": res["stage_ms"],
           "compactions": res["compactions"], "tracked_models": res["tracked_models"],
           "setup_seconds": res["setup_seconds"], "host_input": None, "cpu_baseline": None, "ranks_seen": ranks_seen(1, local_rank)}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--frames", type=int, default=0, help="distinct synthetic frames kept in HBM (0: the config's default; ping-ponged)")
    ap.add_argument("--config", default="1", choices=tuple(CONFIGS), help="workload (1 = the metric's)")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="the timed region is repeated until it has lasted this long")
    ap.add_argument("--max-reps", type=int, default=100000, help="runaway guard on the repetitions of the timed region")
    ap.add_argument("--icp-weight", type=float, default=100.0, help="icpWeight (>= 100: geometric term only, the metric's setting; "
                    "the reference GUI default is 20: photometric term on, two launches per Gauss-Newton iteration)")
    ap.add_argument("--so3", action="store_true", help="SO(3) photometric pre-alignment (reference default: on)")
    ap.add_argument("--no-batch", action="store_true", help="config 2s: track the models one after the other (A/B of the batched loop)")
    ap.add_argument("--frame-cache", default="", help="directory to keep / find the rendered frames in (profiler passes)")
    ap.add_argument("--gen-workers", type=int, default=0, help="processes that ray-cast the synthetic frames (0: auto; 1: no fork, for profiler runs)")
    ap.add_argument("--param", action="append", default=[], metavar="KEY=VALUE",
                    help="mf_set_param(KEY, VALUE) on the context before the run (A/B of implementation switches, e.g. persistentIcp=1); "
                         "recorded in config.params")
    ap.add_argument("--static-objects", action="store_true", help="config 4: the object models stand and follow the camera (rounds 4-5's scenario) instead of "
                                                                  "being tracked (S3 as SURVEY.md 8d defines it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-host-input", action="store_true")
    ap.add_argument("--host-input-frames", type=int, default=150, help="frames of the host-pointer side measurement (host_input)")
    ap.add_argument("--no-variants", action="store_true", help="skip the side measurements of config 1: reference defaults (icpWeight 20 + SO(3)) and the dense configs[4] scenario")
    ap.add_argument("--force-sharded-scene", action="store_true", help="run that part at N = 1 as well (rehearsal of the N > 1 code path on one GPU)")
    ap.add_argument("--no-sharded-scene", action="store_true", help="N > 1: skip the model-sharded 8-object scene that is timed beside the weak-scaling line")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    W, H, F = cfg["W"], cfg["H"], cfg["f"]
    CX, CY = W / 2.0, H / 2.0
    multi = cfg["n_objects"] > 0
    n_frames = args.frames or cfg["frames"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the stream is ray-cast before CUDA exists in this process (the generator forks); only rank 0 owns frames
    st, frames = gen_frames(cfg, n_frames, args.gen_workers, args.frame_cache or None) if rank == 0 else (None, None)
    # N > 1 on the metric's config: the north star's multi-GPU scene (configs[3], 8 objects sharded by model) is timed in the same run,
    # beside the weak-scaling line whose per-N values the driver compares
    with_scene = (world > 1 or args.force_sharded_scene) and args.config == "1" and not args.no_sharded_scene
    st3, frames3 = gen_frames(CONFIGS["3"], CONFIGS["3"]["frames"], args.gen_workers) if (with_scene and rank == 0) else (None, None)
    # the dense configs[4] scenario rides along as a variant of the default line (and is --config 4's own line): its frames are ray-cast and
    # its background map generated (a forked child) before CUDA exists in this process
    with_c4 = rank == 0 and world == 1 and ((args.config == "1" and not args.no_variants and args.icp_weight >= 100.0 and not args.so3) or args.config == "4")
    frames4 = room_job = frames2s = frames4n = None
    if with_c4:
        room_job = RoomMapJob(4, fork=args.gen_workers != 1)
        frames4 = frames if args.config == "4" else gen_frames(CONFIGS["4"], C4_LEAD_IN + 12, args.gen_workers, args.frame_cache or None)[1]
        if args.config == "1":   # the two small variants of the default line
            frames2s = gen_frames(CONFIGS["2s"], 60, args.gen_workers, args.frame_cache or None)[1]
            frames4n = gen_frames(CONFIGS["4n"], 40, args.gen_workers, args.frame_cache or None)[1]
    if args.config == "4":
        if world > 1:
            raise SystemExit("--config 4 is a one-GPU scenario (configs[4]'s per-GPU share); run it with --gpus 1")
        return run_config4(args, local_rank, frames4, room_job)
    if args.config == "3":
        if args.steps == 600:
            args.steps = 120
        return run_sharded(args, cfg, rank, local_rank, world, st, frames)

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path); cpu_baseline is only a side measurement")
    dev = torch.device("cuda", local_rank)

    from maskfusion_amd import MaskFusion
    d_rgb = d_depth = d_mask = None
    if rank == 0:
        d_rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
        d_depth = [torch.from_numpy(f[1]).to(dev) for f in frames]
        if multi:
            d_mask = [torch.from_numpy(f[2]).to(dev) for f in frames]
    if multi:
        # trackAllModels, confO = 0.01, confG = 10; the GUI's segmentation parameters (GUI/Tools/GUI.h:367-374) with
        # a new-model size that fits 0.2-0.35 m boxes at VGA; a short spawn offset so that all objects exist after the warm-up
        mf = MaskFusion(W, H, F, F, CX, CY, icpThresh=args.icp_weight, so3=args.so3, device=local_rank, enableMultipleModels=True,
                        numGSurfels=cfg["surfels"], numOSurfels=1 << 20, trackAllModels=True, modelSpawnOffset=2, initConfidenceGlobal=10.0,
                        initConfidenceObject=0.01)
        for k, v in (("mfThreshold", 0.3), ("mfWeightDistance", 150.0), ("mfWeightConvexity", 2.8), ("mfMorphEdgeIterations", 0),
                     ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", 0.004), ("batchTracking", 0 if args.no_batch else 1)):
            mf.setParam(k, v)
        mf.preallocateModels(cfg["n_objects"])
        mf.setMaskClassIDs([0] + [41 + i for i in range(cfg["n_objects"])])
    else:
        mf = MaskFusion(W, H, F, F, CX, CY, icpThresh=args.icp_weight, so3=args.so3, device=local_rank, enableMultipleModels=False,
                        numGSurfels=cfg["surfels"])
    extra_params = {}
    for kv in args.param:                                   # A/B of implementation switches: the same workload, one switch flipped
        key, _, val = kv.partition("=")
        mf.setParam(key, float(val))
        extra_params[key] = float(val)
    # Every rank owns one context.  Rank 0 owns the input stream and publishes frame k to all ranks (RCCL broadcast over xGMI
    # when N > 1) on the library's INPUT stream into a ring of 3 buffers; each rank then enqueues processFrame, whose main
    # stream is torch's current stream for the gather of the per-model state record.  Collectives and kernels are ordered
    # by the streams alone (no host synchronisation inside the timed region).
    from maskfusion_amd import dist as mfd
    order = pingpong(n_frames, 1 << 16)
    ext = torch.cuda.ExternalStream(mf.stream(), device=dev)
    ext_in = torch.cuda.ExternalStream(mf.inputStream(), device=dev)
    loop_state = {}
    cursor = [0]

    def get_frame(_i):
        k = order[cursor[0] % len(order)]
        return d_rgb[k], d_depth[k]

    def model_step(rgb, depth, stats):
        if multi:
            mf.processFrameDevice(rgb.data_ptr(), depth.data_ptr(), d_mask[order[cursor[0] % len(order)]].data_ptr())
        else:
            mf.processFrameDevice(rgb.data_ptr(), depth.data_ptr())
        if world > 1:
            mf.modelStateDevice(0, stats.data_ptr())   # 64 B record for the gather; single GPU reads the pinned mirror
        cursor[0] += 1

    def run(n):
        return mfd.run_steps(get_frame, model_step, n, H, W, dev, stream_ctx=lambda: torch.cuda.stream(ext),
                             input_stream_ctx=lambda: torch.cuda.stream(ext_in), state=loop_state)

    def barrier():
        mf.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    run(args.warmup)
    barrier()
    # timed region: `steps` steps between barriers, repeated (each repetition bracketed the same way) until min_seconds have passed;
    # every rank runs the same number of repetitions (the decision is taken on the max-over-ranks time)
    total_steps, total_dt, reps = 0, 0.0, 0
    while True:
        reps += 1
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        dt = mfd.max_over_ranks(time.perf_counter() - t0, dev)
        total_steps += args.steps
        total_dt += dt
        if total_dt >= args.min_seconds or reps >= args.max_reps:   # (max_reps: a runaway guard only -- 100 000 by default)
            break
    fps = world * total_steps / total_dt

    n_models = len(mf.getModels())
    count = sum(m.lastCount() for m in mf.getModels())
    drift = None
    if rank == 0:
        # sanity: the tracked pose must still follow the synthetic ground truth (a fast wrong answer is worthless)
        pose = mf.getCurrPose()
        gt = st.gt_pose(order[(cursor[0] - 1) % len(order)])
        drift = float(np.linalg.norm(pose[:3, 3] - gt[:3, 3]))

    roofline = roofline_frame = None
    P = W * H
    if rank == 0 and not args.no_roofline and args.icp_weight >= 100.0:
        # instrumented pass over the same stream: per-stage HIP events on the library's own stream
        mf.enableTimings(True)
        acc = {}
        n = 100
        t_frames = 0.0
        for i in range(n):
            k = order[cursor[0] % len(order)]
            cursor[0] += 1
            if multi:
                mf.processFrameDevice(d_rgb[k].data_ptr(), d_depth[k].data_ptr(), d_mask[k].data_ptr())
            else:
                mf.processFrameDevice(d_rgb[k].data_ptr(), d_depth[k].data_ptr())
            for kx, v in mf.timings().items():
                acc[kx] = acc.get(kx, 0.0) + v
        mf.enableTimings(False)
        stages = {kx: v / n for kx, v in acc.items()}
        n_tracked = len(mf.getModels())            # config 1 / 4: the background; 2s: every model (trackAllModels)
        icp_bytes = 552 * P * n_tracked            # (10 + 5/4 + 4/16) * 48 B * P per model-frame
        n_launch = 19
        t_icp = stages["icpIterations"] * 1e-3 / n_launch   # HIP events around the 19 iterations on the library's stream
        achieved = icp_bytes / n_launch / t_icp / 1e9
        batched = multi and not args.no_batch
        kname = "k_icp_batch_solve + k_icp_batch_pixels" if batched else "k_icp_iter"
        levels = None
        if not batched and stages.get("icpFine", 0.0) > 0.0:
            # the same interval split at the level-0 boundary (a third event on the library's stream): one entry per rocprofv3 kernel name,
            # so that each can be checked against ONE row of profiles/*kernel_stats.csv (level 0 runs the 512-thread instantiation at
            # VGA and above, the coarse levels the 256-thread one whenever a level has fewer than 240 chunks of 512 pixels)
            n_fine, n_coarse = 10, 9
            b_fine = 48.0 * P
            b_coarse = (5 * 48.0 * P / 4 + 4 * 48.0 * P / 16) / n_coarse
            us_f, us_c = stages["icpFine"] * 1e3 / n_fine, stages["icpCoarse"] * 1e3 / n_coarse
            # kernel names as rocprofv3 prints them (template arguments: threads per workgroup, pixel slots per thread): VGA runs level 0 as
            # <512, 3> and levels 1 and 2 as <256, 1>; at 1280x960 level 1 is a <512, 3> launch as well
            levels = {"L0": {"kernel": "void mf::k_icp_iter<512, 3>(mf::IcpKArgs)", "launches_per_frame": n_fine, "us": us_f, "bytes": b_fine,
                             "frac": b_fine / (us_f * 1e-6) / 1e9 / HBM_PEAK_GBS},
                      "coarse": {"kernel": "void mf::k_icp_iter<256, 1>(mf::IcpKArgs)" if P <= 240 * 512 * 4 else "void mf::k_icp_iter<512, 3>(mf::IcpKArgs) (level 1) / <256, 1> (level 2)",
                                 "launches_per_frame": n_coarse, "us": us_c,
                                 "bytes": b_coarse, "frac": b_coarse / (us_c * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                 "note": "5 launches at level 1 (12 P bytes each) + 4 at level 2 (3 P): the average launch"}}
        ceiling = measured_copy_ceiling(dev)
        roofline = {"bound": "hbm", "kernel": f"{kname} (19 iterations/frame, L2:4 L1:5 L0:10; {n_tracked} model(s) per launch)" if batched
                    else f"{kname} (19 launches per model and frame, L2:4 L1:5 L0:10)",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": icp_bytes / n_launch, "us_per_launch": t_icp * 1e6,
                    # the committed PMC profile was taken on configs[1] (VGA, one model): it says nothing about the other workloads
                    "traffic": pmc_traffic("k_icp_iter") if args.config == "1" else None, "levels": levels, "stage_ms": stages,
                    # the same figure over a ceiling MEASURED on this box (a device-to-device copy: read + write bytes)
                    "measured_ceiling": dict(ceiling, frac_of_measured=achieved / ceiling["GB/s"]) if ceiling else None}
        frame_bytes = 741 * P * n_tracked + 192 * count
        roofline_frame = {"bound": "hbm", "algorithmic_bytes": frame_bytes, "ms": 1e3 * total_dt / total_steps,
                          "achieved": frame_bytes / (total_dt / total_steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": frame_bytes / (total_dt / total_steps) / 1e9 / HBM_PEAK_GBS,
                          "note": "(741 P per tracked model + 192 N) bytes per frame, SURVEY.md 8d; N = live surfels of all models"}

    host_input = None
    if rank == 0 and world == 1 and not args.no_host_input:
        # the reference's own boundary: FrameData in host memory (MaskFusion.cpp:212-216 uploads it every frame)
        n = args.host_input_frames
        ks = [order[(cursor[0] + i) % len(order)] for i in range(n)]
        cursor[0] += n
        cls = [0] + [41 + i for i in range(cfg["n_objects"])]
        for k in ks[:12]:     # untimed: the entry point's first calls (pinned buffers, helper thread, and the frame graphs if that switch is on)
            if multi:
                mf.processFrame(frames[k][0], frames[k][1], mask=frames[k][2], classIDs=cls)
            else:
                mf.processFrame(frames[k][0], frames[k][1])
        mf.setParam("hostProfileReset", 1)
        mf.sync()
        t0 = time.perf_counter()
        for k in ks:
            if multi:
                mf.processFrame(frames[k][0], frames[k][1], mask=frames[k][2], classIDs=cls)
            else:
                mf.processFrame(frames[k][0], frames[k][1])
        dt_calls = time.perf_counter() - t0     # the calls alone: how long the HOST needs per frame (staging copy + uploads + ~34 launches)
        mf.sync()
        dt_h = time.perf_counter() - t0
        host_input = {"value": n / dt_h, "unit": "frames/s", "ms_per_step": 1e3 * dt_h / n, "host_ms_per_call": 1e3 * dt_calls / n,
                      "host_us_inside_the_call": {k: mf.getParam(k) for k in ("hostWaitUs", "hostStageUs", "hostUploadUs", "hostEnqueueUs", "hostCallUs")},
                      "note": f"mf_process_frame with host pointers (pageable numpy arrays): {(7 + (1 if multi else 0)) * P / 1e6:.2f} MB per frame copied into a "
                              "pinned double buffer and uploaded asynchronously (one packed copy) under the previous frame's kernels; no synchronisation per frame, "
                              "the host kept at most two frames ahead (hostLockstep: its wait is inside hostUploadUs) "
                              "(rounds 1-3: one hipStreamSynchronize per frame)"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not multi:   # rank 0 at N = 1 only (the other ranks would sit in a collective)
        cpu = cpu_baseline(cfg, frames)

    variants = None
    if rank == 0 and world == 1 and args.config == "1" and not args.no_variants and args.icp_weight >= 100.0 and not args.so3:
        try:
            variants = {"reference_default": reference_default_variant(cfg, local_rank, d_rgb, d_depth, order)}
        except Exception as e:   # a side measurement: never the reason the bench line is missing
            print(f"[bench] reference-default variant not measured: {e!r}", file=sys.stderr)
        if with_c4:
            for name, key, fr, n_timed, warm in (("multi_model_2s", "2s", frames2s, 120, 60), ("vga_1280x960", "4n", frames4n, 120, 40)):
                try:
                    variants = dict(variants or {}, **{name: small_variant(key, local_rank, fr, n_timed, warm)})
                except Exception as e:
                    print(f"[bench] variant {name} not measured: {e!r}", file=sys.stderr)
            try:
                mf.close()        # the dense scenario holds ~6 GB of maps: it gets the GPU to itself
                d_rgb = d_depth = None
                room4 = room_job.result() if room_job is not None else None

                class _Room:          # (the generated map is handed to both runs)
                    def result(self):
                        return room4
                c4 = config4_scene(local_rank, frames4, _Room(), seconds=0.0, frames_per_rep=30, tracked=True)
                try:                  # rounds 4-5's scenario beside it: the objects stand and follow the camera (one Gauss-Newton loop per frame)
                    st4 = config4_scene(local_rank, frames4, _Room(), seconds=0.0, frames_per_rep=30, stages_frames=4, tracked=False)
                    c4["static_objects"] = {k: st4[k] for k in ("workload", "value", "unit", "ms_per_step", "steps", "tracked_models", "surfels_at_start", "roofline_frame")}
                except Exception as e:
                    print(f"[bench] configs[4] with static objects not measured: {e!r}", file=sys.stderr)
                variants = dict(variants or {}, config4_stress=c4)
            except Exception as e:
                print(f"[bench] configs[4] stress variant not measured: {e!r}", file=sys.stderr)

    seen = ranks_seen(world, local_rank)
    scene = None
    if with_scene:
        mf.close()   # the weak-scaling contexts are done: the scene gets the GPUs to itself
        scene = sharded_scene(args, CONFIGS["3"], rank, local_rank, world, st3, frames3, steps=120, min_seconds=min(args.min_seconds, 2.0))
        if rank == 0:
            try:
                ref = single_context_scene(args, CONFIGS["3"], local_rank, st3, frames3, steps=60, min_seconds=min(args.min_seconds, 1.0))
            except Exception as e:   # the reference figure must not cost the other ranks (waiting at the barrier below) their run
                ref = {"value": float("nan"), "error": repr(e)}
            scene = {"frames_per_s": scene["value"], "ms_per_frame": scene["ms_per_step"], "n_gpus": world, "scaling": "strong",
                     "models": scene["config"]["models"], "models_per_rank": scene["config"]["models_per_rank"],
                     "pose_drift_vs_gt_m": scene["config"]["pose_drift_vs_gt_m"], "workload": CONFIGS["3"]["workload"],
                     "one_gpu_one_context": ref, "speedup_vs_one_gpu": scene["value"] / ref["value"],
                     "note": "north star: >= 6x at 8 GPUs on this scene; both figures through host-pointer frames"}
        if world > 1:
            dist.barrier()
    if rank == 0:
        variant = "" if args.icp_weight >= 100.0 and not args.so3 else f" [variant: icpWeight={args.icp_weight:g}, so3={int(args.so3)}]"
        out = {
            "metric": f"frames/sec ({W}x{H} RGB-D, " + (f"background + {n_models - 1} object models" if multi else "single background model") +
                      ", ICP + surfel fusion)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": total_steps, "steps_requested": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * total_dt / total_steps, "timed_seconds": total_dt, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["workload"] + variant, "frames_in_hbm": n_frames, "models": n_models, "surfels": count,
                       "pose_drift_vs_gt_m": drift, "parallelism": f"context-per-gpu x{world}",
                       **({"params": extra_params} if extra_params else {})},
            "roofline": roofline, "roofline_frame": roofline_frame, "host_input": host_input, "cpu_baseline": cpu, "variants": variants,
            "ranks_seen": seen, "sharded_scene": scene,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
