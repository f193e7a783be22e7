"""The SPMD form of the model-sharded scene end to end on the CPU: two ranks under `gloo`, each with its own context of the product's
kernels EXECUTED ON THE CPU (tests/hipcpu), frames broadcast from rank 0, projection keys all-reduced (MIN), states gathered, labels and
the control record broadcast -- against ONE context holding every model.  Label images, poses, ids, counts and the final surfel clouds
must be bit-identical: sharding moves models, it does not change an operation (the GPU twin is tests/test_gpu_sharded.py,
the orchestration-only twin tests/test_sharded_gloo.py).  Everything runs in subprocesses: the emulated library is test tooling and
never enters this process."""
import os
import pickle
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, F, N_FRAMES = 240, 160, 198.0, 9
SEG = dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphMaskIterations=0,
           newModelMinRelativeSize=0.004)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _activate():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipcpu"))
    import emu
    emu.activate()


def _make(track_all):
    from maskfusion_amd import MaskFusion
    m = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=1 << 17, numOSurfels=1 << 15, enableMultipleModels=True,
                   modelSpawnOffset=2, trackAllModels=track_all)
    for k, v in SEG.items():
        m.setParam(k, v)
    m.setParam("batchTracking", 0)
    return m


def _frames(track_all):
    from maskfusion_amd import synth
    st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2.0, cy=H / 2.0, n_objects=2, noise=True, object_motion=1.0 if track_all else 0.0)
    return [st.frame(k) for k in range(N_FRAMES)]


def _one_context(_rank, out_dir, track_all):
    _activate()
    frames, one, rec = _frames(track_all), _make(track_all), []
    for k, (rgb, depth, mask) in enumerate(frames):
        one.processFrame(rgb, depth, mask=mask, classIDs=[0, 41, 42], timestamp=k)
        ms = one.getModels()
        rec.append(dict(ids=[m.getID() for m in ms], poses=[m.getPose() for m in ms], counts=[m.lastCount() for m in ms], seg=one.downloadSegmentation(),
                        clouds=[m.downloadMap() for m in ms] if k == N_FRAMES - 1 else None))
    one.close()
    pickle.dump(rec, open(os.path.join(out_dir, "one.pkl"), "wb"))


def _spmd(rank, world, port, out_dir, track_all):
    _activate()
    import torch
    import torch.distributed as dist
    from maskfusion_amd import sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mf = _make(track_all)
    sm = sharded.ShardedMaskFusion(mf, torch.device("cpu"), sharded.default_cfg(trackAllModels=track_all, modelSpawnOffset=2))
    frames = _frames(track_all) if rank == 0 else None
    rec = []
    for k in range(N_FRAMES):
        if rank == 0:
            rgb, depth, mask = frames[k]
            sm.process_frame(rgb, depth, mask, [0, 41, 42], 1.0, k)
        else:
            sm.process_frame(timestamp=k)
        ms = mf.getModels()
        own = [(i, m) for i, m in enumerate(ms) if not (rank > 0 and i == 0)]        # rank > 0: model 0 is the background stand-in
        rec.append(dict(ids=[m.getID() for _, m in own], poses=[m.getPose() for _, m in own], counts=[m.lastCount() for _, m in own],
                        seg=mf.downloadSegmentation(), clouds=[m.downloadMap() for _, m in own] if k == N_FRAMES - 1 else None))
    mf.close()
    pickle.dump(rec, open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb"))
    dist.destroy_process_group()


@pytest.mark.parametrize("track_all", [False, True], ids=["static-objects", "tracked-objects"])
def test_two_ranks_equal_one_context(tmp_path, track_all):
    mp.spawn(_one_context, args=(str(tmp_path), track_all), nprocs=1, join=True)
    mp.spawn(_spmd, args=(2, _free_port(), str(tmp_path), track_all), nprocs=2, join=True)
    one = pickle.load(open(tmp_path / "one.pkl", "rb"))
    ranks = [pickle.load(open(tmp_path / f"rank{r}.pkl", "rb")) for r in range(2)]
    most = 0
    for k in range(N_FRAMES):
        want = one[k]
        got = {}
        for r in range(2):
            for j, mid in enumerate(ranks[r][k]["ids"]):
                got[mid] = (ranks[r][k]["poses"][j], ranks[r][k]["counts"][j], ranks[r][k]["clouds"][j] if ranks[r][k]["clouds"] else None, r)
        assert sorted(got) == sorted(want["ids"]), k
        most = max(most, len(ranks[1][k]["ids"]))
        assert np.array_equal(ranks[0][k]["seg"], want["seg"]), k
        if k > 0:
            assert np.array_equal(ranks[1][k]["seg"], want["seg"]), k                  # the label image reached the object rank
        for i, mid in enumerate(want["ids"]):
            pose, count, cloud, r = got[mid]
            assert r == (0 if mid == 0 else 1), (k, mid)                                # objects live on rank 1, the background on rank 0
            assert np.array_equal(pose, want["poses"][i]), (k, mid, np.abs(pose - want["poses"][i]).max())
            assert count == want["counts"][i], (k, mid)
            if want["clouds"] is not None:
                assert np.array_equal(cloud, want["clouds"][i], equal_nan=True), (k, mid)
    # standing boxes: both objects spawn; moving boxes are dropped by the jump rule and re-spawned along the way (identically on both sides)
    assert max(len(r["ids"]) for r in one) >= (2 if track_all else 3) and most >= (1 if track_all else 2), "the scenario must spawn object models, on rank 1"


# ---- N = 1: every model on the one rank (bench.py --config 3 on one GPU).  The loop-level calls then run the SAME batches as
# mf_process_frame -- one Gauss-Newton loop over the background and all objects, one launch per surfel pass for all objects -- so the
# sharded form must reproduce the single context bit for bit with batching ON (the two-rank test above switches batchTracking off,
# because two ranks cannot form the one-context batch).
def _make_batched(track_all):
    from maskfusion_amd import MaskFusion
    m = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=1 << 17, numOSurfels=1 << 15, enableMultipleModels=True,
                   modelSpawnOffset=2, trackAllModels=track_all)
    for k, v in SEG.items():
        m.setParam(k, v)
    return m


def _one_rank_both(_rank, out_dir, track_all):
    _activate()
    import torch
    from maskfusion_amd import sharded
    frames = _frames(track_all)

    def record(mf, k):
        ms = mf.getModels()
        return dict(ids=[m.getID() for m in ms], poses=[m.getPose() for m in ms], counts=[m.lastCount() for m in ms], seg=mf.downloadSegmentation(),
                    clouds=[m.downloadMap() for m in ms] if k == N_FRAMES - 1 else None)

    one, rec_one = _make_batched(track_all), []
    for k, (rgb, depth, mask) in enumerate(frames):
        one.processFrame(rgb, depth, mask=mask, classIDs=[0, 41, 42], timestamp=k)
        rec_one.append(record(one, k))
    one.close()
    mf, rec_sh = _make_batched(track_all), []
    sm = sharded.ShardedMaskFusion(mf, torch.device("cpu"), sharded.default_cfg(trackAllModels=track_all, modelSpawnOffset=2))
    for k, (rgb, depth, mask) in enumerate(frames):
        sm.process_frame(rgb, depth, mask, [0, 41, 42], 1.0, k)
        rec_sh.append(record(mf, k))
    mf.close()
    pickle.dump((rec_one, rec_sh), open(os.path.join(out_dir, "one_rank.pkl"), "wb"))


@pytest.mark.parametrize("track_all", [False, True], ids=["static-objects", "tracked-objects"])
def test_one_rank_sharded_form_equals_one_context(tmp_path, track_all):
    mp.spawn(_one_rank_both, args=(str(tmp_path), track_all), nprocs=1, join=True)
    rec_one, rec_sh = pickle.load(open(tmp_path / "one_rank.pkl", "rb"))
    for k, (a, b) in enumerate(zip(rec_one, rec_sh)):
        assert a["ids"] == b["ids"], k
        assert np.array_equal(a["seg"], b["seg"]), k
        assert a["counts"] == b["counts"], k
        for i in range(len(a["ids"])):
            assert np.array_equal(a["poses"][i], b["poses"][i]), (k, a["ids"][i])
        if a["clouds"] is not None:
            for i in range(len(a["ids"])):
                assert np.array_equal(a["clouds"][i], b["clouds"][i], equal_nan=True), (k, a["ids"][i])
    assert max(len(r["ids"]) for r in rec_one) >= (2 if track_all else 3)


# ---- the loop bench.py --gpus N times (weak scaling: one context per rank, frames broadcast from rank 0, per-model state gathered) ----
def _weak_worker(rank, world, port, out_dir):
    _activate()
    import torch
    import torch.distributed as dist
    from maskfusion_amd import MaskFusion, dist as mfd, synth
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2.0, cy=H / 2.0, noise=True)
    frames = [st.frame(k) for k in range(6)] if rank == 0 else None
    mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 17)
    keep = []

    def get_frame(i):
        return torch.from_numpy(frames[i][0]), torch.from_numpy(frames[i][1])

    def model_step(rgb, depth, stats):                      # bench.py's model_step, device pointers = host pointers here
        keep.append((rgb, depth))
        mf.processFrameDevice(rgb.data_ptr(), depth.data_ptr())
        mf.modelStateDevice(0, stats.data_ptr())

    dev = torch.device("cpu")
    gathered = mfd.run_steps(get_frame, model_step, 6, H, W, dev)
    mf.sync()
    rec = dict(pose=mf.getCurrPose(), count=mf.getBackgroundModel().lastCount(), gathered=[g.numpy().copy() for g in gathered] if gathered is not None and rank == 0 else None)
    mf.close()
    pickle.dump(rec, open(os.path.join(out_dir, f"weak{world}_{rank}.pkl"), "wb"))
    if world > 1:
        dist.destroy_process_group()


def test_weak_scaling_loop_two_ranks(tmp_path):
    """dist.run_steps with the real ABI calls of bench.py (mf_process_frame_dev + mf_model_state_dev) under gloo, world 2: both ranks track
    the broadcast frames to the same bits as a single process, and rank 0 receives both state records"""
    mp.spawn(_weak_worker, args=(1, 0, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_weak_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    one = pickle.load(open(tmp_path / "weak1_0.pkl", "rb"))
    r0, r1 = (pickle.load(open(tmp_path / f"weak2_{r}.pkl", "rb")) for r in range(2))
    for r in (r0, r1):
        assert np.array_equal(r["pose"], one["pose"]) and r["count"] == one["count"]
    g = r0["gathered"]
    assert len(g) == 2 and np.array_equal(g[0], g[1]) and np.array_equal(g[0], one["gathered"][0])
    assert g[0][15] == 1.0 and g[0][14] == one["count"] and np.allclose(g[0][9:12], one["pose"][:3, 3], atol=1e-7)
