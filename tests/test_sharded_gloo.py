"""The SPMD form of the model-sharded scene (maskfusion_amd/sharded.py, ShardedMaskFusion: SURVEY.md 8e, BASELINE.json configs[3])
under `gloo` with two ranks on the CPU, against the in-process form (LocalGroup) that tests/test_gpu_sharded.py holds bit-identical to
a single multi-model context on the GPU.  The context is tests/fake_mf.py, a toy with the same couplings; what is compared is the
ORCHESTRATION: every model-level call each rank makes, in order, with its arguments; the control records; who owns which model."""
import os
import pickle
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_mf import FakeMaskFusion  # noqa: E402
from maskfusion_amd import sharded  # noqa: E402

W, H, N_FRAMES = 32, 24, 12


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frame(k):
    """frame k of the toy stream: instance 1 (class 41) from frame 2 on, instance 2 (class 99: it will 'jump') from frame 5 on"""
    rgb = np.full((H, W, 3), k, np.uint8)
    depth = np.full((H, W), 1.0 + k, np.float32)
    mask = np.zeros((H, W), np.uint8)
    if k >= 2:
        mask[4:12, 3:11] = 1
    if k >= 5:
        mask[10:20, 18:28] = 2
    return rgb, depth, mask, (0, 41, 99)


def _cfg(track_all):
    return sharded.default_cfg(trackAllModels=track_all, modelSpawnOffset=1, maxModels=8)


def _spmd_worker(rank, world, port, out_dir, track_all):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mf = FakeMaskFusion(W, H)
    sm = sharded.ShardedMaskFusion(mf, torch.device("cpu"), _cfg(track_all))
    ctls = []
    for k in range(N_FRAMES):
        rgb, depth, mask, cls = _frame(k)
        if rank == 0:
            c = sm.process_frame(rgb, depth, mask, cls, timestamp=k)
        else:
            c = sm.process_frame(timestamp=k)      # only rank 0's inputs are used
        ctls.append((c.has_new, c.new_id, c.new_class, c.owner, tuple(c.order)))
    with open(os.path.join(out_dir, f"spmd{rank}.pkl"), "wb") as f:
        pickle.dump(dict(log=mf.log, ctls=ctls, ids=[m.id for m in mf.models], collectives=sm.collectives,
                         table=[(g.id, g.class_id, g.rank) for g in sm.shard.table] if rank == 0 else None), f)
    dist.destroy_process_group()


def _local(track_all):
    mfs = [FakeMaskFusion(W, H) for _ in range(2)]
    shards = [sharded.Shard(r, 2, mfs[r], torch.device("cpu")) for r in range(2)]
    grp = sharded.LocalGroup(shards, _cfg(track_all))
    ctls = []
    for k in range(N_FRAMES):
        rgb, depth, mask, cls = _frame(k)
        c = grp.process_frame(rgb, depth, mask, cls, timestamp=k)
        ctls.append((c.has_new, c.new_id, c.new_class, c.owner, tuple(c.order)))
    return mfs, shards, ctls


@pytest.mark.parametrize("track_all", [True, False], ids=["trackAllModels", "staticObjects"])
def test_spmd_world2_equals_in_process_form(tmp_path, track_all):
    mp.spawn(_spmd_worker, args=(2, _free_port(), str(tmp_path), track_all), nprocs=2, join=True)
    spmd = [pickle.load(open(tmp_path / f"spmd{r}.pkl", "rb")) for r in range(2)]
    mfs, shards, ctls = _local(track_all)
    # every rank received the same control records, and they are the in-process form's
    assert spmd[0]["ctls"] == spmd[1]["ctls"] == ctls
    # every rank made exactly the calls its shard makes in the in-process form (frames reach rank 1 through the broadcast: the
    # ("stage", rgb[0,0,0], depth[0,0]) entries prove the payload arrived)
    for r in range(2):
        assert spmd[r]["log"] == mfs[r].log, r
        assert spmd[r]["ids"] == [m.id for m in mfs[r].models]
    assert spmd[0]["table"] == [(g.id, g.class_id, g.rank) for g in shards[0].table]
    # the latency budget of a sharded frame, in code: ONE packed frame broadcast, the key all-reduce, the state gather, ONE packed
    # post-segmentation broadcast (labels | background state | control) = 4 collectives per frame on every rank; the first frame only
    # publishes the frame; static objects add the early 64-byte pose broadcast (they follow the background's NEW pose, Model.h:263)
    per_frame = 4 if track_all else 5
    assert spmd[0]["collectives"] == spmd[1]["collectives"] == 1 + per_frame * (N_FRAMES - 1)


@pytest.mark.parametrize("track_all", [True, False], ids=["trackAllModels", "staticObjects"])
def test_scene_story(track_all):
    """what the toy scene must do, whichever form runs it: both objects are spawned on rank 1 (rank 0 keeps the background), in list
    order; the class-99 object is dropped everywhere by the jump rule (only when objects are tracked) and its id leaves the table"""
    mfs, shards, ctls = _local(track_all)
    spawns = [(k, c[1], c[2], c[3]) for k, c in enumerate(ctls) if c[0]]
    assert [(s[1], s[2], s[3]) for s in spawns][:2] == [(1, 41, 1), (2, 99, 1)]
    assert [e for e in mfs[0].log if e[0] == "spawn"] == [] and [e[1] for e in mfs[1].log if e[0] == "spawn"][:2] == [1, 2]
    # rank 1 holds a background stand-in that is never tracked, fused or projected
    assert not any(e[0] in ("track", "fuse", "clean", "combinedPredict", "initialise") and e[1] == 0 for e in mfs[1].log)
    assert all(e[1][0] == -1 for e in mfs[1].log if e[0] == "project")
    # the spawn-frame sequence of a new model: predictIndices, fuse at maxDepthProcessed with weight 100, clean (MaskFusion.cpp:342-353)
    log1 = mfs[1].log
    i = log1.index(("spawn", 1, 41))
    assert [e[0] for e in log1[i + 1:i + 4]] == ["predictIndices", "fuse", "clean"] and log1[i + 2][3:5] == (20.0, 100.0)
    dropped = [e for e in mfs[1].log if e[0] == "drop"]
    if track_all:
        # ... after which its mask region is unexplained again and comes back under the next free id (class 99 again: it will jump again)
        assert dropped[0] == ("drop", 2) and all(e[1] >= 2 for e in dropped)
        assert 2 not in [g.id for g in shards[0].table] and 1 in [g.id for g in shards[0].table]
        assert [c[1] for c in ctls if c[0]][:3] == [1, 2, 3]
        assert any(e[0] == "track" and e[1] == 1 for e in log1)
    else:
        # static objects are not tracked: they follow the background pose rank 0 publishes before the projection (Model.h:263)
        assert dropped == [] and not any(e[0] == "track" and e[1] > 0 for e in log1)
        assert any(e == ("static_pose", 1) for e in log1)
        assert np.allclose(mfs[1].models[1].pose, mfs[0].models[0].pose)
