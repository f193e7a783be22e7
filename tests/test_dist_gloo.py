"""world_size-2 gloo run of the model-per-rank loop (maskfusion_amd/dist.py) on CPU: rank 0 broadcasts the frames, every rank
runs a model step on what it received, stats are gathered to rank 0.  The per-rank model step here is the CPU oracle at low
resolution (allowed: tests may use oracle/), which also proves that both ranks saw bit-identical frames."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

W, H, F = 160, 120, 132.0
N_STEPS = 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from maskfusion_amd import dist as mfd, synth
    from oracle import mfo
    dev = torch.device("cpu")
    st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2, cy=H / 2, noise=True)
    frames = [st.frame(k) for k in range(N_STEPS)] if rank == 0 else None
    o = mfo.Oracle(W, H, F, F, W / 2, H / 2, icpWeight=100.0, capacity=W * H * 3, so3=0)
    checks = []

    def get_frame(i):
        return torch.from_numpy(frames[i][0]), torch.from_numpy(frames[i][1])

    def model_step(rgb, depth, stats):
        o.process_frame(rgb.numpy(), depth.numpy())
        checks.append(float(depth.double().sum()) + float(rgb.long().sum()))
        T = o.pose
        e, c = o.icp_stats()
        stats[:9] = torch.from_numpy(T[:3, :3].reshape(-1).astype(np.float32))
        stats[9:12] = torch.from_numpy(T[:3, 3].astype(np.float32))
        stats[12], stats[13], stats[14], stats[15] = e, c, float(o.count), 1.0

    state = {}
    gathered = mfd.run_steps(get_frame, model_step, N_STEPS, H, W, dev, state=state)
    assert state["bc"].n_broadcasts == N_STEPS, "one packed broadcast per frame (rgb and depth travel together)"
    t = mfd.max_over_ranks(float(rank + 1), dev)
    assert t == float(world)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), torch.stack(list(gathered)).numpy())
    np.save(os.path.join(out_dir, f"checks{rank}.npy"), np.array(checks))
    o.close()
    dist.destroy_process_group()


def test_two_ranks_broadcast_track_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = np.load(tmp_path / "gathered.npy")
    c0, c1 = np.load(tmp_path / "checks0.npy"), np.load(tmp_path / "checks1.npy")
    assert np.array_equal(c0, c1), "rank 1 must receive bit-identical frames"
    assert g.shape == (2, 16)
    # both ranks ran the same deterministic model on the same frames -> identical stats, and a plausible pose
    assert np.array_equal(g[0], g[1])
    assert g[0, 15] == 1.0 and g[0, 14] > W * H * 0.9 and abs(g[0, 9]) < 0.05


def test_single_process_path():
    """world = 1 (no process group): publish is a local copy and gather returns the local record."""
    from maskfusion_amd import dist as mfd
    dev = torch.device("cpu")
    seen = []

    def frame(i):
        return torch.full((4, 8, 3), i, dtype=torch.uint8), torch.full((4, 8), float(i))

    def step(rgb, depth, stats):
        seen.append((int(rgb[0, 0, 0]), float(depth[0, 0])))
        stats[0] = depth[0, 0]

    out = mfd.run_steps(frame, step, 3, 4, 8, dev)
    assert seen == [(0, 0.0), (1, 1.0), (2, 2.0)]
    assert float(out[0][0]) == 2.0


def _merge_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from maskfusion_amd import dist as mfd
    Hh, Ww = 24, 32
    rng = np.random.default_rng(100 + rank)
    # every rank projects ITS model: depth image with holes, model id = rank + 1, order = rank
    z = rng.uniform(0.5, 4.0, (Hh, Ww)).astype(np.float32)
    hole = rng.random((Hh, Ww)) < 0.3
    # the library's key layout (include/maskfusion_amd.h, mf_export_projection_keys_dev): float_bits(z) << 32 | order << 8 | id
    z[0, :4] = 1.25                       # an exact depth tie between the ranks: the earlier model in the list (lower order) wins
    hole[0, :4] = False
    key = (z.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.uint64((rank << 8) | (rank + 1))
    key[hole] = np.uint64(mfd.EMPTY_KEY_U64)
    wire = mfd.keys_to_wire(torch.from_numpy(key.view(np.int64).copy()))
    merged = mfd.merge_projection_keys(wire)
    ids = mfd.ids_from_keys(merged).numpy()
    from maskfusion_amd import sharded
    back = sharded._wire_to_keys(merged).numpy().view(np.uint64)      # what mf_import_projection_keys_dev receives
    assert np.array_equal(back == np.uint64(mfd.EMPTY_KEY_U64), ids == 0) and np.array_equal((back & np.uint64(0xFF)).astype(np.uint8)[ids > 0], ids[ids > 0])
    labels = torch.from_numpy((ids * 3).astype(np.uint8)) if rank == 0 else torch.zeros((Hh, Ww), dtype=torch.uint8)
    pose = torch.arange(16, dtype=torch.float32) if rank == 0 else torch.zeros(16)
    labels, pose = mfd.broadcast_labels(labels, pose, 0)
    np.savez(os.path.join(out_dir, f"merge{rank}.npz"), z=z, hole=hole, ids=ids, labels=labels.numpy(), pose=pose.numpy())
    dist.destroy_process_group()


def test_projection_merge_and_label_broadcast_world2(tmp_path):
    """SURVEY.md 8e couplings 2 + 3 on two ranks: all-reduce(MIN) of the packed z/id keys == the nearest model per pixel,
    the label image and the camera pose reach every rank."""
    mp.spawn(_merge_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "merge0.npz"), np.load(tmp_path / "merge1.npz")
    assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["labels"], b["labels"]) and np.array_equal(a["pose"], b["pose"])
    z0 = np.where(a["hole"], np.inf, a["z"]); z1 = np.where(b["hole"], np.inf, b["z"])
    expect = np.where(np.isinf(z0) & np.isinf(z1), 0, np.where(z0 <= z1, 1, 2)).astype(np.uint8)
    assert np.array_equal(a["ids"], expect) and (a["ids"][0, :4] == 1).all()
    assert np.array_equal(a["labels"], expect * 3) and a["pose"].tolist() == list(range(16))
