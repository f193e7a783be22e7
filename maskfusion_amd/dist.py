"""Model-per-rank execution over torch.distributed (RCCL on MI355X, gloo in the CPU tests).

MaskFusion's per-model state (surfel map, tracker pyramids, pose) is independent (Core/Model/Model.h:271-323); what the
models share is the incoming frame and, for multi-model scenes, the label image.  One process per GPU owns one model:

    rank 0 owns the input stream  --ONE broadcast(depth 4P B | rgb 3P B, packed)-->  every rank tracks + fuses ITS model
    every rank                    --gather(16 floats: pose, ICP error, inliers, surfels)-->  rank 0 (logging / decisions)

xGMI is point to point (each peer has its own link to rank 0), so the rank-0-rooted broadcast of 2.15 MB (VGA) costs about
size / 153 GB/s = 14 us of link time; the gather is 64 B per rank.  No collective sits inside the per-model work.

The functions here only sequence collectives; the per-rank step is a callable, so the same code is exercised with gloo on
CPU tensors in tests/test_dist_gloo.py and with RCCL + the HIP library in bench.py.
"""
from __future__ import annotations

import contextlib
from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist

STATS_WIDTH = 16  # R(9) t(3) icpError icpCount surfels alive


class FrameBroadcaster:
    """Owns the per-rank frame buffers and moves frame k from rank `src` to every rank: ONE broadcast of one packed buffer per frame
    (depth 4P bytes | rgb 3P bytes, the layout mf_process_frame's own upload uses), as the north star names it."""

    RING = 3  # the library may still read frame k-2 while frame k is published (include/maskfusion_amd.h, input stream)

    def __init__(self, height: int, width: int, device: torch.device, src: int = 0):
        self.src = src
        self.device = device
        self.P = height * width
        self.shape = (height, width)
        self.blocks = [torch.empty(7 * self.P, dtype=torch.uint8, device=device) for _ in range(self.RING)]
        self.k = 0
        self.n_broadcasts = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0

    def views(self, block: torch.Tensor):
        """(rgb HxWx3 uint8, depth HxW float32) views into a packed block (depth first: its 4-byte alignment)"""
        P, (H, W) = self.P, self.shape
        return block[4 * P:].view(H, W, 3), block[:4 * P].view(torch.float32).view(H, W)

    def publish(self, rgb: Optional[torch.Tensor], depth: Optional[torch.Tensor]):
        """rank `src` passes its frame (already on `device`); the others pass None.  Returns (rgb, depth) buffers valid on
        every rank, ordered on the current stream."""
        if self.world == 1:
            return rgb, depth          # nothing to move: the model reads the caller's buffers
        blk = self.blocks[self.k % self.RING]
        self.k += 1
        r, d = self.views(blk)
        if self.rank == self.src:
            r.copy_(rgb, non_blocking=True)
            d.copy_(depth, non_blocking=True)
        dist.broadcast(blk, self.src)
        self.n_broadcasts += 1
        return r, d


class StatsGatherer:
    """gather of one STATS_WIDTH-float record per rank to rank `dst` (north_star: "a gather of per-model residuals")."""

    def __init__(self, device: torch.device, dst: int = 0):
        self.dst = dst
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.mine = torch.zeros(STATS_WIDTH, dtype=torch.float32, device=device)
        self.all = [torch.zeros(STATS_WIDTH, dtype=torch.float32, device=device) for _ in range(self.world)] \
            if self.rank == dst else None

    def gather(self) -> Optional[Sequence[torch.Tensor]]:
        if self.world == 1:
            return [self.mine]
        dist.gather(self.mine, self.all if self.rank == self.dst else None, dst=self.dst)
        return self.all


def run_steps(frames: Callable[[int], tuple], model_step: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], None],
              n_steps: int, height: int, width: int, device: torch.device, stream_ctx=None, input_stream_ctx=None,
              gather_every: int = 1, state=None):
    """The loop bench.py times: for each step rank 0 publishes frame(i) (on the library's input stream), every rank runs
    model_step(rgb, depth, stats_out), stats are gathered to rank 0 (on the library's main stream).  `stream_ctx` /
    `input_stream_ctx` are callables returning a context manager that makes the respective stream torch's current one.
    Returns the list of gathered stats (rank 0) of the LAST step, or None."""
    if state is None:
        state = {}
    bc = state.setdefault("bc", FrameBroadcaster(height, width, device))
    sg = state.setdefault("sg", StatsGatherer(device))
    null = contextlib.nullcontext
    main_ctx = stream_ctx if stream_ctx is not None else null
    in_ctx = input_stream_ctx if input_stream_ctx is not None else main_ctx
    last = None
    for i in range(n_steps):
        with in_ctx():
            rgb, depth = frames(i) if bc.rank == bc.src else (None, None)
            r, d = bc.publish(rgb, depth)
        with main_ctx():
            model_step(r, d, sg.mine)
            if gather_every and (i % gather_every == gather_every - 1 or i == n_steps - 1):
                last = sg.gather()
    return last


def max_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------
# Multi-model scenes sharded by model (SURVEY.md 8e couplings 2 and 3): the z-merged model-id image and the label image.
# The library's projection keys are uint64 (float_bits(z) << 32 | order << 8 | id) with z > 0, i.e. always below 2^63, and
# 0xFFFF...F for "empty"; as int64 the empty key would be -1 and win every MIN, so it is mapped to INT64_MAX for the wire.
# ------------------------------------------------------------------------------------------------
EMPTY_KEY_U64 = 0xFFFFFFFFFFFFFFFF
INT64_MAX = 0x7FFFFFFFFFFFFFFF


def keys_to_wire(keys_u64: torch.Tensor) -> torch.Tensor:
    """uint64 projection keys (viewed as int64 by torch) -> int64 values whose MIN is the nearest surface."""
    k = keys_u64.view(torch.int64)
    return torch.where(k < 0, torch.full_like(k, INT64_MAX), k)


def merge_projection_keys(keys_wire: torch.Tensor) -> torch.Tensor:
    """GlobalProjection across ranks: every rank scatters ITS models into a private key image; one all-reduce(MIN) of
    8 B x P (2.46 MB at VGA: ~28 us of ring time over xGMI) gives every rank the z-merged image."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(keys_wire, op=dist.ReduceOp.MIN)
    return keys_wire


def ids_from_keys(keys_wire: torch.Tensor) -> torch.Tensor:
    """model id per pixel (0 where nothing projects), as k_global_resolve does on one GPU."""
    ids = (keys_wire & 0xFF).to(torch.uint8)
    return torch.where(keys_wire == INT64_MAX, torch.zeros_like(ids), ids)


def broadcast_labels(labels_u8: torch.Tensor, bg_pose16: torch.Tensor, src: int = 0):
    """Rank `src` (background model + label stage) publishes the label image (P bytes) and the camera pose (64 B): static
    objects follow it (Model.h:263), every rank fuses / cleans its models against the labels."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(labels_u8, src)
        dist.broadcast(bg_pose16, src)
    return labels_u8, bg_pose16
