"""Model-per-rank execution over torch.distributed (RCCL on MI355X, gloo in the CPU tests).

MaskFusion's per-model state (surfel map, tracker pyramids, pose) is independent (Core/Model/Model.h:271-323); what the
models share is the incoming frame and, for multi-model scenes, the label image.  One process per GPU owns one model:

    rank 0 owns the input stream  --broadcast(rgb 3P B + depth 4P B)-->  every rank tracks + fuses ITS model
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
    """Owns the per-rank frame buffers and moves frame k from rank `src` to every rank."""

    def __init__(self, height: int, width: int, device: torch.device, src: int = 0):
        self.src = src
        self.device = device
        self.rgb = torch.empty((height, width, 3), dtype=torch.uint8, device=device)
        self.depth = torch.empty((height, width), dtype=torch.float32, device=device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0

    def publish(self, rgb: Optional[torch.Tensor], depth: Optional[torch.Tensor]):
        """rank `src` passes its frame (already on `device`); the others pass None.  Returns (rgb, depth) buffers valid on
        every rank, ordered on the current stream."""
        if self.world == 1:
            return rgb, depth          # nothing to move: the model reads the caller's buffers
        if self.rank == self.src:
            self.rgb.copy_(rgb, non_blocking=True)
            self.depth.copy_(depth, non_blocking=True)
        dist.broadcast(self.rgb, self.src)
        dist.broadcast(self.depth, self.src)
        return self.rgb, self.depth


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
              n_steps: int, height: int, width: int, device: torch.device, stream_ctx=None, gather_every: int = 1):
    """The loop bench.py times: for each step rank 0 publishes frame(i), every rank runs model_step(rgb, depth, stats_out),
    stats are gathered to rank 0.  Returns the list of gathered stats (rank 0) of the LAST step, or None."""
    bc = FrameBroadcaster(height, width, device)
    sg = StatsGatherer(device)
    ctx = stream_ctx if stream_ctx is not None else contextlib.nullcontext()
    last = None
    with ctx:
        for i in range(n_steps):
            rgb, depth = frames(i) if bc.rank == bc.src else (None, None)
            r, d = bc.publish(rgb, depth)
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
