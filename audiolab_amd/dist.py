"""Multi-GPU sharding of the separation path: one process per GPU, model windows (the overlap
chunks of mdxnet.py:158-163) split into contiguous ranges, and ONE all-gather of the finished
stem segments (RCCL over xGMI when the process group's backend is "nccl"; gloo on CPU for
tests).  The reference has no multi-device path (device is hard-wired, stem_separator.py:99-100):
windows are independent given the mix, so no other exchange exists on this path.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def window_range(n_win: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first ``n_win % world`` ranks get one extra window."""
    base, extra = divmod(n_win, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sample_range(n_win: int, gen: int, n_sample: int, world: int, rank: int) -> Tuple[int, int]:
    """Output samples [lo, hi) of a segment produced by ``rank``'s windows (margin-style
    stitching: window w yields samples [w*gen, (w+1)*gen), the tail is cut at n_sample)."""
    w_lo, w_hi = window_range(n_win, world, rank)
    return min(w_lo * gen, n_sample), min(w_hi * gen, n_sample)


class PendingGather:
    """An all-gather in flight (RCCL runs it on its own stream): ``result()`` waits and assembles."""

    def __init__(self, work, recv, ranges, lead, n_sample):
        self.work, self.recv, self.ranges, self.lead, self.n_sample = work, recv, ranges, lead, n_sample

    def result(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()
        out = torch.empty(self.lead + (self.n_sample,), dtype=self.recv.dtype, device=self.recv.device)
        for r, (l, h) in enumerate(self.ranges):
            out[..., l:h] = self.recv[r][..., : h - l]
        return out


def all_gather_segments(local: torch.Tensor, n_win: int, gen: int, n_sample: int,
                        group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
    """local: [..., n_local] this rank's stem samples (its ``sample_range``) -> [..., n_sample] on
    every rank.  One collective: shards are padded to the largest shard so that a single
    ``all_gather_into_tensor`` moves everything (each peer's shard travels its own xGMI link).
    ``async_op=True`` returns a :class:`PendingGather` so the next model's kernels overlap the transfer."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranges = [sample_range(n_win, gen, n_sample, world, r) for r in range(world)]
    lo, hi = ranges[rank]
    if local.shape[-1] != hi - lo:
        raise ValueError(f"rank {rank}: local shard has {local.shape[-1]} samples, expected {hi - lo}")
    width = max(h - l for l, h in ranges)
    lead = local.shape[:-1]
    send = torch.zeros(lead + (width,), dtype=local.dtype, device=local.device)
    send[..., : hi - lo] = local
    recv = torch.empty((world,) + lead + (width,), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "nccl":
        work = dist.all_gather_into_tensor(recv, send.contiguous(), group=group, async_op=async_op)   # one ncclAllGather (RCCL)
    else:                                                                     # gloo has no _allgather_base
        work = dist.all_gather(list(recv.unbind(0)), send.contiguous(), group=group, async_op=async_op)
    pending = PendingGather(work if async_op else None, recv, ranges, lead, n_sample)
    return pending if async_op else pending.result()


def all_reduce_partial(acc: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Sum of the ranks' weighted partial overlap-add accumulators (Demucs segments with triangular weights, Hann-windowed
    MDX chunks): every rank holds the units of a contiguous range, so only the seam regions carry more than one non-zero
    contribution; one collective (ncclAllReduce on RCCL, gloo in tests)."""
    dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc


def all_gather_ranges(local: torch.Tensor, ranges: List[Tuple[int, int]], total: int,
                      group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """local: [..., hi - lo] this rank's finished samples of its own range ``ranges[rank]`` (ranges: disjoint, ascending, covering
    [0, total)) -> [..., total] on every rank.  ONE all-gather of the stem segments (padded to the longest range), as north_star names it."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = ranges[rank]
    if local.shape[-1] != hi - lo:
        raise ValueError(f"rank {rank}: local segment has {local.shape[-1]} samples, expected {hi - lo}")
    width = max(max(h - l for l, h in ranges), 1)
    lead = local.shape[:-1]
    send = torch.zeros(lead + (width,), dtype=local.dtype, device=local.device)
    send[..., : hi - lo] = local
    recv = torch.empty((world,) + lead + (width,), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    else:
        dist.all_gather(list(recv.unbind(0)), send.contiguous(), group=group)
    out = torch.empty(lead + (total,), dtype=local.dtype, device=local.device)
    for r, (l, h) in enumerate(ranges):
        out[..., l:h] = recv[r][..., : h - l]
    return out


def all_gather_multi_ranges(locals_: List[torch.Tensor], ranges: List[List[Tuple[int, int]]], totals: List[int],
                            group: Optional[dist.ProcessGroup] = None) -> List[torch.Tensor]:
    """Several ``all_gather_ranges`` in ONE collective (the shift passes of the Demucs runner: a rank's range differs per pass):
    locals_[p] is this rank's [..., hi - lo] piece of ranges[p][rank]; every piece is padded to its pass's longest range, the padded
    pieces travel side by side in one buffer.  -> [[..., totals[p]] for p]."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    widths = [max(max(h - l for l, h in rg), 1) for rg in ranges]
    lead = locals_[0].shape[:-1]
    send = torch.zeros(lead + (sum(widths),), dtype=locals_[0].dtype, device=locals_[0].device)
    at = 0
    for loc, rg, w in zip(locals_, ranges, widths):
        lo, hi = rg[rank]
        if loc.shape[-1] != hi - lo or loc.shape[:-1] != lead:
            raise ValueError(f"rank {rank}: a local piece is {tuple(loc.shape)}, expected {tuple(lead) + (hi - lo,)}")
        send[..., at: at + hi - lo] = loc
        at += w
    recv = all_gather_fixed(send, group)
    outs, at = [], 0
    for rg, w, total in zip(ranges, widths, totals):
        out = torch.empty(lead + (total,), dtype=send.dtype, device=send.device)
        for r, (l, h) in enumerate(rg):
            out[..., l:h] = recv[r][..., at: at + h - l]
        outs.append(out)
        at += w
    return outs


def all_gather_fixed(local: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """[...] of the same shape on every rank -> [world, ...] (the seam sums of neighbouring shards: a few MB per rank)"""
    world = dist.get_world_size(group)
    recv = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(recv, local.contiguous(), group=group)
    else:
        dist.all_gather(list(recv.unbind(0)), local.contiguous(), group=group)
    return recv
