"""Batch-sharded alignment across the GPUs of one node (BASELINE config C4).

Utterances never interact (reference core.pyx:44-45: each prange iteration touches only
paths[i] / values[i]), so the path shards with no data-path collective: one process per
GPU aligns its own utterances and the only exchange is one all-gather of the int32
duration vectors at the end (RCCL over xGMI when the tensors live on GPUs, gloo on CPU).

    plan  = lpt_partition(t_x, t_y, world)          # cost-balanced, deterministic
    mine  = plan[rank]                              # utterance indices of this rank
    dur   = align_fn(value[mine], t_x[mine], t_y[mine])     # aligner_amd.align on the GPU
    full  = gather_durations(dur, mine, n_total, Tx)        # [N, Tx] on every rank

`align_fn` is injected so the host logic (partitioning, padding, gather, un-permutation)
is testable on CPU with a gloo group; the product path passes `align_durations`, which
runs the HIP kernels and raises without a GPU.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np
import torch
import torch.distributed as dist


def dp_cost(t_x: np.ndarray, t_y: np.ndarray) -> np.ndarray:
    """In-band cells of the DP, the work of one utterance: t_x * (t_y - t_x + 1)  (core.pyx:18)."""
    t_x = np.asarray(t_x, dtype=np.int64)
    t_y = np.asarray(t_y, dtype=np.int64)
    return t_x * np.maximum(t_y - t_x + 1, 1)


def lpt_partition(t_x: Sequence[int], t_y: Sequence[int], world: int) -> List[np.ndarray]:
    """Longest-processing-time-first assignment of utterances to ranks, with equal shard
    sizes (+-1) so that every rank launches the same padded batch.  Deterministic: ties
    break on the utterance index.  Returns one sorted index array per rank."""
    cost = dp_cost(t_x, t_y)
    n = len(cost)
    order = sorted(range(n), key=lambda i: (-int(cost[i]), i))
    cap = [(n + world - 1 - r) // world for r in range(world)]      # sizes differ by at most one
    load = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min((r for r in range(world) if len(shards[r]) < cap[r]), key=lambda r: (load[r], r))
        shards[r].append(i)
        load[r] += int(cost[i])
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


def align_durations(value: torch.Tensor, t_x: torch.Tensor, t_y: torch.Tensor) -> torch.Tensor:
    """The product align_fn: durations [B,Tx] int32 from the HIP path (GPU tensors)."""
    from .maxpath import align
    return align(value, t_x, t_y, want_path=False, want_durations=True).durations


def boundary_durations(max_duration: int) -> Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor]:
    """An align_fn for sharded_align(): durations [B,Tx] int32 of the MoBoAligner boundary search (its MAP sequence,
    window `max_duration`) instead of maximum_path's -- BASELINE config 5's other search shards the same way, utterance
    by utterance, with the same single gather of int32 durations at the end."""
    def fn(energies: torch.Tensor, t_x: torch.Tensor, t_y: torch.Tensor) -> torch.Tensor:
        from .mobo import boundary_search
        return boundary_search(energies, t_x, t_y, max_duration).durations
    return fn


def gather_durations(local_dur: torch.Tensor, mine: np.ndarray, n_total: int, Tx: int,
                     group=None) -> torch.Tensor:
    """All-gather the per-rank duration blocks and scatter them back into utterance order.

    local_dur [len(mine), Tx] int32 on this rank's device.  Every rank gets [n_total, Tx].
    Shards may differ in size by one: blocks are padded to the largest shard, and the
    index vectors travel in the same collective."""
    world = dist.get_world_size(group)
    cap = (n_total + world - 1) // world
    dev = local_dur.device
    block = torch.zeros((cap, Tx + 1), dtype=torch.int32, device=dev)
    block[:, Tx] = -1                                            # utterance index column, -1 = padding
    k = len(mine)
    if k:
        block[:k, :Tx] = local_dur
        block[:k, Tx] = torch.as_tensor(mine, dtype=torch.int32, device=dev)
    out = torch.empty((world * cap, Tx + 1), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, block, group=group)
    idx = out[:, Tx].long()
    keep = idx >= 0
    full = torch.zeros((n_total, Tx), dtype=torch.int32, device=dev)
    full[idx[keep]] = out[keep, :Tx]
    return full


def sharded_align(value_of: Callable[[np.ndarray], torch.Tensor], t_x: np.ndarray, t_y: np.ndarray, Tx: int,
                  align_fn: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor] = align_durations,
                  device=None, group=None) -> torch.Tensor:
    """Align n utterances across the ranks of `group`; every rank returns all durations [n, Tx].

    value_of(indices) must return this rank's scores [len(indices), Tx, Ty] (already on
    `device`): each rank only ever materialises its own shard."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    plan = lpt_partition(t_x, t_y, world)
    mine = plan[rank]
    if len(mine):
        v = value_of(mine)
        dev = v.device if device is None else torch.device(device)
        tx = torch.as_tensor(np.asarray(t_x)[mine], dtype=torch.int32, device=dev)
        ty = torch.as_tensor(np.asarray(t_y)[mine], dtype=torch.int32, device=dev)
        dur = align_fn(v, tx, ty).to(torch.int32)
    else:
        dev = torch.device("cpu") if device is None else torch.device(device)
        dur = torch.zeros((0, Tx), dtype=torch.int32, device=dev)
    return gather_durations(dur, mine, len(t_x), Tx, group=group)
