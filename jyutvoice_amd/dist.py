"""Utterance-batch data parallelism: one process per GPU, contiguous shards, no exchange during compute, and one
all-gather(v) of the generated mels at the end (RCCL over xGMI on the GPU box; `gloo` in the CPU tests).

The path shards naturally -- utterances are independent units: weights are replicated, the CFG twin
rows of an utterance stay on its GPU, and the fixed CFM noise prefix is identical everywhere."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(n_utts: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) shard of `n_utts` utterances for `rank`; the first n_utts % world ranks get one extra"""
    base, extra = divmod(n_utts, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def balanced_shards(lengths, world: int, gap: int = 4) -> List[List[int]]:
    """Length-aware shards for RAGGED global batches (SURVEY.md 8(e) "optionally length-bucketed"): with the compact row geometry
    a rank's time follows the frames it holds, so contiguous shards of unequal utterances finish at different times.  Longest
    first, each utterance to the rank holding the fewest rows so far (rows = frames + the gap of the row layout; ties: the
    lower rank); indices inside a rank stay in ascending order.  Deterministic: every rank computes the same answer from the
    same lengths.  `gather_order` restores the global order after `all_gather_mels`."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards: List[List[int]] = [[] for _ in range(world)]
    load = [0] * world
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += int(lengths[i]) + gap
    return [sorted(s) for s in shards]


def gather_order(shards: List[List[int]]) -> torch.Tensor:
    """index tensor `inv` with gathered[inv] in the original utterance order, where `gathered` is what all_gather_mels returns
    when rank r contributed the utterances shards[r] (rank-major concatenation)"""
    flat = [i for s in shards for i in s]
    inv = torch.empty(len(flat), dtype=torch.int64)
    inv[torch.tensor(flat, dtype=torch.int64)] = torch.arange(len(flat), dtype=torch.int64)
    return inv


def all_gather_mels(mel: torch.Tensor, lengths: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """mel [b,80,T_local], lengths [b] on every rank -> (mel [sum b,80,T_max], lengths [sum b]) on every rank.

    all-gatherv: lengths and shard sizes first (tiny), then one all-gather of the mels padded to the global
    (max batch, max T) so every rank contributes an equal-size block -- the shape RCCL's all-gather wants."""
    world = dist.get_world_size(group)
    if world == 1:
        return mel, lengths
    meta = torch.tensor([mel.shape[0], mel.shape[2]], dtype=torch.int64, device=mel.device)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    bs = [int(m[0]) for m in metas]
    tmax = max(int(m[1]) for m in metas)
    bmax = max(bs)
    block = torch.zeros(bmax, mel.shape[1], tmax, dtype=mel.dtype, device=mel.device)
    block[: mel.shape[0], :, : mel.shape[2]] = mel
    lens = torch.zeros(bmax, dtype=torch.int64, device=mel.device)
    lens[: lengths.shape[0]] = lengths.to(torch.int64)
    out = torch.empty(world * bmax, mel.shape[1], tmax, dtype=mel.dtype, device=mel.device)
    out_l = torch.empty(world * bmax, dtype=torch.int64, device=mel.device)
    dist.all_gather_into_tensor(out, block, group=group)
    dist.all_gather_into_tensor(out_l, lens, group=group)
    keep = torch.cat([torch.arange(r * bmax, r * bmax + bs[r], device=mel.device) for r in range(world)])
    return out.index_select(0, keep), out_l.index_select(0, keep)
