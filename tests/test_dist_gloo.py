"""CPU, world_size 2 over gloo: the utterance-DP all-gather(v) used by the N > 1 bench path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jyutvoice_amd.dist import all_gather_mels, shard_range
    n_total = 5
    lo, hi = shard_range(n_total, rank, world)                 # ragged shards: 3 + 2
    lens = torch.tensor([7 + 3 * i for i in range(lo, hi)])
    T = int(lens.max())
    mel = torch.zeros(hi - lo, 80, T)
    for j, i in enumerate(range(lo, hi)):
        mel[j, :, : lens[j]] = float(i + 1)
    out, out_l = all_gather_mels(mel, lens)
    q.put((rank, out.shape, out_l.tolist(), [float(out[i, 0, 0]) for i in range(out.shape[0])],
           [float(out[i, :, int(out_l[i]):].abs().sum()) for i in range(out.shape[0])]))
    dist.destroy_process_group()


def test_all_gather_mels_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, shape, lens, firsts, tails in results:
        assert tuple(shape) == (5, 80, 19)                      # T_max over all shards = 7 + 3*4
        assert lens == [7, 10, 13, 16, 19]
        assert firsts == [1.0, 2.0, 3.0, 4.0, 5.0]              # utterance order preserved across ranks
        assert all(t == 0.0 for t in tails)


def _worker_balanced(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jyutvoice_amd.dist import all_gather_mels, balanced_shards, gather_order
    lens_all = [40, 7, 33, 9, 12, 38, 8]                        # a ragged global batch: contiguous halves would hold 89 + 70 frames ...
    shards = balanced_shards(lens_all, world)                   # ... length-aware ones 80 + 67 (+ gaps)
    mine = shards[rank]
    lens = torch.tensor([lens_all[i] for i in mine])
    mel = torch.zeros(len(mine), 80, int(lens.max()))
    for j, i in enumerate(mine):
        mel[j, :, : lens[j]] = float(i + 1)
    out, out_l = all_gather_mels(mel, lens)
    inv = gather_order(shards)
    out, out_l = out.index_select(0, inv), out_l.index_select(0, inv)
    q.put((rank, shards, out_l.tolist(), [float(out[i, 0, 0]) for i in range(out.shape[0])]))
    dist.destroy_process_group()


def test_balanced_shards_world2():
    """length-aware sharding of a ragged batch: both ranks compute the same shards, the loads differ by less than the longest
    utterance, and all-gather + gather_order returns the utterances in their original order"""
    from jyutvoice_amd.dist import balanced_shards
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_balanced, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    lens_all = [40, 7, 33, 9, 12, 38, 8]
    for rank, shards, lens, firsts in results:
        assert shards == results[0][1] and sorted(i for s in shards for i in s) == list(range(7))
        loads = [sum(lens_all[i] + 4 for i in s) for s in shards]
        assert abs(loads[0] - loads[1]) < max(lens_all)
        assert lens == lens_all
        assert firsts == [float(i + 1) for i in range(7)]
    # edge cases: more ranks than utterances, equal lengths (round-robin by load), a single rank
    assert balanced_shards([5], 3) == [[0], [], []]
    assert balanced_shards([10, 10, 10, 10], 2) == [[0, 2], [1, 3]]
    assert balanced_shards([3, 9, 4], 1) == [[0, 1, 2]]
