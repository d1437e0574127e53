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
