"""One rank of the two-rank data-parallel rehearsal (tests/test_gpu_dist.py): synthesise this rank's contiguous shard of a
ragged batch on cuda:0, all-gather the mels over gloo, rank 0 writes the gathered result.  A fresh process per rank, as the
driver launches ranks (one process per GPU); on the 1-GPU box both ranks share the card."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--utts", type=int, default=8)
    ap.add_argument("--tokens", type=int, default=24)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)      # rendezvous before any GPU call
    import jyutvoice_amd
    from jyutvoice_amd import dist as jdist
    from jyutvoice_amd import synth
    tts, _ = jyutvoice_amd.build_default("cuda:0")
    tts.load_state_dict(synth.tts_state_dict())
    lengths = [args.tokens - 3 * (i % 4) for i in range(args.utts)]
    b = synth.batch(args.utts, args.tokens, first_index=200, lengths=lengths)
    lo, hi = jdist.shard_range(args.utts, rank, world)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    res = tts.synthesise(*[b[k][lo:hi] for k in keys], None, n_timesteps=args.steps, batched=True)
    mel, lens = jdist.all_gather_mels(res["mel"].cpu(), res["mel_lengths"].cpu())       # gloo: CPU tensors
    if rank == 0:
        torch.save({"mel": mel, "lens": lens}, args.out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
