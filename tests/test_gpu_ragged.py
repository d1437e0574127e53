"""GPU: ragged batches in the COMPACT row geometry (flow.hip cfm_solve: utterance b at rows uoff[b] .. + len_b + gap, nothing
padded to the longest; jyutvoice/utils/mask.py:232-255 is the reference's padding of a ragged batch).

Every per-row sum of the estimator is formed in the same order wherever the row sits, the measured bounds are per
utterance and padded frames never enter them: the compact geometry must therefore give the uniform geometry's mel BIT FOR
BIT (JV_NO_COMPACT=1 keeps every utterance padded to the longest).  The full-size case is checked against the CPU oracle run
utterance by utterance (B = 1, what the reference runs), at the north-star tolerances."""
import json
import os

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu

KEYS = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")


def ragged_lengths(B, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    n = torch.randint(lo, hi + 1, (B,), generator=g).tolist()
    n[int(torch.randint(0, B, (1,), generator=g))] = hi
    return n


def test_compact_geometry_equals_uniform(monkeypatch):
    """12 utterances (tile height 3 or 4, q | k | v in its own launch or fused) and 32 utterances (the headline's batch, 69 % fill:
    another tile height than the padded batch takes) with seeded unequal lengths: mel and lengths equal bit for bit"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    sd = synth.tts_state_dict(fixed_duration=1.5)
    cases = [synth.batch(12, 150, first_index=5, lengths=[150, 61, 97, 133, 60, 149, 88, 120, 75, 142, 101, 66]),
             synth.batch(32, 150, lengths=ragged_lengths(32, 60, 150, 4321)),
             synth.batch(9, 131, first_index=3, lengths=[131, 131, 130, 60, 131, 129, 131, 77, 131])]

    def run():
        tts, _ = jyutvoice_amd.build_default("cuda:0")
        tts.load_state_dict(sd)
        out = []
        for b in cases:
            r = tts.synthesise(*[b[k] for k in KEYS], None, n_timesteps=2, batched=True)
            out.append((r["mel"].cpu(), r["mel_lengths"].cpu()))
        return out

    compact = run()
    monkeypatch.setenv("JV_NO_COMPACT", "1")
    uniform = run()
    for (a, la), (b, lb) in zip(compact, uniform):
        assert torch.isfinite(a).all() and torch.equal(la, lb)
        assert torch.equal(a, b), float((a - b).abs().max())


def test_vocoder_compact_geometry_equals_uniform(monkeypatch, hift_sd):
    """HiFT on a ragged batch: the compact geometry (every level's rows are the mel level's times a factor, so the utterances
    lie end to end at every level: hift.hip hift_decode) against the uniform one (JV_NO_COMPACT=1) -- the waveforms are equal bit
    for bit and zero behind each utterance's last sample"""
    import jyutvoice_amd
    from jyutvoice_amd.runtime import get_runtime
    g = torch.Generator().manual_seed(5)
    T, lens = 70, [70, 31, 70, 12, 55, 64]
    mel = torch.randn(len(lens), 80, T, generator=g) * 1.5
    s = torch.tanh(torch.randn(len(lens), 1, 480 * T, generator=g) * 0.3)

    def run():
        _, hift = jyutvoice_amd.build_default("cuda:0")
        hift.load_state_dict(hift_sd)
        return get_runtime("cuda:0").ensure(8, 512, 128).hift_decode(mel, s, torch.tensor(lens)).cpu()

    compact = run()
    monkeypatch.setenv("JV_NO_COMPACT", "1")
    uniform = run()
    assert torch.isfinite(compact).all()
    assert torch.equal(compact, uniform), float((compact - uniform).abs().max())
    for b, L in enumerate(lens):
        if L < T:
            assert float(compact[b, 480 * L:].abs().max()) == 0.0


def test_ragged_fullsize_vs_oracle(hift_sd, noise):
    """32 utterances of 60 .. 150 tokens (bench.py --ragged's lengths), n = 10, encoder -> CFM -> HiFT: the shortest, a middle and
    the longest utterance against oracle.tts.synthesise / oracle.hift.decode run on each alone"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    from jyutvoice_amd.runtime import get_runtime
    from oracle import hift as ohift
    from oracle import tts as otts
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    B, Tt = 32, 150
    toks = ragged_lengths(B, 60, Tt, 4321)
    sd = synth.tts_state_dict(fixed_duration=1.5)
    tts, hift = jyutvoice_amd.build_default("cuda:0")
    get_runtime("cuda:0").ensure(B, 2 * Tt, Tt)
    tts.load_state_dict(sd)
    hift.load_state_dict(hift_sd)
    b = synth.batch(B, Tt, lengths=toks)
    res = tts.synthesise(*[b[k] for k in KEYS], None, n_timesteps=10, batched=True)
    mel, mlen = res["mel"], res["mel_lengths"]
    assert mlen.tolist() == [2 * t for t in toks] and torch.isfinite(mel).all()
    hift.manual_seed(7)
    wav, s = hift.inference(mel, lengths=mlen)
    hw = ohift.fold_weight_norm(hift_sd)
    order = sorted(range(B), key=lambda i: toks[i])
    measured = {}
    for i in (order[0], order[B // 2], order[-1]):
        n = toks[i]
        with torch.inference_mode():
            ref = otts.synthesise(sd, noise, *[b[k][i:i + 1, :n] if b[k].dim() > 1 and k != "spk_embed" else b[k][i:i + 1] for k in KEYS],
                                  None, n_timesteps=10)["mel"]
            T = 2 * n
            e = float((mel[i:i + 1, :, :T].cpu() - ref).abs().max())
            want = ohift.decode(hw, mel[i:i + 1, :, :T].cpu(), s[i:i + 1, :, :480 * T].cpu())
            r = float((wav[i:i + 1, :480 * T].cpu() - want).pow(2).mean().sqrt())
        assert float(mel[i, :, T:].abs().max()) == 0.0 if T < 2 * Tt else True      # padding frames come back as zeros
        measured[f"utt{i} ({n} tokens)"] = {"mel_max_abs": float(f"{e:.3e}"), "wav_rms": float(f"{r:.3e}")}
        assert e <= 1e-3, (i, e)
        assert r <= 1e-4, (i, r)
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_ragged.json"), "w") as fh:
        json.dump({"tolerance": {"mel_max_abs": 1e-3, "wav_rms": 1e-4}, "batch": "32 utterances, 60..150 tokens (seed 4321), n = 10, compact geometry",
                   "measured": measured}, fh, indent=1, sort_keys=True)
