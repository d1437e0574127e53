"""CPU: the oracle (oracle/) against the golden fixtures captured from the reference itself."""
import hashlib

import numpy as np
import torch

from conftest import load_golden
from oracle import flow as oflow
from oracle import hift as ohift
from oracle import textenc as otext
from oracle import tts as otts


def md(a, b):
    return float((a.float() - b.float()).abs().max())


def test_noise_known_answer(noise):
    g = load_golden("G7_noise")
    assert torch.equal(noise.flatten()[:16], g["first16"])
    assert abs(float(noise.flatten()[0]) - (-1.1258)) < 1e-4
    digest = np.frombuffer(hashlib.sha256(noise.numpy().tobytes()).digest(), dtype=np.uint8)
    assert np.array_equal(digest, g["sha256"].numpy())


def test_encoder_and_dp(tts_sd):
    g = load_golden("G1_encoder")
    x, mu, mask = otext.text_encoder(tts_sd, g["x_ids"], g["x_lengths"], g["lang"], g["tone"], g["word_pos"],
                                     g["syllable_pos"], g["spk_embed"])
    assert md(x, g["x"]) <= 1e-5 and md(mu, g["mu_x"]) <= 1e-5 and torch.equal(mask, g["x_mask"])
    logw = otext.duration_predictor(tts_sd, x, mask, g["spk_embed"])
    assert md(logw, g["logw"]) <= 1e-5
    # padded token positions are exactly zero
    assert float(x[1, :, 41:].abs().max()) == 0.0 and float(mu[1, :, 41:].abs().max()) == 0.0


def test_length_regulation():
    g = load_golden("G2_length")
    for ls, tag in ((1.0, "ls10"), (0.9, "ls09")):
        w_ceil, yl, attn, mu_y = otext.length_regulate(g["logw"], g["x_mask"], g["mu_x"], ls)
        assert torch.equal(yl, g[f"y_lengths_{tag}"])
        assert torch.equal(attn.to(torch.uint8), g[f"attn_{tag}"].squeeze(1))
        assert md(w_ceil, g[f"w_ceil_{tag}"]) == 0.0
        assert md(mu_y, g[f"mu_y_{tag}"]) <= 1e-6


def test_integer_paths():
    g = load_golden("G8_paths")
    yl, attn = otext.monotonic_path((g["duration"] * g["x_mask"]).unsqueeze(1), g["x_mask"].unsqueeze(1))
    assert torch.equal(yl, g["y_lengths"])
    assert torch.equal(attn.to(torch.uint8), g["path"])
    assert torch.equal(g["pad_mask"], torch.arange(5)[None] >= torch.tensor([5, 3, 2])[:, None])


def test_estimator_ragged_batch(tts_sd):
    g = load_golden("G3_estimator")
    taps = {}
    out = oflow.estimator(tts_sd, g["x"], g["mask"], g["mu"], g["t"], g["spks"], g["cond"], taps=taps)
    assert md(out, g["out"]) <= 2e-5
    assert md(taps["decoder.estimator.down_blocks.0.resnet"], g["down_resnet"]) <= 1e-5
    for k in ("down", "mid0", "mid11", "up"):
        assert md(taps[k] * g["mask"], g[k] * g["mask"]) <= 1e-4, k
    assert float(out[1, :, 20:].abs().max()) == 0.0
    s = load_golden("G6_singles")
    assert md(out[:1], s["out0"]) <= 2e-5 and md(out[1:, :, :20], s["out1"]) <= 2e-5


def test_cfm_loop(tts_sd, noise):
    g = load_golden("G4_cfm")
    T = g["mu"].shape[2]
    for n in (10,):
        assert md(oflow.t_span(n), g[f"t_span_n{n}"]) == 0.0
        mel = oflow.cfm_solve(tts_sd, noise, g["mu"], torch.ones(1, 1, T), g["spks"], torch.zeros(1, 80, T), n)
        assert md(mel, g[f"mel_n{n}"]) <= 1e-4


def test_hift(hift_sd):
    g = load_golden("G5_hift")
    w = ohift.fold_weight_norm(hift_sd)
    assert md(ohift.f0_predict(w, g["mel"]), g["f0"]) <= 1e-4
    s = ohift.source(w, g["f0"], g["phase"], g["noise"].float())
    # the fixture stores the noise draw as fp16: amplitude <= 0.1/3 * |n| -> <= 2e-5 perturbation
    assert md(s, g["s"]) <= 2e-4
    assert md(ohift.stft(g["s"].squeeze(1)), g["s_stft"]) <= 1e-5
    taps = {}
    wav = ohift.decode(w, g["mel"], g["s"], taps=taps)
    assert md(taps["post"], g["post"]) <= 1e-4
    assert float((wav - g["wav"]).pow(2).mean().sqrt()) <= 1e-5
    assert wav.shape == (2, 480 * 16) and float(wav.abs().max()) <= 0.99 + 1e-6


def test_synthesise_end_to_end(tts_sd, noise):
    from jyutvoice_amd import synth
    g = load_golden("G9_synthesise")
    one = synth.batch(1, int(g["n_tokens"]))
    res = otts.synthesise(tts_sd, noise, one["x"], one["x_lengths"], one["lang"], one["tone"], one["word_pos"],
                          one["syllable_pos"], one["spk_embed"], None)
    assert torch.equal(res["mel_lengths"], g["mel_lengths"])
    assert torch.equal(res["attn"].squeeze(1).to(torch.uint8), g["attn"])
    assert md(res["encoder_outputs"], g["encoder_outputs"]) <= 1e-5
    assert md(res["mel"], g["mel"]) <= 1e-4


def test_batch_size_error(tts_sd, noise):
    import pytest
    from jyutvoice_amd import synth
    b = synth.batch(2, 9)
    with pytest.raises(ValueError, match="requires batch_size=1"):
        otts.synthesise(tts_sd, noise, b["x"], b["x_lengths"], b["lang"], b["tone"], b["word_pos"], b["syllable_pos"],
                        b["spk_embed"], None)


def test_streaming_estimator(tts_sd):
    g = load_golden("G10_streaming")
    out = oflow.estimator(tts_sd, g["x"], g["mask"], g["mu"], g["t"], g["spks"], g["cond"], streaming=True)
    assert md(out, g["out"]) <= 2e-5


def test_prompt_encoder(prompt_sd):
    """FlowEncoder (infer.py:35-83) restatement against the imported UpsampleConformerEncoder, two prompt lengths; the
    padded batch is defined as the per-utterance loop (the reference's only usage is B = 1)"""
    from oracle import prompt as oprompt
    g = load_golden("G11_prompt")
    for tag in "ab":
        tok = g["tok_" + tag]
        h, hl = oprompt.flow_encoder(prompt_sd, tok, torch.tensor([tok.shape[1]]))
        assert int(hl[0]) == 2 * tok.shape[1]
        assert md(h, g["h_" + tag]) <= 1e-5
    tok = torch.zeros(2, 50, dtype=torch.int64)
    tok[0], tok[1, :23] = g["tok_b"][0], g["tok_a"][0]
    h, hl = oprompt.flow_encoder(prompt_sd, tok, torch.tensor([50, 23]))
    assert hl.tolist() == [100, 46]
    assert md(h[0:1], g["h_b"]) <= 1e-5 and md(h[1:2, :46], g["h_a"]) <= 1e-5 and float(h[1, 46:].abs().max()) == 0.0
    # rel_shift known answer: out[i, j] = x[i, j - i + T - 1]
    T = 7
    x = torch.arange(T * (2 * T - 1), dtype=torch.float32).view(1, 1, T, 2 * T - 1)
    y = oprompt.rel_shift(x)
    for i in range(T):
        for j in range(T):
            assert float(y[0, 0, i, j]) == float(x[0, 0, i, j - i + T - 1])


def test_prompt_mel():
    """mel_spectrogram / extract_speech_feat restatement against the reference's utils/audio.py run on the same (restated,
    librosa-style) filterbank; and the filterbank's own known answers: Slaney area normalisation, band layout"""
    from oracle import audio as oaudio
    g = load_golden("G12_prompt_mel")
    basis = oaudio.mel_basis_slaney()
    assert basis.shape == (80, 961)
    assert abs(float(basis.double().sum()) - float(g["basis_sum"])) < 1e-9 and torch.equal(basis.max(dim=1).values, g["basis_rowmax"])
    peak = basis.argmax(dim=1)
    assert bool((peak[1:] > peak[:-1]).all()) and int(peak[-1]) * 12.5 < 8000 and float(basis[:, 641:].abs().max()) == 0.0
    mel = oaudio.mel_spectrogram(g["wav"], basis)
    assert md(mel, g["mel"]) <= 1e-6
    feat, n = oaudio.extract_speech_feat(g["wav"], basis)
    assert feat.shape == (1, 50, 80) and int(n[0]) == 50
