"""GPU parity of the prompt (voice-cloning) branch: FlowEncoder (infer.py:35-83) through the C ABI (jv_prompt_encoder_fwd)."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def md(a, b):
    return float((a.detach().cpu().float() - b.detach().cpu().float()).abs().max())


@pytest.fixture(scope="module")
def fenc(prompt_sd):
    from jyutvoice_amd.flow.encoder import FlowEncoder
    m = FlowEncoder(vocab_size=6561, input_size=512, output_size=80, device="cuda:0")
    m.load_state_dict(prompt_sd)
    return m


def test_prompt_encoder_golden(fenc):
    """the imported reference encoder's outputs (G11), two prompt lengths"""
    g = load_golden("G11_prompt")
    for tag in "ab":
        tok = g["tok_" + tag]
        h, hl = fenc(tok, torch.tensor([tok.shape[1]]))
        assert h.shape == (1, 2 * tok.shape[1], 80) and int(hl[0]) == 2 * tok.shape[1]
        assert md(h, g["h_" + tag]) <= 5e-5, tag            # measured 6e-6 on outputs of magnitude 4


def test_prompt_ragged_batch_equals_singles(fenc, prompt_sd):
    """a padded batch is the per-utterance loop of the B = 1 reference usage: oracle per utterance, zeros beyond 2*len"""
    from jyutvoice_amd import synth
    from oracle import prompt as oprompt
    tok, lens = synth.prompt_tokens(3, 61, lengths=[61, 17, 40])
    h, hl = fenc(tok, lens)
    want, wl = oprompt.flow_encoder(prompt_sd, tok, lens)
    assert hl.cpu().tolist() == wl.tolist() == [122, 34, 80]
    assert md(h, want) <= 5e-5
    for b, L in enumerate([61, 17, 40]):
        assert float(h[b, 2 * L:].abs().max()) == 0.0 if L < 61 else True
        solo, _ = fenc(tok[b:b + 1, :L], lens[b:b + 1])
        assert md(solo, h[b:b + 1, :2 * L]) <= 2e-5, b          # same kernels, different tile occupancy


def test_prompt_longer_sequence_and_regrow(fenc, prompt_sd):
    """a longer prompt than any earlier call (the workspace regrows) and a 1-token one"""
    from jyutvoice_amd import synth
    from oracle import prompt as oprompt
    tok, lens = synth.prompt_tokens(1, 150, first_index=9)
    h, _ = fenc(tok, lens)
    want, _ = oprompt.flow_encoder(prompt_sd, tok, lens)
    assert md(h, want) <= 1e-4
    tok1, lens1 = synth.prompt_tokens(1, 1, first_index=3)
    h1, _ = fenc(tok1, lens1)
    want1, _ = oprompt.flow_encoder(prompt_sd, tok1, lens1)
    assert h1.shape == (1, 2, 80) and md(h1, want1) <= 2e-4


def test_prompt_registry_and_errors(fenc):
    from jyutvoice_amd import spec
    from jyutvoice_amd._lib import JvError
    from jyutvoice_amd.engine import JV_MODEL_PROMPT
    from jyutvoice_amd.flow.encoder import DIV_TERM_KEY, FlowEncoder
    from jyutvoice_amd.runtime import get_runtime
    eng = get_runtime("cuda:0").ensure(1, 64, 1)
    reg = eng.registry(JV_MODEL_PROMPT)
    assert reg.pop(DIV_TERM_KEY) == (256,)
    assert reg == {k: tuple(v) for k, v in spec.PROMPT_INVENTORY.items()}
    with pytest.raises(NotImplementedError):
        FlowEncoder(vocab_size=1000)
    with pytest.raises(RuntimeError):
        FlowEncoder().forward(torch.zeros(1, 4, dtype=torch.int64), torch.tensor([4]))     # weights not loaded
    with pytest.raises(RuntimeError):
        FlowEncoder().load_state_dict({})
    with pytest.raises(JvError):
        eng.prompt_encoder(torch.zeros(1, 4096, dtype=torch.int64), torch.tensor([4096]))   # beyond the 2048-token limit


def test_prompted_synthesis_uses_prompt_h(fenc, prompt_sd):
    """end to end: prompt tokens -> prompt_h -> synthesise(prompt_feat, prompt_h) against the oracle pipeline"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    from oracle import prompt as oprompt
    from oracle import tts as otts
    tts, _ = jyutvoice_amd.build_default("cuda:0")
    tts_sd = synth.tts_state_dict()
    tts.load_state_dict(tts_sd)
    u = synth.batch(1, 24)
    tok, lens = synth.prompt_tokens(1, 15, first_index=1)
    prompt_h, _ = fenc(tok, lens)                                    # [1, 30, 80]
    g = torch.Generator().manual_seed(3)
    prompt_feat = torch.randn(1, 30, 80, generator=g)
    res = tts.synthesise(u["x"], u["x_lengths"], u["lang"], u["tone"], u["word_pos"], u["syllable_pos"], u["spk_embed"],
                         prompt_feat, prompt_h=prompt_h, n_timesteps=4)
    ph_o, _ = oprompt.flow_encoder(prompt_sd, tok, lens)
    want = otts.synthesise(tts_sd, synth.rand_noise(), u["x"], u["x_lengths"], u["lang"], u["tone"], u["word_pos"],
                           u["syllable_pos"], u["spk_embed"], prompt_feat, prompt_h=ph_o, n_timesteps=4)
    assert res["mel"].shape == want["mel"].shape
    assert md(res["mel"], want["mel"]) <= 1e-3


def test_checkpoint_files_roundtrip(tmp_path, prompt_sd):
    """the reference's file layout (infer.py:209-230, 341-345; jyutvoice_tts.py:73-107): flow.pt split into flow_encoder.pt
    and the decoder part loaded by load_pretrain, then a Lightning-style {"state_dict": ...} checkpoint for the rest"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    from jyutvoice_amd.flow.encoder import extract_flow_weights, load_flow_encoder
    tts_sd = synth.tts_state_dict()
    flow_pt = dict(prompt_sd)
    flow_pt.update({k: v for k, v in tts_sd.items() if k.startswith(("decoder.", "spk_embed_affine_layer."))})
    enc_part, dec_part = extract_flow_weights(flow_pt)
    torch.save(enc_part, tmp_path / "flow_encoder.pt")
    torch.save(dec_part, tmp_path / "pretrain.pt")
    torch.save({"state_dict": {k: v for k, v in tts_sd.items() if k not in dec_part}, "epoch": 3}, tmp_path / "tts.ckpt")
    assert load_flow_encoder(None) is None
    fe = load_flow_encoder(str(tmp_path / "flow_encoder.pt"))
    tok, lens = synth.prompt_tokens(1, 12)
    h, _ = fe(tok, lens)
    tts, _ = jyutvoice_amd.build_default("cuda:0")
    with pytest.raises(FileNotFoundError):
        tts.load_pretrain(str(tmp_path / "nope.pt"))
    missing, unexpected = tts.load_pretrain(str(tmp_path / "pretrain.pt"))
    assert len(missing) == 117 + 12 and not unexpected
    u = synth.batch(1, 16)
    with pytest.raises(RuntimeError):
        tts.synthesise(u["x"], u["x_lengths"], u["lang"], u["tone"], u["word_pos"], u["syllable_pos"], u["spk_embed"], None)
    tts.load_pretrain(str(tmp_path / "tts.ckpt"))
    res = tts.synthesise(u["x"], u["x_lengths"], u["lang"], u["tone"], u["word_pos"], u["syllable_pos"], u["spk_embed"],
                         torch.zeros(1, 24, 80), prompt_h=h, n_timesteps=2)
    assert res["mel"].shape[0] == 1 and torch.isfinite(res["mel"]).all()


def test_cli_with_voice_prompt(tmp_path):
    """infer.py --synthetic with a synthetic voice prompt: the call sequence of the reference CLI incl. infer.py:386-392"""
    import infer
    out = tmp_path / "p.wav"
    infer.main(["--output", str(out), "--synthetic", "20", "--synthetic-prompt", "10", "--n_timesteps", "2"])
    assert out.stat().st_size > 44 + 2 * 480 * 10


def test_prompt_mel_golden():
    """GPU mel front-end (jv_mel_spectrogram) against the reference's utils/audio.py output (G12) and, on a batch and an
    odd length, against the oracle.  Log-mel of a signal with a noise floor: the direct fp32 DFT and torch's FFT agree to
    1e-5 in the log domain"""
    from jyutvoice_amd.utils.audio import extract_speech_feat, mel_spectrogram
    from oracle import audio as oaudio
    g = load_golden("G12_prompt_mel")
    mel = mel_spectrogram(g["wav"])
    assert mel.shape == (1, 80, 50)
    assert md(mel, g["mel"]) <= 5e-5                           # measured 8e-6
    feat, n = extract_speech_feat(g["wav"])
    assert feat.shape == (1, 50, 80) and int(n[0]) == 50 and md(feat.transpose(1, 2), g["mel"]) <= 5e-5
    gen = torch.Generator().manual_seed(7)
    wav = (torch.randn(3, 9000, generator=gen) * 0.2).clamp(-1, 1)
    want = oaudio.mel_spectrogram(wav, oaudio.mel_basis_slaney())
    got = mel_spectrogram(wav)
    assert got.shape == want.shape == (3, 80, 1 + (9000 - 480) // 480)
    assert md(got, want) <= 2e-4
    with pytest.raises(NotImplementedError):
        mel_spectrogram(wav, n_fft=1024)
    from jyutvoice_amd._lib import JvError
    with pytest.raises(JvError):
        mel_spectrogram(torch.zeros(1, 500))
