"""GPU: text encoder, length regulation, HiFT vocoder and the end-to-end synthesise() path (HIP through the
C ABI and the mirror classes) against the golden fixtures and the CPU oracle.
Tolerances (BASELINE.json north_star): mel <= 1e-3 max-abs, waveform <= 1e-4 RMS with the source signal injected."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def md(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


def rms(a, b):
    return float((a.float().cpu() - b.float().cpu()).pow(2).mean().sqrt())


@pytest.fixture(scope="module")
def models(tts_sd, hift_sd):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    import jyutvoice_amd
    tts, hift = jyutvoice_amd.build_default("cuda:0")
    tts.load_state_dict(tts_sd)
    hift.load_state_dict(hift_sd)
    return tts, hift


@pytest.fixture(scope="module")
def eng(models):
    from jyutvoice_amd.runtime import get_runtime
    return get_runtime("cuda:0").ensure(4, 512, 128)


def test_encoder_golden(eng):
    g = load_golden("G1_encoder")
    h, mu_x, logw, c = eng.encoder(g["x_ids"], g["x_lengths"], g["lang"], g["tone"], g["word_pos"], g["syllable_pos"],
                                   g["spk_embed"])
    assert md(h, g["x"]) <= 2e-4
    assert md(mu_x, g["mu_x"]) <= 2e-4
    assert md(logw, g["logw"]) <= 2e-4
    assert float(h[1, :, 41:].abs().max()) == 0.0 and float(mu_x[1, :, 41:].abs().max()) == 0.0


def test_speaker_projection(eng, tts_sd):
    import torch.nn.functional as F
    g = load_golden("G1_encoder")
    _, _, _, c = eng.encoder(g["x_ids"], g["x_lengths"], g["lang"], g["tone"], g["word_pos"], g["syllable_pos"], g["spk_embed"])
    want = F.linear(F.normalize(g["spk_embed"], dim=1), tts_sd["spk_embed_affine_layer.weight"],
                    tts_sd["spk_embed_affine_layer.bias"])
    assert md(c, want) <= 1e-5


def test_length_regulation_golden(eng):
    g = load_golden("G2_length")
    xl = g["x_mask"].sum(dim=(1, 2)).long()
    for ls, tag in ((1.0, "ls10"), (0.9, "ls09")):
        w_ceil, yl, attn, mu_y = eng.length_regulate(g["logw"].cuda(), xl, g["mu_x"].cuda(), ls)
        assert torch.equal(yl.cpu(), g[f"y_lengths_{tag}"])
        assert md(w_ceil, g[f"w_ceil_{tag}"]) == 0.0
        assert torch.equal(attn.cpu().to(torch.uint8), g[f"attn_{tag}"].squeeze(1))
        assert md(mu_y, g[f"mu_y_{tag}"]) == 0.0        # a gather: bit-exact


def test_hift_f0_golden(eng):
    g = load_golden("G5_hift")
    f0 = eng.hift_f0(g["mel"])
    assert md(f0, g["f0"]) <= 2e-3 and md(f0, g["f0"]) / float(g["f0"].abs().max()) <= 2e-5


def test_hift_source(eng, hift_sd):
    from oracle import hift as ohift
    g = load_golden("G5_hift")
    noise = g["noise"].float()
    s = eng.hift_source(g["f0"], g["phase"].squeeze(-1), noise)
    want = ohift.source(ohift.fold_weight_norm(hift_sd), g["f0"], g["phase"], noise)
    assert md(s, want) <= 2e-5                 # same fp64-accumulated, fp32-rounded phase as torch.cumsum
    assert md(s, g["s"]) <= 2e-4               # fixture keeps the noise draw in fp16


def test_hift_source_long_and_unvoiced(eng, hift_sd):
    """the phase of the sine generator is torch.cumsum's sequential fp64 sum rounded to fp32 per sample; the kernel gets it in
    parallel from per-frame start values (hiftops.hip: exact whenever the sum's ulp is no coarser than the increment's last
    bit) and re-runs the sequential additions where that fails -- frames with f0 ~ 0 after seconds of audio.  Both paths,
    4 s of audio, against the oracle sample for sample"""
    from oracle import hift as ohift
    g = torch.Generator().manual_seed(91)
    B, T = 2, 200
    f0 = 60.0 + 300.0 * torch.rand(B, T, generator=g)
    f0[0, 40:60] = 0.0                     # unvoiced stretch
    f0[0, 100:110] = 3.0e-5                # increments far below the running sum's ulp: the sequential fallback
    f0[1, 150:170] = 1.0e-3
    f0[1, 5] = 1.0e-7
    phase = (torch.rand(B, 9, 1, generator=g) * 2 - 1) * 3.14159
    phase[:, 0] = 0
    noise = torch.randn(B, 9, 480 * T, generator=g)
    s = eng.hift_source(f0, phase.squeeze(-1), noise)
    want = ohift.source(ohift.fold_weight_norm(hift_sd), f0, phase, noise)
    assert md(s, want) <= 2e-5


def test_hift_source_seeded_noise_is_standard_normal(eng, hift_sd):
    """jv_hift_source_seeded draws the nine N(0,1) values of a sample inside the kernel (Philox4x32-10 + Box-Muller of seed,
    call, utterance, sample) where generator.py:171 calls torch.randn_like.  With f0 = 0 everywhere the sines vanish and
    atanh(s) - bias = (0.1 / 3) sum_h w_h z_h: mean, variance, kurtosis, independence along time / across utterances / across
    calls, and repeatability for a given (seed, call)."""
    from oracle import hift as ohift
    w = ohift.fold_weight_norm(hift_sd)
    lw, lb = w["m_source.l_linear.weight"].double().flatten(), float(w["m_source.l_linear.bias"])
    B, T = 2, 150
    f0 = torch.zeros(B, T)
    phase = torch.zeros(B, 9)
    s = eng.hift_source_seeded(f0, phase, 1234, 0).double().cpu().squeeze(1)
    assert torch.equal(eng.hift_source_seeded(f0, phase, 1234, 0).double().cpu().squeeze(1), s)          # repeatable
    s_call = eng.hift_source_seeded(f0, phase, 1234, 1).double().cpu().squeeze(1)
    s_seed = eng.hift_source_seeded(f0, phase, 1235, 0).double().cpu().squeeze(1)
    sigma = (0.1 / 3.0) * float(lw.pow(2).sum().sqrt())
    z = (torch.atanh(s) - lb) / sigma
    n = z.numel()
    assert abs(float(z.mean())) < 5.0 / n ** 0.5
    assert abs(float(z.var()) - 1.0) < 0.01
    assert abs(float((z ** 4).mean()) - 3.0) < 0.05 and abs(float((z ** 3).mean())) < 0.03      # (z is a sum of 9 weighted normals: normal)
    corr = lambda a, b: float(((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std()))
    assert abs(corr(z[:, 1:].flatten(), z[:, :-1].flatten())) < 0.01             # along time
    assert abs(corr(z[0], z[1])) < 0.01                                          # across utterances
    for other in (s_call, s_seed):
        zo = (torch.atanh(other) - lb) / sigma
        assert abs(corr(z.flatten(), zo.flatten())) < 0.01 and not torch.equal(other, s)
    # tails: P(|z| > 3) = 0.0027
    assert abs(float((z.abs() > 3).double().mean()) - 0.0027) < 0.0006


def test_hift_decode_golden(eng):
    g = load_golden("G5_hift")
    wav = eng.hift_decode(g["mel"], g["s"])
    assert wav.shape == (2, 480 * 16)
    assert rms(wav, g["wav"]) <= 1e-4          # north-star tolerance
    assert rms(wav, g["wav"]) <= 2e-5 and md(wav, g["wav"]) <= 5e-4
    assert float(wav.abs().max()) <= 0.99 + 1e-6


def test_hift_ragged_batch_equals_singles(eng, hift_sd):
    from oracle import hift as ohift
    g = torch.Generator().manual_seed(77)
    T, lens = 24, [24, 13]
    mel = torch.randn(2, 80, T, generator=g) * 1.5
    s = torch.tanh(torch.randn(2, 1, 480 * T, generator=g) * 0.3)
    wav = eng.hift_decode(mel, s, torch.tensor(lens)).cpu()
    w = ohift.fold_weight_norm(hift_sd)
    for b, L in enumerate(lens):
        want = ohift.decode(w, mel[b:b + 1, :, :L], s[b:b + 1, :, :480 * L])
        assert rms(wav[b:b + 1, :480 * L], want) <= 5e-5, b
        if L < T:
            assert float(wav[b, 480 * L:].abs().max()) == 0.0


def test_hift_pair_matches_separate_launches(monkeypatch, hift_sd):
    """A vocoder ResBlock's convolution pair in ONE launch at 64 / 128 channels (hiftpair_kernel.h: the intermediate stays in LDS,
    its fp16x3 scale comes from a load-time bound instead of a measured maximum) against the two hiftconv launches it replaces
    (JV_NO_HIFT_PAIR=1) and against the CPU oracle: a ragged batch (utterance boundaries and masked tails inside tiles), the same
    injected source.  The two forms split the intermediate with different powers of two: they agree to rounding."""
    import jyutvoice_amd
    from oracle import hift as ohift
    g = torch.Generator().manual_seed(123)
    T, lens = 61, [61, 37, 50]
    mel = torch.randn(3, 80, T, generator=g) * 1.5
    s = torch.tanh(torch.randn(3, 1, 480 * T, generator=g) * 0.3)

    def run():
        _, hift = jyutvoice_amd.build_default("cuda:0")
        hift.load_state_dict(hift_sd)
        from jyutvoice_amd.runtime import get_runtime
        return get_runtime("cuda:0").ensure(4, 512, 128).hift_decode(mel, s, torch.tensor(lens)).cpu()

    fused = run()
    monkeypatch.setenv("JV_NO_HIFT_PAIR", "1")
    separate = run()
    assert torch.isfinite(fused).all()
    assert rms(fused, separate) <= 5e-6 and md(fused, separate) <= 2e-4
    assert not torch.equal(fused, separate)      # (different code: identical bits would mean the switch does nothing)
    w = ohift.fold_weight_norm(hift_sd)
    for b, L in enumerate(lens):
        want = ohift.decode(w, mel[b:b + 1, :, :L], s[b:b + 1, :, :480 * L])
        assert rms(fused[b:b + 1, :480 * L], want) <= 5e-5, b
        if L < T:
            assert float(fused[b, 480 * L:].abs().max()) == 0.0


def test_synthesise_golden(models):
    from jyutvoice_amd import synth
    tts, _ = models
    g = load_golden("G9_synthesise")
    one = synth.batch(1, int(g["n_tokens"]))
    res = tts.synthesise(one["x"], one["x_lengths"], one["lang"], one["tone"], one["word_pos"], one["syllable_pos"],
                         one["spk_embed"], None)
    assert set(res) == {"encoder_outputs", "decoder_outputs", "attn", "mel", "mel_lengths", "rtf"}
    assert torch.equal(res["mel_lengths"].cpu(), g["mel_lengths"])
    assert res["attn"].shape == (1, 1, 33, 78) and torch.equal(res["attn"][0, 0].cpu().to(torch.uint8), g["attn"][0])
    assert md(res["encoder_outputs"], g["encoder_outputs"]) <= 2e-4
    assert md(res["mel"], g["mel"]) <= 1e-3      # north-star tolerance
    assert md(res["mel"], g["mel"]) <= 3e-4
    assert isinstance(res["rtf"], float) and res["rtf"] > 0


def test_synthesise_batched_extension(models, tts_sd, noise):
    from jyutvoice_amd import synth
    from oracle import tts as otts
    tts, _ = models
    b = synth.batch(3, 21, first_index=40, lengths=[21, 15, 9])
    args = (b["x"], b["x_lengths"], b["lang"], b["tone"], b["word_pos"], b["syllable_pos"], b["spk_embed"], None)
    with pytest.raises(ValueError, match="requires batch_size=1"):
        tts.synthesise(*args)
    res = tts.synthesise(*args, n_timesteps=4, batched=True)
    for i in range(3):
        L = int(b["x_lengths"][i])
        one = otts.synthesise(tts_sd, noise, b["x"][i:i + 1, :L], b["x_lengths"][i:i + 1], b["lang"][i:i + 1, :L],
                              b["tone"][i:i + 1, :L], b["word_pos"][i:i + 1, :L], b["syllable_pos"][i:i + 1, :L],
                              b["spk_embed"][i:i + 1], None, n_timesteps=4)
        ty = int(one["mel_lengths"][0])
        assert int(res["mel_lengths"][i]) == ty
        assert md(res["mel"][i:i + 1, :, :ty], one["mel"]) <= 5e-4, i


def test_ragged_batch_is_bit_identical_to_singles(models):
    """an utterance's result must not depend on its batch (DP shard == single-process result).
    Three utterances of different lengths, batched vs one at a time: mel and waveform identical bit for bit, in the
    benchmarked contraction mode (per-utterance measured bounds) and in exact-range mode"""
    from jyutvoice_amd import synth
    from jyutvoice_amd.runtime import get_runtime
    tts, hift = models
    lens = [33, 21, 12]
    b = synth.batch(3, 33, first_index=60, lengths=lens)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    try:
        for exact in (False, True):
            get_runtime("cuda:0").ensure(4, 512, 128).set_exact_range(exact)
            res = tts.synthesise(*[b[k] for k in keys], None, n_timesteps=4, batched=True)
            ty = res["mel_lengths"].tolist()
            g = torch.Generator(device="cuda:0").manual_seed(3)
            s = torch.randn(3, 1, 480 * res["mel"].shape[2], device="cuda:0", generator=g) * 0.01
            wav = hift.decode(res["mel"], s, res["mel_lengths"])
            for i, L in enumerate(lens):
                one = tts.synthesise(*[b[k][i:i + 1, :L] if b[k].dim() == 2 and k != "spk_embed" else b[k][i:i + 1] for k in keys],
                                     None, n_timesteps=4)
                assert one["mel"].shape[2] == ty[i]
                assert torch.equal(one["mel"], res["mel"][i:i + 1, :, :ty[i]]), (exact, i, md(one["mel"], res["mel"][i:i + 1, :, :ty[i]]))
                w1 = hift.decode(one["mel"], s[i:i + 1, :, :480 * ty[i]])
                assert torch.equal(w1, wav[i:i + 1, :480 * ty[i]]), (exact, i)
    finally:
        get_runtime("cuda:0").ensure(4, 512, 128).set_exact_range(False)


def test_full_chain_shapes(models):
    from jyutvoice_amd import synth
    tts, hift = models
    one = synth.batch(1, 17, first_index=3)
    res = tts.synthesise(one["x"], one["x_lengths"], one["lang"], one["tone"], one["word_pos"], one["syllable_pos"],
                         one["spk_embed"], None, n_timesteps=3, length_scale=0.9)
    hift.manual_seed(0)
    wav, s = hift.inference(res["mel"])
    T = res["mel"].shape[2]
    assert wav.shape == (1, 480 * T) and s.shape == (1, 1, 480 * T)
    assert torch.isfinite(wav).all() and float(wav.abs().max()) <= 0.99 + 1e-6
    # cache_source continuation overrides the head of the source signal (generator.py:462-465)
    wav2, s2 = hift.inference(res["mel"], cache_source=s[:, :, :960].clone())
    assert torch.equal(s2[:, :, :960], s[:, :, :960])


def test_load_errors(tts_sd):
    import jyutvoice_amd
    tts, _ = jyutvoice_amd.build_default("cuda:0")
    bad = dict(tts_sd)
    bad.pop("dp.proj.bias")
    with pytest.raises(RuntimeError, match="Missing key"):
        tts.load_state_dict(bad)
    bad = dict(tts_sd)
    bad["dp.proj.bias"] = torch.zeros(2)
    with pytest.raises(RuntimeError, match="size mismatch"):
        tts.load_state_dict(bad)
    with pytest.raises(FileNotFoundError):
        tts.load_pretrain("/nonexistent/pretrain.pt")


def test_full_size_batch_invariance(models):
    """BASELINE.json's headline shape (32 utterances x 150 tokens -> 300 frames), where the oracle takes minutes: the
    size-independent properties instead -- every utterance of the batch equals its own single-utterance run (mel, vocoder
    output with the same source), lengths are exact, padding stays zero, everything is finite and inside the audio limit"""
    from jyutvoice_amd import synth
    tts, hift = models
    sd = synth.tts_state_dict(fixed_duration=1.5)
    tts.load_state_dict(sd)
    try:
        B, Tt = 32, 150
        b = synth.batch(B, Tt)
        args = lambda sl: [b[k][sl] for k in ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")] + [None]
        res = tts.synthesise(*args(slice(0, B)), n_timesteps=3, batched=True)
        assert res["mel"].shape == (B, 80, 2 * Tt) and res["mel_lengths"].tolist() == [2 * Tt] * B
        assert torch.isfinite(res["mel"]).all()
        attn = res["attn"]
        assert attn.shape[-2:] == (Tt, 2 * Tt) and float(attn.sum()) == B * 2 * Tt          # one token per frame, exactly
        f0 = hift._engine(B, 2 * Tt).hift_f0(res["mel"], None)
        g = torch.Generator(device="cuda:0").manual_seed(5)
        s = torch.randn(B, 1, 480 * 2 * Tt, device="cuda:0", generator=g) * 0.01
        wav = hift.decode(res["mel"], s)
        assert wav.shape == (B, 480 * 2 * Tt) and torch.isfinite(wav).all() and float(wav.abs().max()) <= 0.99 + 1e-6
        # bit for bit across shardings: the measured fp16x3 bounds are per utterance (ConvGemmArgs::amax_G/S/nb) and a kernel
        # sums a row in the same order wherever the row sits, so an utterance does not see its batch.  Two 16-utterance
        # shards (what data parallelism over 2 GPUs runs) reproduce the 32-utterance result exactly ...
        for lo in (0, 16):
            half = tts.synthesise(*args(slice(lo, lo + 16)), n_timesteps=3, batched=True)
            assert torch.equal(half["mel"], res["mel"][lo:lo + 16]), (lo, md(half["mel"], res["mel"][lo:lo + 16]))
            assert torch.equal(hift.decode(half["mel"], s[lo:lo + 16]), wav[lo:lo + 16]), lo
        # ... and a single utterance, whose transformer linears run on the tile kernels instead of the row-owning GEMM (too
        # few rows to fill the chip: another MFMA shape, another summation order inside a product), agrees to rounding;
        # the vocoder has one kernel set at every size and stays exact
        for i in (0, 17, 31):
            one = tts.synthesise(*args(slice(i, i + 1)), n_timesteps=3)
            assert md(one["mel"], res["mel"][i:i + 1]) <= 2e-5, i
            assert md(hift._engine(1, 2 * Tt).hift_f0(one["mel"], None), f0[i:i + 1]) <= 1e-2
            assert torch.equal(hift.decode(res["mel"][i:i + 1].clone(), s[i:i + 1]), wav[i:i + 1]), i
    finally:
        tts.load_state_dict(synth.tts_state_dict())


def test_full_size_engines_cross_check(models):
    """BASELINE.json's headline shape again, n_timesteps = 10 as benchmarked: the two contraction engines are independent
    implementations of the same fp32 arithmetic (fp16x3 with proven / measured bounds vs bf16x6 everywhere,
    jv_flow_set_contraction), so at the size where the oracle is out of reach they check each other -- inside the north-star
    tolerances (mel 1e-3 max-abs, waveform 1e-4 RMS with the same source signal), on all 32 utterances"""
    from jyutvoice_amd import synth
    from jyutvoice_amd.runtime import get_runtime
    tts, hift = models
    tts.load_state_dict(synth.tts_state_dict(fixed_duration=1.5))
    try:
        B, Tt = 32, 150
        b = synth.batch(B, Tt)
        args = [b[k] for k in ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")] + [None]
        g = torch.Generator(device="cuda:0").manual_seed(11)
        s = torch.randn(B, 1, 480 * 2 * Tt, device="cuda:0", generator=g) * 0.01
        out = {}
        for exact in (False, True):
            get_runtime("cuda:0").ensure(B, 2 * Tt, Tt).set_exact_range(exact)
            res = tts.synthesise(*args, n_timesteps=10, batched=True)
            out[exact] = (res["mel"].clone(), hift.decode(res["mel"], s).clone())
        assert torch.isfinite(out[False][0]).all() and torch.isfinite(out[False][1]).all()
        assert md(out[False][0], out[True][0]) <= 1e-3
        assert rms(out[False][1], out[True][1]) <= 1e-4
        assert not torch.equal(out[False][0], out[True][0])          # the switch does select different kernels
        # the vocoder alone on identical input: isolates its measured-bound path
        get_runtime("cuda:0").ensure(B, 2 * Tt, Tt).set_exact_range(False)
        wav_fast = hift.decode(out[True][0], s)
        assert rms(wav_fast, out[True][1]) <= 2e-5
    finally:
        get_runtime("cuda:0").ensure(32, 300, 150).set_exact_range(False)
        tts.load_state_dict(synth.tts_state_dict())


def test_fused_feed_forward_equals_two_launches(monkeypatch):
    """rowffn_kernel (ff.net.0 -> GELU -> ff.net.2 + residual (+ LayerNorm) in one launch, the hidden tile kept in LDS)
    against the two row-owning launches it replaces (JV_NO_FFN_FUSE=1): same K order in every sum, same GELU, same plane
    split -- the mels must be equal bit for bit, at a tile height of 5 (32 utterances) and of 2 (ragged 8)"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    sd = synth.tts_state_dict(fixed_duration=1.5)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    cases = [synth.batch(32, 150), synth.batch(8, 150, first_index=40, lengths=[150 - 11 * i for i in range(8)])]

    def run():
        tts, _ = jyutvoice_amd.build_default("cuda:0")
        tts.load_state_dict(sd)
        return [tts.synthesise(*[b[k] for k in keys], None, n_timesteps=2, batched=True)["mel"].cpu() for b in cases]

    fused = run()
    monkeypatch.setenv("JV_NO_FFN_FUSE", "1")
    split = run()
    for f, s in zip(fused, split):
        assert torch.isfinite(f).all()
        assert torch.equal(f, s), float((f - s).abs().max())


def test_fused_block_equals_separate_launches(monkeypatch):
    """rowblock_kernel (to_out + residual + LayerNorm3 -> feed-forward pair + residual -> the next block's LayerNorm1 ->
    q | k | v in one launch, the LayerNorm planes kept in LDS) against the three row-owning launches it replaces
    (JV_NO_BLOCK_FUSE=1: rowgemm_wd<res,ln>, rowffn, rowgemm_wa<qkv>): the same K order in every sum and the same epilogue
    expressions -- the mels must be equal bit for bit, at a tile height of 5 (32 utterances), 2 (ragged 8) and 4 (ragged 20,
    with a last workgroup that is partly past the end)"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    sd = synth.tts_state_dict(fixed_duration=1.5)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    cases = [synth.batch(32, 150), synth.batch(8, 150, first_index=40, lengths=[150 - 11 * i for i in range(8)]),
             synth.batch(20, 131, first_index=7, lengths=[131 - 3 * i for i in range(20)])]

    def run():
        tts, _ = jyutvoice_amd.build_default("cuda:0")
        tts.load_state_dict(sd)
        return [tts.synthesise(*[b[k] for k in keys], None, n_timesteps=2, batched=True)["mel"].cpu() for b in cases]

    fused = run()
    monkeypatch.setenv("JV_NO_BLOCK_FUSE", "1")
    split = run()
    for f, s in zip(fused, split):
        assert torch.isfinite(f).all()
        assert torch.equal(f, s), float((f - s).abs().max())


def test_ff_stagger_equals_lockstep(monkeypatch):
    """rowblock_kernel's staggered feed-forward (waves 0..3 half a hidden chunk ahead of waves 4..7, so that one half's GELU
    pass runs under the other half's MFMAs; three barriers per chunk, each half's weight fragments in its own order) against
    the default schedule with every wave in the same phase (the staggered one is opt-in, JV_FF_STAGGER=1: it measured
    slower, rowblock.hip): the same sums in the same K order -- the mels must
    be equal bit for bit, at tile heights 5 (32 utterances), 2 (ragged 8: the q | k | v split regime, rowblock without phase C)
    and 4 (ragged 20, last workgroup partly past the end)"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    sd = synth.tts_state_dict(fixed_duration=1.5)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    cases = [synth.batch(32, 150), synth.batch(8, 150, first_index=40, lengths=[150 - 11 * i for i in range(8)]),
             synth.batch(20, 131, first_index=7, lengths=[131 - 3 * i for i in range(20)])]

    def run():
        tts, _ = jyutvoice_amd.build_default("cuda:0")
        tts.load_state_dict(sd)
        return [tts.synthesise(*[b[k] for k in keys], None, n_timesteps=2, batched=True)["mel"].cpu() for b in cases]

    lock = run()
    monkeypatch.setenv("JV_FF_STAGGER", "1")
    stag = run()
    for a, b in zip(stag, lock):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), float((a - b).abs().max())


def test_split_qkv_equals_fused_block(monkeypatch):
    """Mid-size batches (64 - 170 row tiles on 256 CUs): the next block's q | k | v leaves the fused launch -- phase B's
    epilogue writes the LayerNorm1 planes to HBM and rowgemm_wa_kernel runs with its six column chunks dealt over 3 or 6
    workgroups per row tile (flow.hip `qkv_split`, RowGemmArgs::nsplit).  Same K order and epilogue expressions as the fused
    launch (JV_NO_QKV_SPLIT=1): the mels must be equal bit for bit, at 3 workgroups per row tile (ragged 8), at 6 (ragged 4) and
    with a last row tile that is partly past the end"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    sd = synth.tts_state_dict(fixed_duration=1.5)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    cases = [synth.batch(8, 150, first_index=40, lengths=[150 - 11 * i for i in range(8)]),
             synth.batch(4, 150, first_index=3, lengths=[150, 97, 141, 150]),
             synth.batch(6, 131, first_index=11, lengths=[131 - 7 * i for i in range(6)])]

    def run():
        tts, _ = jyutvoice_amd.build_default("cuda:0")
        tts.load_state_dict(sd)
        return [tts.synthesise(*[b[k] for k in keys], None, n_timesteps=2, batched=True)["mel"].cpu() for b in cases]

    split = run()
    monkeypatch.setenv("JV_NO_QKV_SPLIT", "1")
    fused = run()
    for f, s in zip(fused, split):
        assert torch.isfinite(s).all()
        assert torch.equal(f, s), float((f - s).abs().max())


def test_ln_fold_matches_separate_norm(monkeypatch):
    """A stage's first norm1 runs in the epilogue of the resnet's last convolution (RowConvArgs::ln2_out, rowconv_wd_kernel)
    instead of as a launch of its own (JV_NO_LN_FOLD=1: layernorm256_planes).  The two sum a row's 256 channels in different
    association orders, so the planes may differ in the last bit: the mels agree to 2e-5 (the cross-regime bound), at a
    tile height of 5 (32 utterances) and 2 (ragged 8, where q | k | v also runs column-split), and on the split-K tiles of a
    single utterance (the norm leaves the reduce kernel's tail, rowops.hip splitk_reduce_kernel)"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    sd = synth.tts_state_dict(fixed_duration=1.5)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    cases = [synth.batch(32, 150), synth.batch(8, 150, first_index=40, lengths=[150 - 11 * i for i in range(8)]),
             synth.batch(1, 64, first_index=9)]      # (one utterance: the split-K tiles, whose reduce tail writes the norm)

    def run():
        tts, _ = jyutvoice_amd.build_default("cuda:0")
        tts.load_state_dict(sd)
        return [tts.synthesise(*[b[k] for k in keys], None, n_timesteps=2, batched=True)["mel"].cpu() for b in cases]

    folded = run()
    monkeypatch.setenv("JV_NO_LN_FOLD", "1")
    separate = run()
    for f, s in zip(folded, separate):
        assert torch.isfinite(f).all()
        assert float((f - s).abs().max()) <= 2e-5


def test_res_fold_matches_separate_launch(monkeypatch):
    """A resnet's 1 x 1 res_conv (decoder.py:110-115: output = block2(h) + res_conv(x * mask)) rides in block1's row-owning launch
    as a fourth fragment step per 32-channel chunk (RowConvArgs::res_out, rowconv_wd_kernel<RT, true>) instead of running as a
    64 x 64 tile-kernel launch of its own (JV_NO_RES_FOLD=1).  Same operands and scales; the tile kernel multiplies with
    32x32x16 MFMAs, the row-owning one with 16x16x32 (another summation order inside a product), so the two agree to the
    cross-regime bound of 2e-5 -- at tile heights 5 (32 utterances) and 2 (ragged 8), compact and uniform geometry"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    sd = synth.tts_state_dict(fixed_duration=1.5)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    cases = [synth.batch(32, 150), synth.batch(8, 150, first_index=40, lengths=[150 - 11 * i for i in range(8)]),
             synth.batch(20, 131, first_index=7, lengths=[131 - 3 * i for i in range(20)])]

    def run():
        tts, _ = jyutvoice_amd.build_default("cuda:0")
        tts.load_state_dict(sd)
        return [tts.synthesise(*[b[k] for k in keys], None, n_timesteps=2, batched=True)["mel"].cpu() for b in cases]

    folded = run()
    monkeypatch.setenv("JV_NO_RES_FOLD", "1")
    separate = run()
    for f, s in zip(folded, separate):
        assert torch.isfinite(f).all()
        assert float((f - s).abs().max()) <= 2e-5
        assert not torch.equal(f, s)      # (the two routes are different code: identical bits would mean the switch does nothing)


def test_res_pair_matches_two_launches(monkeypatch):
    """A whole CausalResnetBlock1D in ONE launch (rowres_kernel.h: block1 + res_conv, the time embedding, block2; h2 stays in LDS
    and takes its fp16x3 scale from a bound -- LayerNorm's range from the weights + the embedding's maximum -- instead of a measured
    maximum) against the two row-owning launches it replaces (JV_NO_RES_PAIR=1).  The three fp16 planes hold an fp32 value
    exactly under any power-of-two scale and both routes add the products in the same order, so the mels are EQUAL (the
    2e-5 cross-regime bound is the fallback written here for a low plane that underflows under the smaller scale) -- tile
    heights 5 (32 utterances), 2 (ragged 8), 4 (ragged 20: a last workgroup partly past the end), compact and uniform
    geometry.  The built-in profiler proves the switch changes the launches: 13 of the 14 resnets x 2 steps fused"""
    import jyutvoice_amd
    from jyutvoice_amd import synth, engine
    sd = synth.tts_state_dict(fixed_duration=1.5)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    cases = [synth.batch(32, 150), synth.batch(8, 150, first_index=40, lengths=[150 - 11 * i for i in range(8)]),
             synth.batch(20, 131, first_index=7, lengths=[131 - 3 * i for i in range(20)])]

    def run():
        tts, _ = jyutvoice_amd.build_default("cuda:0")
        tts.load_state_dict(sd)
        mels = [tts.synthesise(*[b[k] for k in keys], None, n_timesteps=2, batched=True)["mel"].cpu() for b in cases]
        engine.profile_enable(True)
        try:
            tts.synthesise(*[cases[0][k] for k in keys], None, n_timesteps=2, batched=True)
            rep = engine.profile_report()
        finally:
            engine.profile_enable(False)
        return mels, rep

    fused, rep_f = run()
    monkeypatch.setenv("JV_NO_RES_PAIR", "1")
    separate, rep_s = run()
    n_f = sum(v["launches"] for k, v in rep_f.items() if k.startswith("rowres_h3"))
    n_s = sum(v["launches"] for k, v in rep_s.items() if k.startswith("rowres_h3"))
    assert n_f == 2 * 13 and n_s == 0, (n_f, n_s)      # (the first resnet's input has no measured bound: never fused)
    for f, s in zip(fused, separate):
        assert torch.isfinite(f).all()
        assert float((f - s).abs().max()) <= 2e-5
    assert torch.equal(fused[0], separate[0])


def test_res_qkv_matches_separate_launch(monkeypatch):
    """The first q | k | v of a stage inside the resnet launch that precedes it (rowres_kernel<RT, true>: the following
    block's LayerNorm1 planes stay in LDS as the operand of a third product, rowblock_kernel's phase C) against its own launch
    over planes written to HBM (rowgemm_wa_kernel, JV_NO_RES_QKV=1).  The same planes, K order and epilogue expressions: the
    mels must be EQUAL -- tile heights 5, 2, 4 as in the test above, compact and uniform geometry.  The profiler shows the
    switch: 13 fused launches per step (the first resnet's input has no measured bound), and only the first stage's q | k | v
    left as a launch of its own"""
    import jyutvoice_amd
    from jyutvoice_amd import synth, engine
    sd = synth.tts_state_dict(fixed_duration=1.5)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    cases = [synth.batch(32, 150), synth.batch(8, 150, first_index=40, lengths=[150 - 11 * i for i in range(8)]),
             synth.batch(20, 131, first_index=7, lengths=[131 - 3 * i for i in range(20)])]

    def run():
        tts, _ = jyutvoice_amd.build_default("cuda:0")
        tts.load_state_dict(sd)
        mels = [tts.synthesise(*[b[k] for k in keys], None, n_timesteps=2, batched=True)["mel"].cpu() for b in cases]
        engine.profile_enable(True)
        try:
            tts.synthesise(*[cases[0][k] for k in keys], None, n_timesteps=2, batched=True)
            rep = engine.profile_report()
        finally:
            engine.profile_enable(False)
        return mels, rep

    def count(rep, prefix, tag):
        return sum(v["launches"] for k, v in rep.items() if k.startswith(prefix) and (tag in k))

    fused, rep_f = run()
    monkeypatch.setenv("JV_NO_RES_QKV", "1")
    separate, rep_s = run()
    assert count(rep_f, "rowres_h3", ",qkv>") == 2 * 13 and count(rep_s, "rowres_h3", ",qkv>") == 0
    assert count(rep_f, "rowgemm_h3", ",qkv>") == 2 * 1 and count(rep_s, "rowgemm_h3", ",qkv>") == 2 * 14
    for f, s in zip(fused, separate):
        assert torch.isfinite(f).all()
        assert torch.equal(f, s)


def test_timestep_embeddings_once_per_solve(monkeypatch):
    """cfm_solve computes the timestep embedding of every Euler step ahead of the loop (three GEMMs of n_timesteps rows) and
    each step copies its row into place; JV_NO_TEMB_PRE=1 computes it inside every step for all 2B (identical) rows, as the
    TRT-seam entry point does for a caller's own t.  GEMM rows are independent of the row count: the mels must be equal bit
    for bit -- one utterance (split-K tiles), a ragged 8 (row-owning kernels), and n_timesteps = 1 and 7"""
    import jyutvoice_amd
    from jyutvoice_amd import synth
    sd = synth.tts_state_dict(fixed_duration=1.5)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    cases = [(synth.batch(1, 64), 7), (synth.batch(8, 150, first_index=40, lengths=[150 - 11 * i for i in range(8)]), 2),
             (synth.batch(2, 33, first_index=5), 1)]

    def run():
        tts, _ = jyutvoice_amd.build_default("cuda:0")
        tts.load_state_dict(sd)
        return [tts.synthesise(*[b[k] for k in keys], None, n_timesteps=n, batched=True)["mel"].cpu() for b, n in cases]

    pre = run()
    monkeypatch.setenv("JV_NO_TEMB_PRE", "1")
    per_step = run()
    for a, b in zip(pre, per_step):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), float((a - b).abs().max())
