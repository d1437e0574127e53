"""GPU: edge shapes the reference's path can see -- one-frame / one-token utterances, lengths that straddle tile
boundaries, ragged batches with a length-1 member, the longest supported sequence -- all against the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def md(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


@pytest.fixture(scope="module")
def eng(tts_sd, hift_sd, noise):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    from jyutvoice_amd.engine import JV_MODEL_HIFT, JV_MODEL_TTS, Engine
    e = Engine("cuda:0", max_batch=4, max_frames=1100, max_tokens=64)
    e.load_state_dict(JV_MODEL_TTS, tts_sd)
    e.load_state_dict(JV_MODEL_HIFT, hift_sd)
    e.load_noise(noise)
    yield e
    e.close()


@pytest.mark.parametrize("T", [1, 2, 3, 31, 33, 64, 65, 129])
def test_estimator_tiny_and_boundary_lengths(eng, tts_sd, T):
    from oracle import flow as oflow
    g = torch.Generator().manual_seed(1000 + T)
    x, mu = torch.randn(2, 80, T, generator=g), torch.randn(2, 80, T, generator=g)
    cond, spks = torch.randn(2, 80, T, generator=g), torch.randn(2, 80, generator=g)
    t = torch.tensor([0.12, 0.12])
    want = oflow.estimator(tts_sd, x, torch.ones(2, 1, T), mu, t, spks, cond)
    out = eng.flow_estimator(x, None, mu, t, spks, cond)
    assert md(out, want) <= 2e-4


def test_estimator_ragged_with_length_one(eng, tts_sd):
    from oracle import flow as oflow
    g = torch.Generator().manual_seed(7)
    T, lens = 70, torch.tensor([70, 1, 64, 33], dtype=torch.int32)
    mask = (torch.arange(T)[None] < lens[:, None]).unsqueeze(1).float()
    x = torch.randn(4, 80, T, generator=g)
    mu, cond = torch.randn(4, 80, T, generator=g) * mask, torch.randn(4, 80, T, generator=g) * mask
    spks, t = torch.randn(4, 80, generator=g), torch.tensor([0.5, 0.5, 0.9, 0.0])
    want = oflow.estimator(tts_sd, x, mask, mu, t, spks, cond)
    out = eng.flow_estimator(x, lens, mu, t, spks, cond)
    assert md(out, want) <= 2e-4
    assert float(out[1, :, 1:].abs().max()) == 0.0


def test_cfm_long_sequence(eng, tts_sd, noise):
    """T = 1100 frames (22 s): more keys than any bench config; 2 Euler steps keep the CPU oracle affordable"""
    from oracle import flow as oflow
    g = torch.Generator().manual_seed(11)
    T = 1100
    mu, spks = torch.randn(1, 80, T, generator=g), torch.randn(1, 80, generator=g)
    cond = torch.zeros(1, 80, T)
    want = oflow.cfm_solve(tts_sd, noise, mu, torch.ones(1, 1, T), spks, cond, 2)
    mel = eng.cfm_solve(mu, None, spks, cond, 2, 1.0)
    assert md(mel, want) <= 3e-4


@pytest.mark.parametrize("T", [1, 2, 5])
def test_hift_tiny(eng, hift_sd, T):
    from oracle import hift as ohift
    g = torch.Generator().manual_seed(50 + T)
    mel = torch.randn(1, 80, T, generator=g)
    s = torch.tanh(torch.randn(1, 1, 480 * T, generator=g) * 0.2)
    w = ohift.fold_weight_norm(hift_sd)
    want = ohift.decode(w, mel, s)
    wav = eng.hift_decode(mel, s)
    assert float((wav.cpu() - want).pow(2).mean().sqrt()) <= 5e-5
    f0 = eng.hift_f0(mel)
    assert md(f0, ohift.f0_predict(w, mel)) / float(ohift.f0_predict(w, mel).abs().max()) <= 2e-5


@pytest.mark.parametrize("Tt,lens", [(1, [1]), (2, [2, 1]), (37, [37, 36, 1])])
def test_encoder_tiny_and_ragged(eng, tts_sd, Tt, lens):
    from jyutvoice_amd import synth
    from oracle import textenc as otext
    b = synth.batch(len(lens), Tt, first_index=200, lengths=lens)
    h, mu_x, logw, _ = eng.encoder(b["x"], b["x_lengths"], b["lang"], b["tone"], b["word_pos"], b["syllable_pos"], b["spk_embed"])
    x_o, mu_o, mask_o = otext.text_encoder(tts_sd, b["x"], b["x_lengths"], b["lang"], b["tone"], b["word_pos"],
                                           b["syllable_pos"], b["spk_embed"])
    logw_o = otext.duration_predictor(tts_sd, x_o, mask_o, b["spk_embed"])
    assert md(h, x_o) <= 3e-4 and md(mu_x, mu_o) <= 3e-4 and md(logw, logw_o) <= 3e-4


def test_capacity_errors_are_loud(eng):
    from jyutvoice_amd._lib import JvError
    with pytest.raises(JvError, match="capacity"):
        eng.flow_estimator(torch.zeros(2, 80, 2000), None, torch.zeros(2, 80, 2000), torch.zeros(2), torch.zeros(2, 80),
                           torch.zeros(2, 80, 2000))
    with pytest.raises(JvError, match="capacity"):
        eng.hift_decode(torch.zeros(9, 80, 4), torch.zeros(9, 1, 1920))


def test_reserve_grows_workspace_and_keeps_weights(eng):
    """jv_reserve: a different capacity re-creates the workspace only -- same context, no weight upload or re-packing --
    and results before and after are identical bit for bit; a call beyond the capacity is refused loudly"""
    g = torch.Generator().manual_seed(21)
    mu = torch.randn(2, 80, 40, generator=g).cuda()
    spks = torch.randn(2, 80, generator=g).cuda()
    cond = torch.zeros(2, 80, 40).cuda()
    handle = eng._h.value
    before = eng.cfm_solve(mu, None, spks, cond, 3, 1.0).clone()
    eng.reserve(7, 1164, 64)
    assert eng._h.value == handle
    assert torch.equal(eng.cfm_solve(mu, None, spks, cond, 3, 1.0), before)
    eng.reserve(1, 64, 32)
    with pytest.raises(Exception, match="capacity"):
        eng.cfm_solve(mu, None, spks, cond, 3, 1.0)
    eng.reserve(4, 1100, 64)
    assert torch.equal(eng.cfm_solve(mu, None, spks, cond, 3, 1.0), before)


def test_runtime_grows_without_reloading(tts_sd, hift_sd):
    """Runtime.ensure on a live runtime keeps its context (one weight upload per load_state_dict)"""
    import jyutvoice_amd
    from jyutvoice_amd.runtime import get_runtime
    tts, hift = jyutvoice_amd.build_default("cuda:0")
    tts.load_state_dict(tts_sd)
    hift.load_state_dict(hift_sd)
    rt = get_runtime("cuda:0")
    e0 = rt.engine
    caps = rt.caps
    e1 = rt.ensure(caps[0] + 1, caps[1] + 64, caps[2] + 32)
    assert e1 is e0 and rt.caps[0] == caps[0] + 1


def test_reserve_is_failure_atomic(eng):
    """a jv_reserve whose workspace cannot be allocated (here: capacities worth terabytes) must leave the context
    at its previous capacities with a complete workspace -- not half-created buffers behind enlarged caps -- so the next call
    at an old shape computes the same bits, and a shape beyond the old capacity is still refused with JV_ERR_SHAPE"""
    from jyutvoice_amd._lib import JvError
    g = torch.Generator().manual_seed(11)
    mu = torch.randn(2, 80, 100, generator=g).cuda()
    spks = torch.randn(2, 80, generator=g).cuda()
    cond = torch.zeros(2, 80, 100).cuda()
    lens = torch.tensor([100, 61], dtype=torch.int32).cuda()
    caps = (eng.max_batch, eng.max_frames, eng.max_tokens)
    want = eng.cfm_solve(mu, lens, spks, cond, 3, 1.0).cpu()
    with pytest.raises(JvError, match="capacities unchanged"):
        eng.reserve(3000, 150000, 64)
    assert not eng.broken()
    assert (eng.max_batch, eng.max_frames, eng.max_tokens) == caps       # the Python mirror did not move either
    assert torch.equal(eng.cfm_solve(mu, lens, spks, cond, 3, 1.0).cpu(), want)
    with pytest.raises(JvError, match="capacity"):
        eng.cfm_solve(torch.randn(caps[0] + 1, 80, 64).cuda(), None, torch.randn(caps[0] + 1, 80).cuda(),
                      torch.zeros(caps[0] + 1, 80, 64).cuda(), 2, 1.0)
