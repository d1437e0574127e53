"""GPU, at BASELINE.json's full sizes and in the engine mode bench.py runs: the HIP path against the CPU oracle.

C3 (the headline): batch 32 x 150 tokens -> 300 frames, n_timesteps = 10, exactly as bench.py drives it
(fixed_duration = 1.5, batched=True, default contraction = fp16x3 where a bound exists); utterances 0, 17, 31 against
oracle.tts.synthesise (B = 1, what the reference runs) and the vocoder on the HIP mel with the same injected source
against oracle.hift.decode.  C2: the CFM loop alone, 8 x 512 frames, n = 10, one utterance against oracle.flow.cfm_solve.
C4: C3 with n_timesteps = 32, one utterance.  Both contraction engines (jv_flow_set_contraction) are held to the
north-star tolerances -- mel <= 1e-3 max-abs, waveform <= 1e-4 RMS (jyutvoice/models/jyutvoice_tts.py:108-253,
flow/flow_matching.py:215-265, flow/transformer.py:355-443, hifigan/generator.py:396-432).
The measured errors are written to gpurun_out/parity_fullsize.json (committed as profiles/r02_parity_fullsize.json and
quoted in DESIGN.md 3)."""
import json
import os

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu

MEL_TOL, WAV_TOL = 1e-3, 1e-4
RESULTS = {}


def md(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


def rms(a, b):
    return float((a.float().cpu() - b.float().cpu()).pow(2).mean().sqrt())


def record(key, **vals):
    RESULTS.setdefault(key, {}).update({k: float(f"{v:.3e}") for k, v in vals.items()})
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_fullsize.json"), "w") as fh:
        json.dump({"tolerance": {"mel_max_abs": MEL_TOL, "wav_rms": WAV_TOL}, "measured": RESULTS}, fh, indent=1, sort_keys=True)
    print(f"[parity] {key}: " + ", ".join(f"{k}={v:.3e}" for k, v in vals.items()))


@pytest.fixture(scope="module")
def c3(hift_sd, noise):
    """models loaded as bench.py loads them + the oracle's answers for utterances 0, 17, 31 (computed once)"""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    import jyutvoice_amd
    from jyutvoice_amd import synth
    from jyutvoice_amd.runtime import get_runtime
    from oracle import hift as ohift
    from oracle import tts as otts
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    B, Tt = 32, 150
    sd = synth.tts_state_dict(fixed_duration=1.5)
    tts, hift = jyutvoice_amd.build_default("cuda:0")
    get_runtime("cuda:0").ensure(B, 2 * Tt, Tt)
    tts.load_state_dict(sd)
    hift.load_state_dict(hift_sd)
    b = synth.batch(B, Tt)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    ref = {}

    def oracle(i, n):
        if (i, n) not in ref:
            with torch.inference_mode():
                ref[(i, n)] = otts.synthesise(sd, noise, *[b[k][i:i + 1] for k in keys], None, n_timesteps=n)["mel"]
        return ref[(i, n)]

    yield {"tts": tts, "hift": hift, "args": [b[k] for k in keys] + [None], "oracle": oracle, "B": B, "Tt": Tt,
           "hw": ohift.fold_weight_norm(hift_sd), "ohift": ohift, "rt": get_runtime("cuda:0")}
    get_runtime("cuda:0").ensure(B, 2 * Tt, Tt).set_exact_range(False)
    tts.load_state_dict(synth.tts_state_dict())


@pytest.mark.parametrize("exact", [False, True], ids=["fp16x3", "bf16x6"])
def test_c3_headline_vs_oracle(c3, exact):
    B, Tt = c3["B"], c3["Tt"]
    c3["rt"].ensure(B, 2 * Tt, Tt).set_exact_range(exact)
    res = c3["tts"].synthesise(*c3["args"], n_timesteps=10, batched=True)
    mel = res["mel"]
    assert mel.shape == (B, 80, 2 * Tt) and res["mel_lengths"].tolist() == [2 * Tt] * B and torch.isfinite(mel).all()
    c3["hift"].manual_seed(7)
    wav, s = c3["hift"].inference(mel)
    assert wav.shape == (B, 480 * 2 * Tt) and torch.isfinite(wav).all()
    worst_mel = worst_wav = 0.0
    for i in (0, 17, 31):
        e = md(mel[i:i + 1], c3["oracle"](i, 10))
        with torch.inference_mode():
            want = c3["ohift"].decode(c3["hw"], mel[i:i + 1].cpu(), s[i:i + 1].cpu())
        r = rms(wav[i:i + 1], want)
        worst_mel, worst_wav = max(worst_mel, e), max(worst_wav, r)
        record(f"C3 32x300 n=10 {'bf16x6' if exact else 'fp16x3'} utt{i}", mel_max_abs=e, wav_rms=r)
    assert worst_mel <= MEL_TOL, worst_mel
    assert worst_wav <= WAV_TOL, worst_wav


@pytest.mark.parametrize("exact", [False, True], ids=["fp16x3", "bf16x6"])
def test_c4_32_steps_vs_oracle(c3, exact):
    B, Tt = c3["B"], c3["Tt"]
    c3["rt"].ensure(B, 2 * Tt, Tt).set_exact_range(exact)
    res = c3["tts"].synthesise(*c3["args"], n_timesteps=32, batched=True)
    assert torch.isfinite(res["mel"]).all()
    e = md(res["mel"][17:18], c3["oracle"](17, 32))
    record(f"C4 32x300 n=32 {'bf16x6' if exact else 'fp16x3'} utt17", mel_max_abs=e)
    assert e <= MEL_TOL, e


@pytest.mark.parametrize("exact", [False, True], ids=["fp16x3", "bf16x6"])
def test_c2_cfm_loop_vs_oracle(c3, tts_sd, noise, exact):
    """BASELINE.json configs[1] as bench.py --workload c2 builds it: mu ~ N(0,1) [8,80,512], spks ~ N(0,1), cond = 0"""
    from jyutvoice_amd import synth
    from oracle import flow as oflow
    B, T = 8, 512
    sd = synth.tts_state_dict(fixed_duration=1.5)      # what the fixture loaded (the estimator does not depend on dp.*)
    gen = torch.Generator().manual_seed(1234)
    mu = torch.randn(B, 80, T, generator=gen)
    spks = torch.randn(B, 80, generator=gen)
    cond = torch.zeros(B, 80, T)
    eng = c3["rt"].ensure(32, 512, 256)
    eng.set_exact_range(exact)
    mel = eng.cfm_solve(mu.cuda(), None, spks.cuda(), cond.cuda(), 10, 1.0)
    assert torch.isfinite(mel).all()
    i = 5
    with torch.inference_mode():
        want = oflow.cfm_solve(sd, noise, mu[i:i + 1], torch.ones(1, 1, T), spks[i:i + 1], cond[i:i + 1], 10)
    e = md(mel[i:i + 1], want)
    record(f"C2 8x512 n=10 {'bf16x6' if exact else 'fp16x3'} utt{i}", mel_max_abs=e)
    assert e <= MEL_TOL, e
