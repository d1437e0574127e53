"""GPU, at the headline size (C3: 32 x 150 tokens -> 300 frames, n_timesteps = 10), on a HOSTILE checkpoint.

Every other parity test runs the tame synthetic recipe of jyutvoice_amd/synth.py (fan-in-scaled N(0,1), LayerNorm gains
1 +- 0.1), which is exactly the regime where the fp16x3 engine's load-time bounds (registry.hip: sqrt(255) max|g| + max|b|
behind a LayerNorm, row-L1 norms behind a Linear) are tight.  A trained checkpoint has outlier channels and gains >> 1; the
bound then pushes the power-of-two scale down and ordinary operands towards the engine's absolute floor (2^-40 of the
bound).  `synth.hostile_tts_state_dict` / `hostile_hift_state_dict` build that (outlier LayerNorm channels x7.3, gains x2.5 on
five blocks, outlier rows x30 in to_q / to_k / ff.net.0, a constant channel, a nearly-switched-off ff.net.2, Snake alphas from
0.05 to 8), and this file holds the whole path to the north-star tolerances on it, in both engines:
    mel <= 1e-3 max-abs, waveform <= 1e-4 RMS   (jyutvoice/flow/transformer.py:211-219, 355-443; flow/decoder.py:776-781;
                                                 hifigan/generator.py:90-97; transformer/activation.py:73-84).
The mel is compared with the oracle in fp32 (what the reference computes) AND with the same model evaluated in fp64, and the
fp32 oracle's own distance from fp64 is recorded beside them: on a randomly weighted 70-block network that distance is the
noise floor of any comparison (x8 gains make it 1.4 -- see the recipe's docstring -- which is why the recipe stops at x2.5).
Last, a layer whose bound is UNUSABLE (a 1e31 channel gain, compensated in the Linears that read it) must cost that layer
its fp16x3 engine and nothing else: jv_flow_contraction_info reports 55 of 56 blocks unchanged, and the result still meets
the tolerance.  Measured errors go to gpurun_out/parity_hostile.json (committed as profiles/r04_parity_hostile.json)."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import REPO

pytestmark = pytest.mark.gpu

MEL_TOL, WAV_TOL = 1e-3, 1e-4
B, TT, N_STEPS = 32, 150, 10
UTTS = (0, 31)
RESULTS = {}


def md(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def rms(a, b):
    return float((a.double().cpu() - b.double().cpu()).pow(2).mean().sqrt())


def record(key, **vals):
    RESULTS.setdefault(key, {}).update({k: float(f"{v:.3e}") for k, v in vals.items()})
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_hostile.json"), "w") as fh:
        json.dump({"tolerance": {"mel_max_abs": MEL_TOL, "wav_rms": WAV_TOL}, "measured": RESULTS}, fh, indent=1, sort_keys=True)
    print(f"[hostile] {key}: " + ", ".join(f"{k}={v:.3e}" for k, v in vals.items()))


def oracle_mels(sd, noise, batch, i):
    """utterance i through the oracle: (mel fp32, mel of the same model with the CFM loop evaluated in fp64)"""
    from oracle import flow as oflow
    from oracle import tts as otts
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    with torch.inference_mode():
        r = otts.synthesise(sd, noise, *[batch[k][i:i + 1] for k in keys], None, n_timesteps=N_STEPS)
        c = F.linear(F.normalize(batch["spk_embed"][i:i + 1], dim=1), sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"])
        mu = r["encoder_outputs"]
        sd64 = {k: v.double() for k, v in sd.items() if k.startswith("decoder.")}
        m64 = oflow.cfm_solve(sd64, noise.double(), mu.double(), torch.ones(1, 1, mu.shape[2], dtype=torch.float64), c.double(),
                              torch.zeros_like(mu).double(), N_STEPS)
    return r["mel"], m64


@pytest.fixture(scope="module")
def hostile(noise):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    import jyutvoice_amd
    from jyutvoice_amd import synth
    from jyutvoice_amd.runtime import get_runtime
    from oracle import hift as ohift
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = synth.hostile_tts_state_dict(fixed_duration=1.5)
    hsd = synth.hostile_hift_state_dict()
    tts, hift = jyutvoice_amd.build_default("cuda:0")
    rt = get_runtime("cuda:0")
    rt.ensure(B, 2 * TT, TT)
    tts.load_state_dict(sd)
    hift.load_state_dict(hsd)
    batch = synth.batch(B, TT)
    keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
    want = {i: oracle_mels(sd, noise, batch, i) for i in UTTS}
    for i in UTTS:
        record(f"oracle fp32 vs the same model in fp64, utt{i}", mel_max_abs=md(*want[i]))
    yield {"tts": tts, "hift": hift, "rt": rt, "args": [batch[k] for k in keys] + [None], "want": want, "batch": batch,
           "hw": ohift.fold_weight_norm(hsd), "ohift": ohift}
    rt.ensure(B, 2 * TT, TT).set_exact_range(False)
    tts.load_state_dict(synth.tts_state_dict())
    hift.load_state_dict(synth.hift_state_dict())


@pytest.mark.parametrize("exact", [False, True], ids=["fp16x3", "bf16x6"])
def test_hostile_checkpoint_meets_the_tolerances_at_c3(hostile, exact):
    eng = hostile["rt"].ensure(B, 2 * TT, TT)
    eng.set_exact_range(exact)
    info = eng.contraction_info()
    assert info["blocks"] == 56 and info["blocks_all_h3"] == 56 and info["linears_h3"] == 224, info      # every bound usable
    res = hostile["tts"].synthesise(*hostile["args"], n_timesteps=N_STEPS, batched=True)
    mel = res["mel"]
    assert mel.shape == (B, 80, 2 * TT) and torch.isfinite(mel).all()
    hostile["hift"].manual_seed(11)
    wav, s = hostile["hift"].inference(mel)
    assert torch.isfinite(wav).all()
    tag = "bf16x6" if exact else "fp16x3"
    for i in UTTS:
        m32, m64 = hostile["want"][i]
        e32, e64 = md(mel[i:i + 1], m32), md(mel[i:i + 1], m64)
        with torch.inference_mode():
            want = hostile["ohift"].decode(hostile["hw"], mel[i:i + 1].cpu(), s[i:i + 1].cpu())
        r = rms(wav[i:i + 1], want)
        record(f"C3 hostile {tag} utt{i}", mel_max_abs_vs_fp32_oracle=e32, mel_max_abs_vs_fp64_model=e64, wav_rms=r)
        assert e32 <= MEL_TOL and e64 <= MEL_TOL, (i, e32, e64)
        assert r <= WAV_TOL, (i, r)


def test_unusable_bound_costs_that_layer_only(hostile, noise):
    """a 1e31 LayerNorm channel gain (compensated in to_q / to_k / to_v): the bound of that block's q | k | v input is beyond
    what registry.hip accepts, so its q | k | v, attention and to_out run bf16x6; the other 55 blocks keep fp16x3"""
    from jyutvoice_amd import synth
    sd = synth.hostile_tts_state_dict(fixed_duration=1.5, unusable_bound=True)
    hostile["tts"].load_state_dict(sd)
    eng = hostile["rt"].ensure(B, 2 * TT, TT)
    eng.set_exact_range(False)
    info = eng.contraction_info()
    assert info == {"blocks": 56, "blocks_all_h3": 55, "linears_h3": 222, "attention_h3": 55}, info
    res = hostile["tts"].synthesise(*hostile["args"], n_timesteps=N_STEPS, batched=True)
    mel = res["mel"]
    assert torch.isfinite(mel).all()
    i = UTTS[0]
    m32, m64 = oracle_mels(sd, noise, hostile["batch"], i)
    e32, e64, floor = md(mel[i:i + 1], m32), md(mel[i:i + 1], m64), md(m32, m64)
    record(f"C3 hostile + unusable bound, fp16x3 elsewhere, utt{i}", mel_max_abs_vs_fp32_oracle=e32, mel_max_abs_vs_fp64_model=e64,
           oracle_fp32_vs_fp64=floor)
    assert e32 <= MEL_TOL and e64 <= MEL_TOL, (e32, e64)
