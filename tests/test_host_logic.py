"""CPU: host-side logic -- state-dict inventory, synthetic weights, config mirrors, sharding."""
import pytest
import torch

from jyutvoice_amd import spec, synth


def test_inventory_known_answers():
    # README.md:171,233 of the reference: 910 decoder weights, 71.3 M parameters
    est = [k for k in spec.TTS_INVENTORY if k.startswith("decoder.estimator.")]
    assert len(est) == 910 and spec.EST_PARAMS == 71_302_480
    assert len([k for k in spec.TTS_INVENTORY if k.startswith("encoder.")]) == 117
    assert len([k for k in spec.TTS_INVENTORY if k.startswith("dp.")]) == 12
    assert len(spec.HIFT_INVENTORY) == 328
    assert spec.HIFT_INVENTORY["ups.1.parametrizations.weight.original1"] == (256, 128, 11)
    assert spec.HIFT_INVENTORY["f0_predictor.condnet.0.weight_v"] == (512, 80, 3)


def test_synthetic_weights_are_key_hashed():
    a = synth.synth_state_dict(spec.TTS_INVENTORY, prefix="dp.")
    b = synth.synth_state_dict(spec.TTS_INVENTORY, prefix="dp.")
    assert all(torch.equal(a[k], b[k]) for k in a) and len(a) == 12
    sd = synth.tts_state_dict(fixed_duration=1.5)
    assert float(sd["dp.proj.weight"].abs().max()) == 0.0 and abs(float(sd["dp.proj.bias"]) - 0.405465) < 1e-5
    h = synth.synth_state_dict(spec.HIFT_INVENTORY, prefix="conv_pre.")
    v, g = h["conv_pre.parametrizations.weight.original1"], h["conv_pre.parametrizations.weight.original0"]
    ratio = g.flatten() / v.flatten(1).norm(dim=1)
    assert float((ratio - 1).abs().max()) > 1e-3         # the weight-norm fold is genuinely exercised


def test_synthetic_inputs():
    b = synth.batch(3, 11, lengths=[11, 7, 2])
    assert b["x"].shape == (3, 11) and b["x"].dtype == torch.int64
    assert int(b["x"][1, 7:].abs().sum()) == 0 and int(b["x"][0, 0]) == 0 and int(b["x"][0, 1]) > 0
    assert int(b["x"].max()) < spec.ENC_N_VOCAB and int(b["lang"].max()) < spec.ENC_N_LANG
    assert torch.equal(synth.batch(1, 11, first_index=1)["x"][0], synth.batch(2, 11)["x"][1])


def test_config_mirrors_validate_architecture():
    import jyutvoice_amd
    from jyutvoice_amd.flow.decoder import CausalConditionalDecoder
    from jyutvoice_amd.hifigan.generator import HiFTGenerator
    from jyutvoice_amd.models.duration_predictor import DurationPredictor
    jyutvoice_amd.build_default()                        # base.yaml constants construct fine (no GPU needed)
    with pytest.raises(NotImplementedError):
        CausalConditionalDecoder(320, 80, channels=[256, 256])
    with pytest.raises(NotImplementedError):
        DurationPredictor(576, 128, 3, 0.1, 192)
    with pytest.raises(NotImplementedError):
        HiFTGenerator()                                  # the reference's own defaults are not the base.yaml model


def test_synthesise_requires_weights():
    import jyutvoice_amd
    tts, hift = jyutvoice_amd.build_default()
    with pytest.raises(RuntimeError, match="load_state_dict"):
        tts.synthesise(*([torch.zeros(1, 4, dtype=torch.int64)] * 6), torch.zeros(1, 192), None)
    with pytest.raises(RuntimeError, match="load_state_dict"):
        hift.inference(torch.zeros(1, 80, 4))


def test_shard_range():
    from jyutvoice_amd.dist import shard_range
    for n, w in ((256, 8), (10, 4), (3, 8), (32, 1)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_partial_checkpoints_accumulate():
    """load_pretrain semantics (strict=False): flow.pt-style partial checkpoints merge; nothing runs until complete"""
    import jyutvoice_amd
    tts, _ = jyutvoice_amd.build_default()
    sd = synth.tts_state_dict()
    flow_pt = {k: v for k, v in sd.items() if k.startswith(("decoder.", "spk_embed_affine_layer."))}
    flow_pt["optimizer.junk"] = torch.zeros(1)
    missing, unexpected = tts.load_state_dict(flow_pt, strict=False)
    assert len(missing) == 117 + 12 and unexpected == ["optimizer.junk"] and not tts._loaded
    with pytest.raises(RuntimeError, match="129 tensors missing"):
        tts.synthesise(*([torch.zeros(1, 4, dtype=torch.int64)] * 6), torch.zeros(1, 192), None)
    with pytest.raises(RuntimeError, match="size mismatch for dp.proj.bias"):
        tts.load_state_dict({"dp.proj.bias": torch.zeros(3)}, strict=False)


def test_flow_checkpoint_split(tmp_path):
    """scripts/download_pretrain_weights.py:168-214: a CosyVoice2 flow.pt splits into the FlowEncoder part and the part
    JyutVoiceTTS.load_pretrain takes; together with an encoder/dp checkpoint the 1041 TTS tensors are complete"""
    import torch
    from jyutvoice_amd import spec
    from jyutvoice_amd.flow.encoder import extract_flow_weights
    flow_pt = {k: torch.zeros(1) for k in spec.PROMPT_INVENTORY}
    flow_pt.update({k: torch.zeros(1) for k in spec.TTS_INVENTORY if k.startswith(("decoder.", "spk_embed_affine_layer."))})
    flow_pt["length_regulator.model.0.weight"] = torch.zeros(1)       # present in the real file, used by neither part
    enc, dec = extract_flow_weights(flow_pt)
    assert set(enc) == set(spec.PROMPT_INVENTORY)
    assert len(dec) == 910 + 2 and all(k in spec.TTS_INVENTORY for k in dec)
    rest = [k for k in spec.TTS_INVENTORY if k not in dec]
    assert len(rest) == 117 + 12 and all(k.startswith(("encoder.", "dp.")) for k in rest)
    torch.save(enc, tmp_path / "flow_encoder.pt")
    assert set(torch.load(tmp_path / "flow_encoder.pt", weights_only=True)) == set(enc)
