"""Oracle (test infrastructure): JyutVoiceTTS.synthesise orchestration.

Follows jyutvoice/models/jyutvoice_tts.py:171-253.  `batched=True` is the documented extension:
identical to looping the (batch-1-only) reference over utterances."""
import torch
import torch.nn.functional as F

from . import flow, textenc


def synthesise(sd, noise, x, x_lengths, lang, tone, word_pos, syllable_pos, spk_embed, prompt_feat=None,
               prompt_h=None, n_timesteps=10, temperature=1.0, length_scale=1.0, batched=False, timings=None):
    """timings (optional dict): wall-clock seconds of the two stages are added to timings["encoder_dp"] / timings["cfm"]
    (bench.py's per-stage CPU baseline)"""
    import time
    t0 = time.perf_counter()
    c = F.linear(F.normalize(spk_embed, dim=1), sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"])
    h, mu_x, x_mask = textenc.text_encoder(sd, x, x_lengths, lang, tone, word_pos, syllable_pos, spk_embed)
    logw = textenc.duration_predictor(sd, h, x_mask, spk_embed)
    w_ceil, y_lengths, attn, mu_y = textenc.length_regulate(logw, x_mask, mu_x, length_scale)
    encoder_outputs = mu_y
    if x.shape[0] != 1 and not batched:
        raise ValueError(f"synthesise() requires batch_size=1, got batch_size={x.shape[0]}. "
                         "Please pass one sample at a time.")
    mel_len1 = 0
    if prompt_feat is not None and prompt_h is not None:
        mu_y = torch.cat([prompt_h.transpose(1, 2), mu_y], dim=2)
        mel_len1 = prompt_feat.shape[1]
        total = mu_y.shape[2]
        conds = torch.zeros([x.shape[0], total, 80])
        conds[:, :mel_len1] = prompt_feat
        conds = conds.transpose(1, 2)
        lens = torch.full((x.shape[0],), total, dtype=torch.int64) if x.shape[0] == 1 else y_lengths + mel_len1
    else:
        conds = torch.zeros_like(mu_y)
        lens = y_lengths
    mask = (torch.arange(mu_y.shape[2]).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1).to(mu_y.dtype)
    t1 = time.perf_counter()
    dec = flow.cfm_solve(sd, noise, mu_y, mask, c, conds, n_timesteps, temperature)
    dec = dec[:, :, mel_len1:]
    if timings is not None:
        timings["encoder_dp"] = timings.get("encoder_dp", 0.0) + (t1 - t0)
        timings["cfm"] = timings.get("cfm", 0.0) + (time.perf_counter() - t1)
    return {"encoder_outputs": encoder_outputs, "decoder_outputs": dec, "attn": attn.unsqueeze(1),
            "mel": dec, "mel_lengths": y_lengths, "logw": logw, "w_ceil": w_ceil, "mu_x": mu_x, "x": h}
