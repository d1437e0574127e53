"""`JyutVoiceTTS` drop-in for the synthesis path (jyutvoice/models/jyutvoice_tts.py:23-253).

Same constructor keywords, `load_state_dict` / `load_pretrain` key names, `synthesise()` signature, defaults,
return-dict keys and `ValueError` for batch != 1 as the reference; training (`forward`, Lightning hooks) is out of
scope.  All arithmetic runs in libjyutvoice_hip.so; this class only moves pointers.

Extension (opt-in): `synthesise(..., batched=True)` accepts B > 1 and is defined as looping the batch-1
reference over the utterances (padded frames of shorter utterances are returned as zeros).
"""
from __future__ import annotations

import datetime as dt
import os
from typing import Dict

import torch

from .. import spec
from ..engine import JV_MODEL_TTS
from ..flow.flow_matching import CausalConditionalCFM
from ..runtime import get_runtime
from .duration_predictor import DurationPredictor
from .text_encoder import TextEncoder


class JyutVoiceTTS:
    def __init__(self, encoder: "TextEncoder", decoder: "CausalConditionalCFM", dp: "DurationPredictor", output_size=80,
                 spk_embed_dim=192, freeze_encoder=False, freeze_decoder=False, optimizer=None, scheduler=None,
                 pretrain_path=None, warmup_steps=100, device="cuda:0"):
        if output_size != spec.N_FEATS or spk_embed_dim != spec.SPK_EMBED_DIM:
            raise NotImplementedError("libjyutvoice_hip is built for output_size=80, spk_embed_dim=192")
        self.encoder, self.decoder, self.dp = encoder, decoder, dp
        self.n_feats = getattr(encoder, "n_feats", spec.N_FEATS)
        self.output_size = output_size
        self.freeze_encoder, self.freeze_decoder = freeze_encoder, freeze_decoder
        self.device = torch.device(device)
        if hasattr(decoder, "device"):
            decoder.device = self.device
        self._loaded = False
        self._sd: Dict[str, torch.Tensor] = {}
        self._missing = list(spec.TTS_INVENTORY)
        if pretrain_path:
            self.load_pretrain(pretrain_path)

    # ---- nn.Module-shaped plumbing used by infer.py:341-346 -----------------------------------------
    def to(self, device):
        self.device = torch.device(device)
        if hasattr(self.decoder, "device"):
            self.decoder.device = self.device
        return self

    def eval(self):
        return self

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Same key names / shapes / error text as the reference module.  With strict=False (what `load_pretrain` uses,
        jyutvoice_tts.py:104) keys accumulate across calls -- e.g. CosyVoice2's flow.pt (`decoder.*`,
        `spk_embed_affine_layer.*`) first, a fine-tuned encoder/dp checkpoint later -- and the weights go to the GPU once
        all 1041 tensors are known; until then synthesise() names what is missing (the reference would silently run the
        missing sub-modules with their random initialisation)."""
        missing = [k for k in spec.TTS_INVENTORY if k not in state_dict]
        unexpected = [k for k in state_dict if k not in spec.TTS_INVENTORY]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for JyutVoiceTTS: Missing key(s): {missing[:6]}; "
                               f"Unexpected key(s): {unexpected[:6]}")
        for k, v in state_dict.items():
            if k in spec.TTS_INVENTORY and tuple(v.shape) != tuple(spec.TTS_INVENTORY[k]):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from checkpoint, "
                                   f"the shape in current model is {tuple(spec.TTS_INVENTORY[k])}.")
        self._sd.update({k: v.detach() for k, v in state_dict.items() if k in spec.TTS_INVENTORY})
        self._missing = [k for k in spec.TTS_INVENTORY if k not in self._sd]
        if not self._missing:
            get_runtime(self.device).set_weights(JV_MODEL_TTS, {k: self._sd[k] for k in spec.TTS_INVENTORY})
            self._loaded = True
        return missing, unexpected

    def load_pretrain(self, pretrain_path):
        if not os.path.exists(pretrain_path):
            raise FileNotFoundError(f"Pretrain checkpoint not found: {pretrain_path}")
        ckpt = torch.load(pretrain_path, map_location="cpu")
        sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
        return self.load_state_dict(sd, strict=False)

    # ---- the hot path -----------------------------------------------------------------------------------------
    @torch.inference_mode()
    def synthesise(self, x, x_lengths, lang, tone, word_pos, syllable_pos, spk_embed, prompt_feat, prompt_h=None,
                   n_timesteps=10, temperature=1.0, length_scale=1.0, batched=False):
        if not self._loaded:
            raise RuntimeError(f"JyutVoiceTTS: load_state_dict() has not provided all weights yet; {len(self._missing)} tensors "
                               f"missing, e.g. {self._missing[:4]}")
        t0 = dt.datetime.now()
        B, Tt = x.shape
        rt = get_runtime(self.device)
        eng = rt.ensure(B, 64, Tt)
        # stage_events (a list, set by bench.py for its per-stage pass; None otherwise): events on the launch stream at the
        # start, after encoder + duration predictor + length regulation, and after the CFM loop
        ev = getattr(self, "stage_events", None)

        def mark():
            if ev is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream(self.device))
                ev.append(e)
        mark()
        h, mu_x, logw, c = eng.encoder(x, x_lengths, lang, tone, word_pos, syllable_pos, spk_embed)
        w_ceil, y_lengths, attn, mu_y = eng.length_regulate(logw, x_lengths, mu_x, length_scale)
        mark()
        encoder_outputs = mu_y
        if B != 1 and not batched:
            raise ValueError(f"synthesise() requires batch_size=1, got batch_size={B}. Please pass one sample at a time.")
        mel_len1 = 0
        if prompt_feat is not None and prompt_h is not None:
            # voice-cloning glue (jyutvoice_tts.py:213-225): prompt frames are prepended to mu / cond
            prompt_h = prompt_h.to(self.device, torch.float32)
            prompt_feat = prompt_feat.to(self.device, torch.float32)
            mu_y = torch.cat([prompt_h.transpose(1, 2), mu_y], dim=2)
            mel_len1 = prompt_feat.shape[1]
            conds = torch.zeros(B, mu_y.shape[2], self.output_size, device=self.device)
            conds[:, :mel_len1] = prompt_feat
            conds = conds.transpose(1, 2).contiguous()
            lens = (y_lengths + mel_len1) if B > 1 else torch.full((1,), mu_y.shape[2], dtype=torch.int64, device=self.device)
        else:
            conds = torch.zeros_like(mu_y)
            lens = y_lengths
        T = mu_y.shape[2]
        eng = rt.ensure(B, T, Tt)
        t_span = 1 - torch.cos(torch.linspace(0, 1, n_timesteps + 1) * 0.5 * torch.pi)   # flow_matching.py:387-389
        dec = eng.cfm_solve(mu_y.contiguous(), lens if B > 1 else None, c, conds, n_timesteps, temperature, t_span=t_span)
        dec = dec[:, :, mel_len1:]
        mark()
        torch.cuda.synchronize(self.device)      # the reference's rtf omits this and is meaningless on a GPU
        t = (dt.datetime.now() - t0).total_seconds()
        rtf = t * spec.SAMPLE_RATE / (dec.shape[-1] * spec.HOP_LENGTH * max(B, 1))
        return {"encoder_outputs": encoder_outputs, "decoder_outputs": dec, "attn": attn.unsqueeze(1), "mel": dec,
                "mel_lengths": y_lengths, "rtf": rtf}
