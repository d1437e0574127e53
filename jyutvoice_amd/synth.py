"""Synthetic weights and inputs (there are no checkpoints or datasets on the build/GPU boxes).

Weights are *key-hashed*: every tensor is drawn from a generator seeded with crc32(key), so the
same values can be produced anywhere (the survey container applied the same recipe to the
reference's own modules to make tests/golden/*.npz) without shipping a checkpoint.

Recipe, chosen so a 70-block random network stays numerically tame:
  matrices / conv kernels   N(0,1) * fan_in^-1/2
  biases, LayerNorm betas   N(0,1) * 0.02
  LayerNorm gains           1 + 0.1 N(0,1)
  Snake alpha               1 + 0.1 |N(0,1)|
  weight-norm g             ||v||_row * (1 + 0.1 N(0,1))      (so the fold w = g v/||v|| is exercised)
  f0 classifier             weight * 60, bias 80  -> f0 mostly 20..250 Hz with a few unvoiced frames
  conv_post                 weight * 0.3                       (keeps exp(.) in iSTFT range)
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Optional

import torch

from . import spec


def _gen(key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode("utf-8")))
    return g


def _randn(key: str, shape) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=_gen(key), dtype=torch.float32)


_GAIN_SUFFIX = (".gamma", "block.2.weight", "norm1.weight", "norm3.weight",
                "norm_mha.weight", "norm_ff.weight", "after_norm.weight", "embed.out.1.weight")   # prompt encoder LayerNorms
_WN_G = (".parametrizations.weight.original0", ".weight_g")
_WN_V = (".parametrizations.weight.original1", ".weight_v")


def _fan_in(key: str, shape) -> float:
    if key.startswith("ups."):
        i = int(key.split(".")[1])
        return shape[0] * shape[2] / spec.HIFT_UP_RATES[i]
    if "emb" in key.split(".")[-2] or key.endswith("syllable_pos.weight"):
        return float(shape[1])
    f = 1
    for s in shape[1:]:
        f *= s
    return float(f)


def _one(key: str, shape, inv) -> torch.Tensor:
    if key.endswith(".alpha"):
        return 1.0 + 0.1 * _randn(key, shape).abs()
    if key.endswith(_GAIN_SUFFIX):
        return 1.0 + 0.1 * _randn(key, shape)
    if key.endswith(_WN_G):
        for gs, vs in zip(_WN_G, _WN_V):
            if key.endswith(gs):
                vkey = key[: -len(gs)] + vs
        v = _one(vkey, inv[vkey], inv)
        norm = v.flatten(1).norm(dim=1).view(shape)
        return norm * (1.0 + 0.1 * _randn(key, shape))
    if key.endswith((".bias", ".beta")):
        b = 0.02 * _randn(key, shape)
        if key == "f0_predictor.classifier.bias":
            b = b + 80.0
        return b
    w = _randn(key, shape) * (_fan_in(key, shape) ** -0.5)
    if key == "f0_predictor.classifier.weight":
        w = w * 60.0
    if key.startswith("conv_post."):
        w = w * 0.3
    return w


def synth_state_dict(inventory, prefix: str = "") -> Dict[str, torch.Tensor]:
    """Synthetic fp32 CPU state-dict for `inventory` (spec.TTS_INVENTORY or spec.HIFT_INVENTORY).

    `prefix` filters keys (e.g. "decoder.estimator.")."""
    out = {}
    for key, shape in inventory.items():
        if key.startswith(prefix):
            out[key] = _one(key, shape, inventory).contiguous()
    return out


def tts_state_dict(fixed_duration: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """Synthetic JyutVoiceTTS state-dict.  `fixed_duration` (e.g. 1.5) forces every token to
    ceil(fixed_duration) frames by zeroing dp.proj.weight (fixed-length batches)."""
    sd = synth_state_dict(spec.TTS_INVENTORY)
    if fixed_duration is not None:
        sd["dp.proj.weight"] = torch.zeros_like(sd["dp.proj.weight"])
        sd["dp.proj.bias"] = torch.full_like(sd["dp.proj.bias"], math.log(fixed_duration))
    return sd


def hift_state_dict() -> Dict[str, torch.Tensor]:
    return synth_state_dict(spec.HIFT_INVENTORY)


def prompt_state_dict() -> Dict[str, torch.Tensor]:
    """Synthetic FlowEncoder (prompt branch) state-dict: infer.py:35-83 key names."""
    return synth_state_dict(spec.PROMPT_INVENTORY)


def prompt_tokens(n: int, n_tokens: int, lengths=None, first_index: int = 0):
    """`n` synthetic speech-token sequences (ids < 6561, what the speech tokenizer of infer.py:85-163 emits), padded with 0."""
    tok = torch.zeros(n, n_tokens, dtype=torch.int64)
    lens = torch.tensor(lengths if lengths is not None else [n_tokens] * n, dtype=torch.int64)
    for b in range(n):
        g = torch.Generator(device="cpu")
        g.manual_seed(4321 + first_index + b)
        L = int(lens[b])
        tok[b, :L] = torch.randint(0, spec.PROMPT_VOCAB, (L,), generator=g)
    return tok, lens


def rand_noise() -> torch.Tensor:
    """The CFM's fixed noise tensor: torch.manual_seed(0); torch.randn([1,80,15000]) on the CPU
    generator (jyutvoice/flow/flow_matching.py:353-354).  A private generator with seed 0 yields
    the same stream as the global one without touching global RNG state (unlike the reference)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(0)
    return torch.randn([1, spec.N_FEATS, spec.NOISE_FRAMES], generator=g, dtype=torch.float32)


def utterance(index: int, n_tokens: int):
    """One synthetic utterance: five equal-length id lists with blanks interspersed and pad 0 at
    both ends (the contract of jyutvoice/text/__init__.py:20-35 + utils/utils.py:131-135), and a
    N(0,1) speaker embedding."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1234 + index)
    n_real = (n_tokens - 1) // 2
    def ids(hi):
        real = torch.randint(1, hi, (n_real,), generator=g)
        out = torch.zeros(n_tokens, dtype=torch.int64)
        out[1 : 2 * n_real : 2] = real
        return out
    phone = ids(spec.ENC_N_VOCAB)
    lang = ids(spec.ENC_N_LANG)
    tone = ids(spec.ENC_N_TONE)
    word_pos = ids(spec.ENC_N_WORD_POS)
    syl_pos = ids(spec.ENC_N_SYL_POS)
    spk = torch.randn(spec.SPK_EMBED_DIM, generator=g, dtype=torch.float32)
    return phone, lang, tone, word_pos, syl_pos, spk


def batch(n_utts: int, n_tokens: int, first_index: int = 0, lengths=None):
    """Padded batch of synthetic utterances -> dict of tensors shaped like synthesise()'s inputs."""
    cols = [utterance(first_index + i, n_tokens) for i in range(n_utts)]
    x = torch.stack([c[0] for c in cols])
    out = {
        "x": x,
        "lang": torch.stack([c[1] for c in cols]),
        "tone": torch.stack([c[2] for c in cols]),
        "word_pos": torch.stack([c[3] for c in cols]),
        "syllable_pos": torch.stack([c[4] for c in cols]),
        "spk_embed": torch.stack([c[5] for c in cols]),
    }
    if lengths is None:
        lengths = [n_tokens] * n_utts
    xl = torch.tensor(lengths, dtype=torch.int64)
    for k in ("x", "lang", "tone", "word_pos", "syllable_pos"):
        pad = torch.arange(n_tokens).unsqueeze(0) >= xl.unsqueeze(1)
        out[k] = out[k].masked_fill(pad, 0)
    out["x_lengths"] = xl
    return out


# ---- "hostile" checkpoints: what training does to weights and the tame recipe above does not -------------------------------
def _outlier_norm_channels(sd, blk: str, norm: str, consumers, channels, factor: float) -> None:
    """LayerNorm `blk.norm` gets gain and offset x factor on `channels`, and the Linears that read it get those input columns
    / factor: the same function in exact arithmetic (what a trained network with a few huge-gain channels looks like), but the
    load-time bound sqrt(255) max|g| + max|b| is `factor` times larger while every other channel stays where it was"""
    g, b = sd[blk + norm + ".weight"].clone(), sd[blk + norm + ".bias"].clone()
    g[channels] *= factor
    b[channels] *= factor
    sd[blk + norm + ".weight"], sd[blk + norm + ".bias"] = g, b
    for name in consumers:
        w = sd[blk + name].clone()
        w[:, channels] /= factor
        sd[blk + name] = w


def hostile_tts_state_dict(fixed_duration: Optional[float] = None, unusable_bound: bool = False) -> Dict[str, torch.Tensor]:
    """The synthetic TTS checkpoint with the features of a TRAINED one that stress the fp16x3 engine's load-time bounds
    (registry.hip: sqrt(255) max|g| + max|b| behind a LayerNorm, row-L1 norms behind a Linear), applied to the estimator:
      * LayerNorm gains x2.5 on five transformer blocks of the down / mid / up stages (norm1 and norm3, not compensated:
        sharper attention, larger feed-forward activations.  x8 -- tried first -- makes THIS randomly weighted network
        chaotic: the reference's own fp32 arithmetic then differs from the same model evaluated in fp64 by 1.4 max-abs on the
        mel, so no implementation can be compared with it; x2.5 leaves that gap at ~1e-5, tests/test_gpu_hostile.py records it);
      * outlier LayerNorm channels x7.3 (three channels of norm1 / norm3 in six blocks), compensated in the Linears that read
        them: the bound is 7.3x the ordinary channels' range;
      * two outlier output rows x30 in to_q / to_k / ff.net.0 of three blocks (a few channels carry the row-L1 bound, the rest
        sit 30x below it: the power-of-two scale is set by the outliers and pushes ordinary operands towards the floor);
      * one near-zero-variance channel: a LayerNorm gain of 1e-6 with an offset of 0.5 (the normalised channel is a constant)
        in one norm1 and one resnet LayerNorm;
      * one ff.net.2 whose weights are 1/64 of the recipe's (a layer that training has nearly switched off).
    unusable_bound=True adds one block (mid_blocks.4.1.2) whose norm1 has a channel gain of 1e31, compensated the same way:
    its bound is beyond what registry.hip accepts (>= 1e30), so q | k | v and the attention of THAT block must stay on
    bf16x6 while every other layer keeps fp16x3 (jv_flow_contraction_info)."""
    sd = tts_state_dict(fixed_duration)
    p = "decoder.estimator."
    for blk in ("down_blocks.0.1.1", "mid_blocks.3.1.0", "mid_blocks.3.1.1", "mid_blocks.7.1.2", "up_blocks.0.1.3"):
        sd[p + blk + ".norm1.weight"] = sd[p + blk + ".norm1.weight"] * 2.5
        sd[p + blk + ".norm3.weight"] = sd[p + blk + ".norm3.weight"] * 2.5
    qkv = ("attn1.to_q.weight", "attn1.to_k.weight", "attn1.to_v.weight")
    for blk in ("down_blocks.0.1.2", "mid_blocks.0.1.0", "mid_blocks.3.1.1", "mid_blocks.8.1.3", "mid_blocks.11.1.0", "up_blocks.0.1.0"):
        _outlier_norm_channels(sd, p + blk + ".", "norm1", qkv, [3, 100, 200], 7.3)
        _outlier_norm_channels(sd, p + blk + ".", "norm3", ("ff.net.0.proj.weight",), [9, 64, 255], 7.3)
    for blk in ("down_blocks.0.1.0", "mid_blocks.5.1.2", "mid_blocks.11.1.3"):
        for name in ("attn1.to_q.weight", "attn1.to_k.weight", "ff.net.0.proj.weight"):
            w = sd[p + blk + "." + name].clone()
            w[5] *= 30.0
            w[77] *= 30.0
            sd[p + blk + "." + name] = w
    for key in ("mid_blocks.1.1.1.norm1", "mid_blocks.9.0.block1.block.2"):
        g, b = sd[p + key + ".weight"].clone(), sd[p + key + ".bias"].clone()
        g[100] = 1e-6
        b[100] = 0.5
        sd[p + key + ".weight"], sd[p + key + ".bias"] = g, b
    sd[p + "mid_blocks.6.1.1.ff.net.2.weight"] = sd[p + "mid_blocks.6.1.1.ff.net.2.weight"] / 64.0
    if unusable_bound:
        _outlier_norm_channels(sd, p + "mid_blocks.4.1.2.", "norm1", qkv, [7], 1e31)
    return sd


def hostile_hift_state_dict(level: float = 1.0) -> Dict[str, torch.Tensor]:
    """The synthetic HiFT checkpoint with Snake alphas down to 0.05 on some channels of every ResBlock (1 / alpha is what
    the measured fp16x3 bound adds for the Snake prologue: hift.hip `a_extra`), 0.25 on others and one up to 8."""
    sd = hift_state_dict()
    lo = 1.0 - 0.95 * level
    for key in list(sd):
        if key.endswith(".alpha") and (key.startswith("resblocks.") or key.startswith("source_resblocks.")):
            a = sd[key].clone()
            n = a.numel()
            flat = a.view(-1)
            flat[3 % n] = lo
            flat[(n // 2 + 1) % n] = 0.25 + 0.75 * (1.0 - level)
            flat[n - 1] = 1.0 + 7.0 * level
            sd[key] = a
    return sd
