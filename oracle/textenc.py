"""Oracle (test infrastructure): text encoder, duration predictor, length regulation.

Follows jyutvoice/models/text_encoder.py:11-29 (channel LayerNorm), :75-82 (prenet), :119-172 (RoPE),
:216-248 (attention), :276-281 (FFN), :326-337 (encoder stack), :406-451 (TextEncoder.forward);
jyutvoice/models/duration_predictor.py:48-60; jyutvoice/utils/model.py:7-46;
jyutvoice/models/jyutvoice_tts.py:184-203.
"""
import math

import torch
import torch.nn.functional as F


def channel_layernorm(x, gamma, beta, eps=1e-4):
    # text_encoder.py:20-29 -- over dim 1, biased variance
    mean = x.mean(1, keepdim=True)
    var = ((x - mean) ** 2).mean(1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def rope(x, d):
    """x [B,H,T,C]; rotate the first d dims with half-split pairing (i, i+d/2). text_encoder.py:119-172"""
    t = x.shape[2]
    theta = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
    ang = torch.einsum("n,d->nd", torch.arange(t).float(), theta)
    ang = torch.cat([ang, ang], dim=1)                    # [T, d]
    cos, sin = ang.cos()[None, None], ang.sin()[None, None]
    xr, xp = x[..., :d], x[..., d:]
    half = d // 2
    neg_half = torch.cat([-xr[..., half:], xr[..., :half]], dim=-1)
    return torch.cat([xr * cos + neg_half * sin, xp], dim=-1)


def mha(sd, pre, x, mask2d, n_heads=2):
    # text_encoder.py:216-248
    q = F.conv1d(x, sd[pre + "conv_q.weight"], sd[pre + "conv_q.bias"])
    k = F.conv1d(x, sd[pre + "conv_k.weight"], sd[pre + "conv_k.bias"])
    v = F.conv1d(x, sd[pre + "conv_v.weight"], sd[pre + "conv_v.bias"])
    b, d, t = q.shape
    kc = d // n_heads
    sp = lambda z: z.view(b, n_heads, kc, t).transpose(2, 3)   # b h t c
    q, k, v = sp(q), sp(k), sp(v)
    rd = int(kc * 0.5)
    q, k = rope(q, rd), rope(k, rd)
    scores = q @ k.transpose(-2, -1) / math.sqrt(kc)
    scores = scores.masked_fill(mask2d == 0, -1e4)
    p = torch.softmax(scores, dim=-1)
    o = (p @ v).transpose(2, 3).contiguous().view(b, d, t)
    return F.conv1d(o, sd[pre + "conv_o.weight"], sd[pre + "conv_o.bias"])


def text_encoder(sd, x, x_lengths, lang, tone, word_pos, syllable_pos, spk_embed, pre="encoder."):
    """-> (x [B,576,T], mu [B,80,T], x_mask [B,1,T]).  text_encoder.py:406-451"""
    ch = sd[pre + "emb.weight"].shape[1]
    h = (F.embedding(x, sd[pre + "emb.weight"]) + F.embedding(tone, sd[pre + "tone_emb.weight"])
         + F.embedding(word_pos, sd[pre + "word_pos_emb.weight"])
         + F.embedding(syllable_pos, sd[pre + "syllable_pos.weight"])) * math.sqrt(ch)
    h = h.transpose(1, 2)
    T = h.shape[2]
    x_mask = (torch.arange(T).unsqueeze(0) < x_lengths.unsqueeze(1)).unsqueeze(1).to(h.dtype)

    # prenet: text_encoder.py:75-82
    org = h
    for i in range(3):
        w = sd[pre + f"prenet.conv_layers.{i}.weight"]
        h = F.conv1d(h * x_mask, w, sd[pre + f"prenet.conv_layers.{i}.bias"], padding=w.shape[2] // 2)
        h = channel_layernorm(h, sd[pre + f"prenet.norm_layers.{i}.gamma"], sd[pre + f"prenet.norm_layers.{i}.beta"])
        h = torch.relu(h)
    h = (org + F.conv1d(h, sd[pre + "prenet.proj.weight"], sd[pre + "prenet.proj.bias"])) * x_mask

    B = h.shape[0]
    spk = spk_embed.unsqueeze(-1).expand(B, spk_embed.shape[1], T)
    le = F.embedding(lang, sd[pre + "lang_emb.weight"]).transpose(1, 2)
    h = torch.cat([h, spk, le], dim=1)

    # encoder stack: text_encoder.py:326-337
    mask2d = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    n_layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith(pre + "encoder.attn_layers."))
    for i in range(n_layers):
        e = pre + "encoder."
        h = h * x_mask
        y = mha(sd, e + f"attn_layers.{i}.", h, mask2d)
        h = channel_layernorm(h + y, sd[e + f"norm_layers_1.{i}.gamma"], sd[e + f"norm_layers_1.{i}.beta"])
        w1, w2 = sd[e + f"ffn_layers.{i}.conv_1.weight"], sd[e + f"ffn_layers.{i}.conv_2.weight"]
        y = F.conv1d(h * x_mask, w1, sd[e + f"ffn_layers.{i}.conv_1.bias"], padding=w1.shape[2] // 2)
        y = torch.relu(y)
        y = F.conv1d(y * x_mask, w2, sd[e + f"ffn_layers.{i}.conv_2.bias"], padding=w2.shape[2] // 2) * x_mask
        h = channel_layernorm(h + y, sd[e + f"norm_layers_2.{i}.gamma"], sd[e + f"norm_layers_2.{i}.beta"])
    h = h * x_mask
    mu = F.conv1d(h, sd[pre + "proj.weight"], sd[pre + "proj.bias"]) * x_mask
    return h, mu, x_mask


def duration_predictor(sd, x, x_mask, g, pre="dp."):
    """-> logw [B,1,T].  duration_predictor.py:48-60 (ReLU before LayerNorm)."""
    x = x + F.conv1d(g.unsqueeze(2), sd[pre + "cond.weight"], sd[pre + "cond.bias"])
    for n in ("1", "2"):
        w = sd[pre + f"conv_{n}.weight"]
        x = F.conv1d(x * x_mask, w, sd[pre + f"conv_{n}.bias"], padding=w.shape[2] // 2)
        x = torch.relu(x)
        x = channel_layernorm(x, sd[pre + f"norm_{n}.gamma"], sd[pre + f"norm_{n}.beta"])
    x = F.conv1d(x * x_mask, sd[pre + "proj.weight"], sd[pre + "proj.bias"])
    return x * x_mask


def monotonic_path(w_ceil, x_mask):
    """durations [B,1,Tx] (already ceil'd/scaled) -> (y_lengths [B], attn [B,Tx,Ty]).  utils/model.py:29-46"""
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    ty = int(y_lengths.max())
    y_mask = (torch.arange(ty).unsqueeze(0) < y_lengths.unsqueeze(1)).unsqueeze(1).to(x_mask.dtype)
    amask = (x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)).squeeze(1)       # [B,Tx,Ty]
    cum = torch.cumsum(w_ceil.squeeze(1), 1)                               # [B,Tx]
    lt = (torch.arange(ty).view(1, 1, ty).to(cum.dtype) < cum.unsqueeze(-1)).to(amask.dtype)
    path = lt - F.pad(lt, (0, 0, 1, 0))[:, :-1]
    return y_lengths, path * amask


def length_regulate(logw, x_mask, mu_x, length_scale=1.0):
    """-> (w_ceil [B,1,Tx], y_lengths [B] int64, attn [B,Tx,Ty], mu_y [B,80,Ty]).

    jyutvoice_tts.py:184-203.  Note w_ceil = ceil(w)*length_scale is not re-rounded, so the
    cumulative sums are fractional when length_scale != 1."""
    w = torch.exp(logw) * x_mask
    w_ceil = torch.ceil(w) * length_scale
    y_lengths, attn = monotonic_path(w_ceil, x_mask)
    mu_y = torch.matmul(attn.transpose(1, 2), mu_x.transpose(1, 2)).transpose(1, 2)
    return w_ceil, y_lengths, attn, mu_y
