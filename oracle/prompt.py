"""Oracle (test infrastructure): the prompt branch's FlowEncoder -- speech tokens -> prompt_h [B, 2*Tk, 80].

Follows infer.py:35-83 (FlowEncoder: Embedding -> UpsampleConformerEncoder -> Linear(512, 80)) and
jyutvoice/transformer/upsample_encoder.py:329-375 (forward), :20-61 (Upsample1D), :64-134 (PreLookaheadLayer),
jyutvoice/transformer/subsampling.py:84-115 (LinearNoSubsampling), jyutvoice/transformer/embedding.py:201-296
(EspnetRelPositionalEncoding), jyutvoice/transformer/encoder_layer.py:241-319 (ConformerEncoderLayer, pre-LN, no conv
module, no macaron), jyutvoice/transformer/attention.py:204-334 (RelPositionMultiHeadedAttention, rel_shift),
:86-124 (masked softmax), jyutvoice/transformer/positionwise_feed_forward.py:47-55 (SiLU FFN).

Batches follow the reference's only usage, B = 1 (infer.py:210-262): utterance b of a padded batch is computed as if
alone, i.e. frames at and beyond its length are zero for the look-ahead / upsampling convolutions and masked as keys.
(The reference's own padded-batch result differs from its B = 1 result near the end of the shorter rows, because
LinearNoSubsampling does not re-mask: subsampling.py:113-115.)
"""
import math

import torch
import torch.nn.functional as F

HEADS = 8


def rel_pos_emb(T, d=512):
    """[1, 2T-1, d]: relative positions T-1 ... -(T-1)  (embedding.py:224-254, :290-296 with offset 0)"""
    pos = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pp, pn = torch.zeros(T, d), torch.zeros(T, d)
    pp[:, 0::2], pp[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
    pn[:, 0::2], pn[:, 1::2] = torch.sin(-1 * pos * div), torch.cos(-1 * pos * div)
    return torch.cat([torch.flip(pp, [0]).unsqueeze(0), pn[1:].unsqueeze(0)], dim=1)


def rel_shift(x):
    """[B,H,T,2T-1] -> [B,H,T,T]: out[i, j] = x[i, j - i + T - 1]  (attention.py:226-246)"""
    b, h, t, n = x.shape
    xp = torch.cat([torch.zeros(b, h, t, 1, dtype=x.dtype), x], dim=-1).view(b, h, n + 1, t)
    return xp[:, :, 1:].view_as(x)[:, :, :, : n // 2 + 1]


def rel_attention(sd, pre, x, pos_emb, key_mask):
    """x [B,T,512], pos_emb [1,2T-1,512], key_mask [B,1,T] bool -> [B,T,512]"""
    B, T, D = x.shape
    dk = D // HEADS
    lin = lambda n, z: F.linear(z, sd[pre + f"linear_{n}.weight"], sd.get(pre + f"linear_{n}.bias"))
    q = lin("q", x).view(B, T, HEADS, dk)
    k = lin("k", x).view(B, T, HEADS, dk).transpose(1, 2)
    v = lin("v", x).view(B, T, HEADS, dk).transpose(1, 2)
    p = F.linear(pos_emb, sd[pre + "linear_pos.weight"]).view(1, -1, HEADS, dk).transpose(1, 2)
    qu = (q + sd[pre + "pos_bias_u"]).transpose(1, 2)
    qv = (q + sd[pre + "pos_bias_v"]).transpose(1, 2)
    ac = qu @ k.transpose(-2, -1)
    bd = rel_shift(qv @ p.transpose(-2, -1))
    scores = (ac + bd) / math.sqrt(dk)
    dead = ~key_mask.unsqueeze(1)                                   # [B,1,1,T]
    attn = torch.softmax(scores.masked_fill(dead, -float("inf")), dim=-1).masked_fill(dead, 0.0)
    o = (attn @ v).transpose(1, 2).reshape(B, T, D)
    return lin("out", o)


def conformer_block(sd, pre, x, pos_emb, key_mask):
    ln = lambda n, z: F.layer_norm(z, (z.shape[-1],), sd[pre + n + ".weight"], sd[pre + n + ".bias"], 1e-5)
    x = x + rel_attention(sd, pre + "self_attn.", ln("norm_mha", x), pos_emb, key_mask)
    h = F.silu(F.linear(ln("norm_ff", x), sd[pre + "feed_forward.w_1.weight"], sd[pre + "feed_forward.w_1.bias"]))
    return x + F.linear(h, sd[pre + "feed_forward.w_2.weight"], sd[pre + "feed_forward.w_2.bias"])


def embed(sd, pre, x):
    """Linear -> LayerNorm -> x * sqrt(512), and the relative positional table"""
    x = F.linear(x, sd[pre + "out.0.weight"], sd[pre + "out.0.bias"])
    x = F.layer_norm(x, (x.shape[-1],), sd[pre + "out.1.weight"], sd[pre + "out.1.bias"], 1e-5)
    return x * math.sqrt(x.shape[-1]), rel_pos_emb(x.shape[1], x.shape[-1])


def pre_lookahead(sd, pre, x):
    """x [B,T,512]: conv k4 over t..t+3 (zeros beyond the end) -> LeakyReLU(0.01) -> causal conv k3 -> + x"""
    h = F.pad(x.transpose(1, 2), (0, 3))
    h = F.leaky_relu(F.conv1d(h, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"]))
    h = F.conv1d(F.pad(h, (2, 0)), sd[pre + "conv2.weight"], sd[pre + "conv2.bias"])
    return h.transpose(1, 2) + x


def upsample(sd, pre, x):
    """nearest x2, 4 zeros in front, conv k5"""
    h = F.interpolate(x.transpose(1, 2), scale_factor=2.0, mode="nearest")
    return F.conv1d(F.pad(h, (4, 0)), sd[pre + "conv.weight"], sd[pre + "conv.bias"]).transpose(1, 2)


def _encode_one(sd, token, taps=None):
    """token [1,Tk] -> [1, 2Tk, 80]  (one utterance, full key mask)"""
    x = F.embedding(torch.clamp(token, min=0), sd["input_embedding.weight"])
    pre = "encoder."
    T = x.shape[1]
    ones = torch.ones(1, 1, T, dtype=torch.bool)
    x, pos = embed(sd, pre + "embed.", x)
    x = pre_lookahead(sd, pre + "pre_lookahead_layer.", x)
    if taps is not None:
        taps["lookahead"] = x
    for i in range(6):
        x = conformer_block(sd, pre + f"encoders.{i}.", x, pos, ones)
        if taps is not None and i == 0:
            taps["block0"] = x
    x = upsample(sd, pre + "up_layer.", x)
    if taps is not None:
        taps["up"] = x
    ones2 = torch.ones(1, 1, 2 * T, dtype=torch.bool)
    x, pos = embed(sd, pre + "up_embed.", x)
    for i in range(4):
        x = conformer_block(sd, pre + f"up_encoders.{i}.", x, pos, ones2)
    x = F.layer_norm(x, (x.shape[-1],), sd[pre + "after_norm.weight"], sd[pre + "after_norm.bias"], 1e-5)
    return F.linear(x, sd["encoder_proj.weight"], sd["encoder_proj.bias"])


def flow_encoder(sd, token, token_len, taps=None):
    """token [B,Tk] int64 (padded), token_len [B] -> h [B, 2*Tk, 80] (zero beyond 2*len), h_lengths [B]"""
    B, Tk = token.shape
    h = torch.zeros(B, 2 * Tk, sd["encoder_proj.weight"].shape[0])
    for b in range(B):
        L = int(token_len[b])
        if L > 0:
            h[b, : 2 * L] = _encode_one(sd, token[b : b + 1, :L], taps if b == 0 else None)[0]
    return h, token_len * 2
