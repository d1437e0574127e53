"""CPU: an independent pin for the one boundary of the oracle no reference code could pin (the attention
and GELU arithmetic of rows a6-a8 is diffusers==0.35.2, absent here and from /root/reference).

oracle.flow.transformer_block restates diffusers' Attention + AttnProcessor2_0 (q/k/v Linear without bias, 8 heads x 64,
softmax(q k^T / sqrt(64) + additive key bias) v, to_out Linear with bias) and its exact-erf GELU feed-forward as called
from jyutvoice/flow/transformer.py:355-443.  Here the same block is assembled from torch's OWN library modules --
torch.nn.MultiheadAttention (its in-projection, scaling, masking, softmax and out-projection code, none of it written
for this repo), nn.LayerNorm, nn.Linear, nn.GELU -- loaded with the same to_q / to_k / to_v / to_out.0 / norm / ff
weights, and the two must agree to fp32 rounding.  MultiheadAttention wants square q and output projections, so the
256-wide query / 256-wide output of the block (inner width 512) are embedded by zero padding, which changes no product.
"""
import torch
import torch.nn as nn

from oracle import flow as oflow

PRE = "decoder.estimator.mid_blocks.3.1.2."       # any of the 56 blocks: same shapes
HEADS, HEAD_DIM, DIM, INNER = 8, 64, 256, 512


def independent_block(sd, pre):
    mha = nn.MultiheadAttention(INNER, HEADS, bias=True, batch_first=True, kdim=DIM, vdim=DIM)
    with torch.no_grad():
        wq = torch.zeros(INNER, INNER)
        wq[:, :DIM] = sd[pre + "attn1.to_q.weight"]          # query zero-padded 256 -> 512: the extra columns meet zeros
        mha.q_proj_weight.copy_(wq)
        mha.k_proj_weight.copy_(sd[pre + "attn1.to_k.weight"])
        mha.v_proj_weight.copy_(sd[pre + "attn1.to_v.weight"])
        mha.in_proj_bias.zero_()                              # diffusers: bias=False on q, k, v
        wo = torch.zeros(INNER, INNER)
        wo[:DIM] = sd[pre + "attn1.to_out.0.weight"]          # outputs 256..511 are unused
        mha.out_proj.weight.copy_(wo)
        bo = torch.zeros(INNER)
        bo[:DIM] = sd[pre + "attn1.to_out.0.bias"]
        mha.out_proj.bias.copy_(bo)
    n1, n3 = nn.LayerNorm(DIM, eps=1e-5), nn.LayerNorm(DIM, eps=1e-5)
    ff1, ff2, gelu = nn.Linear(DIM, 1024), nn.Linear(1024, DIM), nn.GELU()      # nn.GELU() = exact erf form
    with torch.no_grad():
        n1.weight.copy_(sd[pre + "norm1.weight"]); n1.bias.copy_(sd[pre + "norm1.bias"])
        n3.weight.copy_(sd[pre + "norm3.weight"]); n3.bias.copy_(sd[pre + "norm3.bias"])
        ff1.weight.copy_(sd[pre + "ff.net.0.proj.weight"]); ff1.bias.copy_(sd[pre + "ff.net.0.proj.bias"])
        ff2.weight.copy_(sd[pre + "ff.net.2.weight"]); ff2.bias.copy_(sd[pre + "ff.net.2.bias"])
    mha.eval()

    def block(h, key_padding_mask=None, attn_mask=None):
        n = n1(h)
        q = torch.cat([n, torch.zeros_like(n)], dim=-1)
        a, _ = mha(q, n, n, key_padding_mask=key_padding_mask, attn_mask=attn_mask, need_weights=False)
        h = h + a[..., :DIM]
        return h + ff2(gelu(ff1(n3(h))))

    return block


def test_transformer_block_matches_torch_multiheadattention(tts_sd):
    g = torch.Generator().manual_seed(99)
    B, T = 3, 77
    lens = torch.tensor([77, 40, 1])
    h = torch.randn(B, T, DIM, generator=g) * 1.7
    valid = torch.arange(T)[None] < lens[:, None]                                  # [B, T]
    bias = ((1.0 - valid.float()) * -1.0e10)[:, None, None, :]                     # utils/common.py:201-209 as the oracle takes it
    with torch.inference_mode():
        want = independent_block(tts_sd, PRE)(h, ~valid)
        got = oflow.transformer_block(tts_sd, PRE, h, bias)
    err = float((got - want).abs().max())
    scale = float(want.abs().max())
    assert err <= 2e-5 * max(1.0, scale), (err, scale)
    # no mask at all: MultiheadAttention's unmasked path
    with torch.inference_mode():
        again = independent_block(tts_sd, PRE)(h[:1, :40])
        ref1 = oflow.transformer_block(tts_sd, PRE, h[:1, :40], torch.zeros(1, 1, 1, 40))
    assert float((again - ref1).abs().max()) <= 2e-5 * max(1.0, scale)


def test_streaming_mask_matches_torch_attn_mask(tts_sd):
    """the chunk-causal mask of oracle.flow.estimator (decoder.py:951-954, utils/mask.py:91-126) expressed as
    MultiheadAttention's boolean attn_mask gives the same block output"""
    g = torch.Generator().manual_seed(5)
    T, chunk = 120, 50
    h = torch.randn(1, T, DIM, generator=g)
    i = torch.arange(T)
    allowed = i[None, :] < ((i // chunk + 1) * chunk)[:, None]                     # [T, T], True = may attend
    bias = ((1.0 - allowed.float()) * -1.0e10)[None, None]
    with torch.inference_mode():
        got = oflow.transformer_block(tts_sd, PRE, h, bias)
        want = independent_block(tts_sd, PRE)(h, attn_mask=~allowed)
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
