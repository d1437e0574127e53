"""Oracle (test infrastructure): CosyVoice2 causal flow estimator and the CFM Euler/CFG solver.

Follows jyutvoice/flow/decoder.py:15-30 (sinusoidal embedding), :159-171 (time MLP), :767-770
(causal conv), :784-788 (conv+LayerNorm+Mish block), :110-115 (resnet), :917-1018 (forward);
jyutvoice/flow/transformer.py:355-443 (pre-LN self-attention + GELU FF); jyutvoice/utils/common.py:201-209
(0 / -1e10 additive key bias); jyutvoice/flow/flow_matching.py:215-265, 356-401 (solver).

The attention/GELU arithmetic itself is diffusers==0.35.2 (not vendored by the reference, absent
here): restated from its published AttnProcessor2_0/GELU semantics -- see oracle/__init__.py.
"""
import math

import torch
import torch.nn.functional as F

HEADS = 8


def time_embedding(sd, t, pre, dim=320, scale=1000.0):
    half = dim // 2
    f = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
    e = scale * t.unsqueeze(1) * f.unsqueeze(0)
    e = torch.cat([e.sin(), e.cos()], dim=-1)
    e = e.to(sd[pre + "time_mlp.linear_1.weight"].dtype)      # (no-op in fp32; lets tests evaluate the same model in fp64)
    e = F.linear(e, sd[pre + "time_mlp.linear_1.weight"], sd[pre + "time_mlp.linear_1.bias"])
    e = F.silu(e)
    return F.linear(e, sd[pre + "time_mlp.linear_2.weight"], sd[pre + "time_mlp.linear_2.bias"])


def causal_conv(x, w, b):
    return F.conv1d(F.pad(x, (w.shape[2] - 1, 0)), w, b)


def causal_block(sd, pre, x, mask):
    h = causal_conv(x * mask, sd[pre + "block.0.weight"], sd[pre + "block.0.bias"])
    h = F.layer_norm(h.transpose(1, 2), (h.shape[1],), sd[pre + "block.2.weight"], sd[pre + "block.2.bias"], 1e-5)
    return F.mish(h.transpose(1, 2)) * mask


def resnet(sd, pre, x, mask, temb):
    h = causal_block(sd, pre + "block1.", x, mask)
    h = h + F.linear(F.mish(temb), sd[pre + "mlp.1.weight"], sd[pre + "mlp.1.bias"]).unsqueeze(-1)
    h = causal_block(sd, pre + "block2.", h, mask)
    return h + F.conv1d(x * mask, sd[pre + "res_conv.weight"], sd[pre + "res_conv.bias"])


def transformer_block(sd, pre, h, bias):
    """h [B,T,256], bias [B,1,1,T] additive (0 / -1e10 on padded keys)."""
    B, T, C = h.shape
    n = F.layer_norm(h, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
    q = F.linear(n, sd[pre + "attn1.to_q.weight"])
    k = F.linear(n, sd[pre + "attn1.to_k.weight"])
    v = F.linear(n, sd[pre + "attn1.to_v.weight"])
    hd = q.shape[-1] // HEADS
    sp = lambda z: z.view(B, T, HEADS, hd).transpose(1, 2)
    s = sp(q) @ sp(k).transpose(-2, -1) / math.sqrt(hd) + bias
    o = torch.softmax(s, dim=-1) @ sp(v)
    o = o.transpose(1, 2).reshape(B, T, HEADS * hd)
    h = h + F.linear(o, sd[pre + "attn1.to_out.0.weight"], sd[pre + "attn1.to_out.0.bias"])
    n = F.layer_norm(h, (C,), sd[pre + "norm3.weight"], sd[pre + "norm3.bias"], 1e-5)
    f = F.gelu(F.linear(n, sd[pre + "ff.net.0.proj.weight"], sd[pre + "ff.net.0.proj.bias"]))
    return h + F.linear(f, sd[pre + "ff.net.2.weight"], sd[pre + "ff.net.2.bias"])


def estimator(sd, x, mask, mu, t, spks, cond, pre="decoder.estimator.", taps=None, streaming=False, chunk=50):
    """x,mu,cond [B,80,T], mask [B,1,T], t [B], spks [B,80] -> [B,80,T].  decoder.py:917-1018.

    `taps` (optional dict) receives intermediate activations for the golden fixtures.
    streaming=True: chunk-causal attention (decoder.py:951-954 -> utils/mask.py:91-126,192-198 with
    static_chunk_size = chunk, all left chunks): query i sees keys j < (i // chunk + 1) * chunk, and the key mask."""
    temb = time_embedding(sd, t, pre)
    T = x.shape[2]
    h = torch.cat([x, mu, spks.unsqueeze(-1).expand(-1, -1, T), cond], dim=1)
    allowed = mask.bool().unsqueeze(1)                    # [B,1,1,T]: every query row sees the key mask
    if streaming:
        i = torch.arange(T)
        allowed = allowed & (i[None, :] < ((i // chunk + 1) * chunk)[:, None])[None, None]   # [B,1,T,T]
    bias = (1.0 - allowed.to(x.dtype)) * -1.0e10

    def stage(prefix, h, n_blocks=4):
        h = resnet(sd, prefix + "0.", h, mask, temb)
        if taps is not None:
            taps[prefix + "resnet"] = h
        g = h.transpose(1, 2).contiguous()
        for j in range(n_blocks):
            g = transformer_block(sd, prefix + f"1.{j}.", g, bias)
        return g.transpose(1, 2).contiguous()

    h = stage(pre + "down_blocks.0.", h)
    skip = h
    if taps is not None:
        taps["down"] = h
    h = causal_conv(h * mask, sd[pre + "down_blocks.0.2.weight"], sd[pre + "down_blocks.0.2.bias"])
    n_mid = 1 + max(int(k[len(pre):].split(".")[1]) for k in sd if k.startswith(pre + "mid_blocks."))
    for i in range(n_mid):
        h = stage(pre + f"mid_blocks.{i}.", h)
        if taps is not None and i in (0, n_mid - 1):
            taps[f"mid{i}"] = h
    h = stage(pre + "up_blocks.0.", torch.cat([h, skip], dim=1))
    if taps is not None:
        taps["up"] = h
    h = causal_conv(h * mask, sd[pre + "up_blocks.0.2.weight"], sd[pre + "up_blocks.0.2.bias"])
    h = causal_block(sd, pre + "final_block.", h, mask)
    out = F.conv1d(h * mask, sd[pre + "final_proj.weight"], sd[pre + "final_proj.bias"])
    return out * mask


def t_span(n_timesteps):
    return 1 - torch.cos(torch.linspace(0, 1, n_timesteps + 1) * 0.5 * torch.pi)


def cfm_solve(sd, noise, mu, mask, spks, cond, n_timesteps, temperature=1.0, cfg_rate=0.7,
              pre="decoder.estimator.", est=None):
    """Euler solve with classifier-free guidance.  flow_matching.py:215-265, 385-389.

    Works for any batch B (the reference is B=1: rows [cond; uncond] of a 2-batch); batch b's
    conditional row and its unconditional row (mu=spks=cond=0) are evaluated together, which is the
    per-utterance loop the reference would run."""
    est = est or estimator
    B, _, T = mu.shape
    x = noise[:, :, :T].expand(B, -1, -1) * temperature
    ts = t_span(n_timesteps)
    t, dt = ts[0], ts[1] - ts[0]
    zeros = torch.zeros_like(mu)
    for step in range(1, n_timesteps + 1):
        x_in = torch.cat([x, x], 0)
        d = est(sd, x_in, torch.cat([mask, mask], 0), torch.cat([mu, zeros], 0),
                t.expand(2 * B), torch.cat([spks, torch.zeros_like(spks)], 0),
                torch.cat([cond, zeros], 0), pre=pre)
        d = (1.0 + cfg_rate) * d[:B] - cfg_rate * d[B:]
        x = x + dt * d
        t = t + dt
        if step < n_timesteps:
            dt = ts[step + 1] - t
    return x.float()
