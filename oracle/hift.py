"""Oracle (test infrastructure): HiFT vocoder (NSF source + ISTFTNet trunk).

Follows jyutvoice/hifigan/f0_predictor.py:52-55; jyutvoice/hifigan/generator.py:141-176 (sine
generator), :220-236 (source module), :371-394 (STFT/iSTFT), :90-97 (ResBlock), :396-432 (decode),
:450-466 (inference); jyutvoice/transformer/activation.py:73-84 (Snake).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import get_window

from jyutvoice_amd import spec  # constants only (shapes / hyper-parameters)


def fold_weight_norm(sd):
    """state-dict -> plain {name.weight} dict with both weight-norm spellings folded (w = g v/||v||,
    norm over all dims but 0: torch._weight_norm)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".parametrizations.weight.original1"):
            base = k[: -len("parametrizations.weight.original1")]
            g = sd[base + "parametrizations.weight.original0"]
        elif k.endswith(".weight_v"):
            base = k[: -len("weight_v")]
            g = sd[base + "weight_g"]
        elif k.endswith((".parametrizations.weight.original0", ".weight_g")):
            continue
        else:
            out[k] = v
            continue
        n = v.flatten(1).norm(dim=1).view(g.shape)
        out[base + "weight"] = v * (g / n)
    return out


def snake(x, alpha):
    a = alpha.view(1, -1, 1)
    return x + (1.0 / (a + 1e-9)) * torch.sin(x * a) ** 2


def f0_predict(w, mel, pre="f0_predictor."):
    h = mel
    for n in range(5):
        h = F.elu(F.conv1d(h, w[pre + f"condnet.{2 * n}.weight"], w[pre + f"condnet.{2 * n}.bias"], padding=1))
    return torch.abs(F.linear(h.transpose(1, 2), w[pre + "classifier.weight"], w[pre + "classifier.bias"]).squeeze(-1))


def source(w, f0, phase_vec, noise, sr=spec.SAMPLE_RATE):
    """f0 [B,T] -> s [B,1,480T].  phase_vec [B,9,1] (harmonic 0 must be 0) and noise [B,9,480T]
    replace the reference's Uniform(-pi,pi) / randn draws (generator.py:155-158,171)."""
    f0u = f0[:, None].repeat_interleave(spec.HIFT_UPSAMPLE_TOTAL, dim=2)        # nearest upsample
    nh = spec.HIFT_NB_HARMONICS + 1
    Fm = torch.cat([f0u * (i + 1) / sr for i in range(nh)], dim=1)               # [B,9,L]
    theta = 2 * np.pi * (torch.cumsum(Fm, dim=-1) % 1)
    sines = spec.HIFT_NSF_ALPHA * torch.sin(theta + phase_vec)
    uv = (f0u > spec.HIFT_VOICED_THRESHOLD).float()
    namp = uv * spec.HIFT_NSF_SIGMA + (1 - uv) * spec.HIFT_NSF_ALPHA / 3
    sines = sines * uv + namp * noise
    s = torch.tanh(F.linear(sines.transpose(1, 2), w["m_source.l_linear.weight"], w["m_source.l_linear.bias"]))
    return s.transpose(1, 2)


def _window():
    return torch.from_numpy(get_window("hann", spec.HIFT_NFFT, fftbins=True).astype(np.float32))


def stft(s):
    """s [B,L] -> [B,18,L/4+1] = cat(real, imag).  generator.py:371-381,399-400"""
    sp = torch.stft(s, spec.HIFT_NFFT, spec.HIFT_HOP, spec.HIFT_NFFT, window=_window(), return_complex=True)
    sp = torch.view_as_real(sp)
    return torch.cat([sp[..., 0], sp[..., 1]], dim=1)


def resblock(w, pre, x, k):
    for j, d in enumerate(spec.HIFT_RB_DILATIONS):
        xt = snake(x, w[pre + f"activations1.{j}.alpha"])
        xt = F.conv1d(xt, w[pre + f"convs1.{j}.weight"], w[pre + f"convs1.{j}.bias"], dilation=d, padding=d * (k - 1) // 2)
        xt = snake(xt, w[pre + f"activations2.{j}.alpha"])
        xt = F.conv1d(xt, w[pre + f"convs2.{j}.weight"], w[pre + f"convs2.{j}.bias"], padding=(k - 1) // 2)
        x = xt + x
    return x


def decode(w, mel, s, taps=None):
    """mel [B,80,T], s [B,1,480T] -> wav [B,480T].  generator.py:396-432 (w = folded weights)."""
    s_stft = stft(s.squeeze(1))
    x = F.conv1d(mel, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(spec.HIFT_UP_RATES, spec.HIFT_UP_KERNELS)):
        x = F.leaky_relu(x, spec.HIFT_LRELU_SLOPE)
        x = F.conv_transpose1d(x, w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if i == 2:
            x = F.pad(x, (1, 0), mode="reflect")
        dk, ds, dp = spec.HIFT_SRC_DOWNS[i]
        si = F.conv1d(s_stft, w[f"source_downs.{i}.weight"], w[f"source_downs.{i}.bias"], stride=ds, padding=dp)
        si = resblock(w, f"source_resblocks.{i}.", si, spec.HIFT_SRC_RB_KERNELS[i])
        x = x + si
        xs = None
        for j, rk in enumerate(spec.HIFT_RB_KERNELS):
            r = resblock(w, f"resblocks.{3 * i + j}.", x, rk)
            xs = r if xs is None else xs + r
        x = xs / 3
        if taps is not None:
            taps[f"stage{i}"] = x
    x = F.leaky_relu(x)                                   # default slope 0.01 (generator.py:423)
    x = F.conv1d(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
    if taps is not None:
        taps["post"] = x
    nb = spec.HIFT_NFFT // 2 + 1
    mag = torch.clip(torch.exp(x[:, :nb]), max=1e2)
    ph = torch.sin(x[:, nb:])
    wav = torch.istft(torch.complex(mag * torch.cos(ph), mag * torch.sin(ph)), spec.HIFT_NFFT, spec.HIFT_HOP,
                      spec.HIFT_NFFT, window=_window())
    return torch.clamp(wav, -spec.HIFT_AUDIO_LIMIT, spec.HIFT_AUDIO_LIMIT)


def inference(sd, mel, phase_vec, noise, cache_source=None):
    """HiFTGenerator.inference with the random draws injected.  -> (wav, s)"""
    w = fold_weight_norm(sd)
    f0 = f0_predict(w, mel)
    s = source(w, f0, phase_vec, noise)
    if cache_source is not None and cache_source.shape[2] != 0:
        s[:, :, : cache_source.shape[2]] = cache_source
    return decode(w, mel, s), s
