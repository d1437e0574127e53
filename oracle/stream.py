"""Oracle (test infrastructure): the streaming helpers around the hot path.

fade_in_out follows jyutvoice/utils/common.py:181-191; hift_inference follows jyutvoice/hifigan/generator.py:450-466
(the `cache_source` continuation: the head of the freshly generated source signal is overwritten with the tail kept from
the previous chunk before decoding)."""
import torch

from . import hift as ohift


def fade_in_out(fade_in, fade_out, window):
    n = int(window.shape[0] / 2)
    out = fade_in.clone()
    out[..., :n] = fade_in[..., :n] * window[:n] + fade_out[..., -n:] * window[n:]
    return out


def hift_inference(w, mel, s_fresh, cache_source=None):
    """mel [B,80,T], s_fresh [B,1,480T] (what the source module produced for this call), cache_source [B,1,n] or None
    -> (wav [B,480T], s as decoded)"""
    s = s_fresh.clone()
    if cache_source is not None and cache_source.shape[2] != 0:
        s[:, :, : cache_source.shape[2]] = cache_source
    return ohift.decode(w, mel, s), s
