"""The two helpers of `jyutvoice/utils/common.py` that streaming synthesis uses on the hot path's output.

`fade_in_out` (utils/common.py:181-191) cross-fades the head of a newly synthesised chunk with the tail of the previous one
under the two halves of a window (CosyVoice2's chunked synthesis applies it to the mel / speech overlap with a Hamming
window of twice the overlap).  The reference moves both tensors to the CPU and back; here the arithmetic stays where the
tensors are -- same values, same result device."""
from __future__ import annotations

import torch


def fade_in_out(fade_in_mel: torch.Tensor, fade_out_mel: torch.Tensor, window: torch.Tensor) -> torch.Tensor:
    mel_overlap_len = int(window.shape[0] / 2)
    window = window.to(device=fade_in_mel.device, dtype=fade_in_mel.dtype)
    out = fade_in_mel.clone()
    out[..., :mel_overlap_len] = (fade_in_mel[..., :mel_overlap_len] * window[:mel_overlap_len]
                                  + fade_out_mel.to(fade_in_mel.device)[..., -mel_overlap_len:] * window[mel_overlap_len:])
    return out
