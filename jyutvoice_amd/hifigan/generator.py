"""`HiFTGenerator` drop-in for inference (jyutvoice/hifigan/generator.py:239-466): same constructor keywords,
state-dict key names (both weight-norm spellings), `inference(speech_feat, cache_source) -> (wav, s)` and
`decode(x, s)`.  All arithmetic runs in libjyutvoice_hip.so (jv_hift_f0 / jv_hift_source / jv_hift_decode).

The reference's sine generator draws Uniform(-pi, pi) phases and N(0,1) noise per call (generator.py:155-158,
171), so `inference` is stochastic there too; here the nine phases per utterance come from a torch generator on the GPU and
the per-sample N(0,1) noise from a counter-based generator INSIDE the source kernel (jv_hift_source_seeded: Philox keyed by
the seed, counted per call -- `manual_seed` makes both repeatable), so the 9 x 480 T noise tensor is never materialised.
The library owns no RNG state: (seed, call) are arguments.  `Engine.hift_source(f0, phase, noise)` (jv_hift_source) injects a
caller's draws instead -- what the parity tests do with the oracle's."""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .. import spec
from ..engine import JV_MODEL_HIFT
from ..runtime import get_runtime
from .f0_predictor import ConvRNNF0Predictor


class HiFTGenerator:
    def __init__(self, in_channels: int = 80, base_channels: int = 512, nb_harmonics: int = 8, sampling_rate: int = 22050,
                 nsf_alpha: float = 0.1, nsf_sigma: float = 0.003, nsf_voiced_threshold: float = 10,
                 upsample_rates: List[int] = [8, 8], upsample_kernel_sizes: List[int] = [16, 16],
                 istft_params: Dict[str, int] = {"n_fft": 16, "hop_len": 4}, resblock_kernel_sizes: List[int] = [3, 7, 11],
                 resblock_dilation_sizes: List[List[int]] = [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 source_resblock_kernel_sizes: List[int] = [7, 11],
                 source_resblock_dilation_sizes: List[List[int]] = [[1, 3, 5], [1, 3, 5]], lrelu_slope: float = 0.1,
                 audio_limit: float = 0.99, f0_predictor: Optional[ConvRNNF0Predictor] = None, device="cuda:0"):
        got = (in_channels, base_channels, nb_harmonics, sampling_rate, float(nsf_alpha), float(nsf_sigma),
               float(nsf_voiced_threshold), tuple(upsample_rates), tuple(upsample_kernel_sizes), istft_params["n_fft"],
               istft_params["hop_len"], tuple(resblock_kernel_sizes), tuple(map(tuple, resblock_dilation_sizes)),
               tuple(source_resblock_kernel_sizes), float(lrelu_slope), float(audio_limit))
        want = (spec.N_FEATS, spec.HIFT_BASE_CH, spec.HIFT_NB_HARMONICS, spec.SAMPLE_RATE, spec.HIFT_NSF_ALPHA,
                spec.HIFT_NSF_SIGMA, spec.HIFT_VOICED_THRESHOLD, spec.HIFT_UP_RATES, spec.HIFT_UP_KERNELS, spec.HIFT_NFFT,
                spec.HIFT_HOP, spec.HIFT_RB_KERNELS, (spec.HIFT_RB_DILATIONS,) * 3, spec.HIFT_SRC_RB_KERNELS,
                spec.HIFT_LRELU_SLOPE, spec.HIFT_AUDIO_LIMIT)
        if got != want:
            raise NotImplementedError(f"libjyutvoice_hip is built for the base.yaml HiFT generator {want}; got {got}")
        self.sampling_rate = sampling_rate
        self.f0_predictor = f0_predictor
        self.device = torch.device(device)
        self._loaded = False
        self._gen: Optional[torch.Generator] = None
        self._seed: Optional[int] = None      # of the in-kernel source noise; drawn on first use unless manual_seed() set it
        self._calls = 0

    def to(self, device):
        self.device = torch.device(device)
        return self

    def eval(self):
        return self

    def manual_seed(self, seed: int):
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        self._seed, self._calls = int(seed), 0
        return self

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in spec.HIFT_INVENTORY if k not in state_dict]
        unexpected = [k for k in state_dict if k not in spec.HIFT_INVENTORY]
        if missing or (strict and unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for HiFTGenerator: Missing key(s): {missing[:6]}; "
                               f"Unexpected key(s): {unexpected[:6]}")
        for k, shape in spec.HIFT_INVENTORY.items():
            if tuple(state_dict[k].shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(state_dict[k].shape)} from "
                                   f"checkpoint, the shape in current model is {tuple(shape)}.")
        get_runtime(self.device).set_weights(JV_MODEL_HIFT, {k: state_dict[k] for k in spec.HIFT_INVENTORY})
        self._loaded = True
        return missing, unexpected

    def _engine(self, B, T):
        if not self._loaded:
            raise RuntimeError("HiFTGenerator: load_state_dict() has not been called")
        rt = get_runtime(self.device)
        return rt.ensure(B, T, 1)

    @torch.inference_mode()
    def decode(self, x: torch.Tensor, s: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """generator.py:396-432: mel [B,80,T] + source [B,1,480T] -> waveform [B,480T]"""
        B, _, T = x.shape
        return self._engine(B, T).hift_decode(x, s, lengths)

    @torch.inference_mode()
    def inference(self, speech_feat: torch.Tensor, cache_source: torch.Tensor = torch.zeros(1, 1, 0),
                  lengths: Optional[torch.Tensor] = None):
        """generator.py:450-466 -> (generated_speech [B,480T], s [B,1,480T])"""
        B, _, T = speech_feat.shape
        eng = self._engine(B, T)
        f0 = eng.hift_f0(speech_feat, lengths)
        phase = (torch.rand(B, 9, device=self.device, generator=self._gen) * 2 - 1) * math.pi
        if self._seed is None:      # (a host draw: no device synchronisation)
            self._seed = int(torch.empty((), dtype=torch.int64).random_().item())
        s = eng.hift_source_seeded(f0, phase, self._seed, self._calls)
        self._calls += 1
        if cache_source.shape[2] != 0:
            s[:, :, : cache_source.shape[2]] = cache_source.to(self.device)
        return eng.hift_decode(speech_feat, s, lengths), s
