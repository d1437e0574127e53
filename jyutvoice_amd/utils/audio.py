"""`mel_spectrogram` / `extract_speech_feat` drop-ins (jyutvoice/utils/audio.py:18-63, infer.py:166-186): the prompt mel of
the voice-cloning branch, computed on the GPU by libjyutvoice_hip.so (jv_mel_spectrogram: STFT as a GEMM against a windowed
DFT basis, magnitude, mel projection, log).

The mel filterbank is data handed to the library.  The reference takes it from `librosa.filters.mel`; when librosa is
importable it is used here too, otherwise `slaney_mel_basis` evaluates the same published construction (Slaney mel scale,
triangular filters on the FFT bin centres, area normalisation) so that the module works without it."""
from __future__ import annotations

import math

import numpy as np
import torch

from ..runtime import get_runtime

_PARAMS = dict(n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000)
_loaded_on = set()


def slaney_mel_basis(sr=24000, n_fft=1920, n_mels=80, fmin=0.0, fmax=8000.0) -> np.ndarray:
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    for i in range(n_mels):
        weights[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    weights *= (2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels]))[:, np.newaxis]
    return weights


def mel_basis() -> torch.Tensor:
    try:
        from librosa.filters import mel as librosa_mel_fn
        m = librosa_mel_fn(sr=24000, n_fft=1920, n_mels=80, fmin=0, fmax=8000)
    except ImportError:
        m = slaney_mel_basis()
    return torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32))


def mel_spectrogram(y, n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000,
                    center=False, device="cuda:0"):
    """y [B, n] -> log-mel [B, 80, T] on the GPU; only the parameter set `extract_speech_feat` uses is built"""
    got = dict(n_fft=n_fft, num_mels=num_mels, sampling_rate=sampling_rate, hop_size=hop_size, win_size=win_size, fmin=fmin,
               fmax=fmax)
    if got != _PARAMS or center:
        raise NotImplementedError(f"libjyutvoice_hip computes the prompt mel of infer.py:166-186 only ({_PARAMS}, center=False)")
    if float(y.min()) < -1.0:
        print("min value is ", float(y.min()))
    if float(y.max()) > 1.0:
        print("max value is ", float(y.max()))
    dev = y.device if y.is_cuda else torch.device(device)
    B, n = y.shape
    eng = get_runtime(dev).ensure(1, 64, 1)
    if id(eng) not in _loaded_on:
        eng.load_mel_basis(mel_basis())
        _loaded_on.clear()
        _loaded_on.add(id(eng))
    return eng.mel_spectrogram(y)


def extract_speech_feat(speech, device="cuda:0"):
    """infer.py:166-186: speech [1, n] at 24 kHz -> (speech_feat [1, T, 80], speech_feat_len [1] int32)"""
    feat = mel_spectrogram(speech, device=device, **_PARAMS).squeeze(dim=0).transpose(0, 1).unsqueeze(dim=0)
    return feat, torch.tensor([feat.shape[1]], dtype=torch.int32, device=feat.device)
