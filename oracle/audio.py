"""Oracle (test infrastructure): the prompt-mel front-end -- `mel_spectrogram` / `extract_speech_feat`.

Follows jyutvoice/utils/audio.py:18-63 (reflect pad, torch.stft with a periodic Hann window, magnitude with the 1e-9 floor,
mel projection, log of the 1e-5 clamp) and infer.py:166-186 (n_fft 1920, hop 480, win 1920, 80 mels, 24 kHz, fmin 0,
fmax 8000, center=False; output [1, T, 80]).

The mel filterbank itself is third-party: `librosa.filters.mel` (librosa is imported by utils/audio.py:3 but is neither
pinned in requirements.txt nor present here).  `mel_basis_slaney` restates its published algorithm (htk=False Slaney mel
scale, triangular filters on the FFT bin centres, norm="slaney") -- **parity unpinned at that boundary**; everything
after the basis is the reference's own arithmetic.
"""
import math

import numpy as np
import torch


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis_slaney(sr=24000, n_fft=1920, n_mels=80, fmin=0.0, fmax=8000.0):
    """librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=) with its defaults htk=False, norm='slaney', dtype float32"""
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return torch.from_numpy(weights)


def mel_spectrogram(y, basis, n_fft=1920, hop_size=480, win_size=1920):
    """y [B, n] in [-1, 1] -> log-mel [B, 80, T]   (utils/audio.py:18-63 with center=False)"""
    pad = int((n_fft - hop_size) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=torch.hann_window(win_size),
                                         center=False, pad_mode="reflect", normalized=False, onesided=True,
                                         return_complex=True))
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5))


def extract_speech_feat(speech, basis):
    """infer.py:166-186: speech [1, n] at 24 kHz -> (speech_feat [1, T, 80], speech_feat_len [1] int32)"""
    feat = mel_spectrogram(speech, basis).squeeze(dim=0).transpose(0, 1).unsqueeze(dim=0)
    return feat, torch.tensor([feat.shape[1]], dtype=torch.int32)
