"""Thin host wrapper around one libjyutvoice_hip context.

PyTorch is plumbing only: it owns the device buffers handed to the C ABI (raw `data_ptr()`s) and the
stream they are enqueued on.  Every numerical step of the hot path runs inside the library.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib, spec
from ._lib import JV_MODEL_HIFT, JV_MODEL_PROMPT, JV_MODEL_TTS, JvError, check


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


class Engine:
    """One library context = packed weights + workspace for up to `max_batch` utterances of
    `max_frames` mel frames / `max_tokens` tokens, bound to one GPU."""

    def __init__(self, device="cuda:0", max_batch=1, max_frames=2048, max_tokens=512):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("jyutvoice_amd runs on an AMD GPU only (device must be cuda:N); there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible to PyTorch-ROCm; jyutvoice_amd has no CPU path")
        self.max_batch, self.max_frames, self.max_tokens = max_batch, max_frames, max_tokens
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        check(self.lib.jv_create(C.byref(h), idx, max_batch, max_frames, max_tokens))
        self._h = h
        self._loaded = {JV_MODEL_TTS: False, JV_MODEL_HIFT: False, JV_MODEL_PROMPT: False}

    def reserve(self, max_batch, max_frames, max_tokens):
        """grow (or shrink) the workspace of the live context; weights stay loaded (jv_reserve)"""
        check(self.lib.jv_reserve(self._h, int(max_batch), int(max_frames), int(max_tokens)))
        self.max_batch, self.max_frames, self.max_tokens = max_batch, max_frames, max_tokens

    def broken(self) -> bool:
        """True once a failed jv_reserve could not restore the previous workspace: the context refuses every call"""
        return not getattr(self, "_h", None) or self.lib.jv_usable(self._h) == 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.jv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- registry ---------------------------------------------------------------------------------
    def registry(self, model: int) -> Dict[str, tuple]:
        out = {}
        for i in range(self.lib.jv_num_tensors(self._h)):
            if self.lib.jv_tensor_model(self._h, i) != model:
                continue
            nd = self.lib.jv_tensor_ndim(self._h, i)
            out[self.lib.jv_tensor_name(self._h, i).decode()] = tuple(
                int(self.lib.jv_tensor_dim(self._h, i, d)) for d in range(nd))
        return out

    # ---- weights ----------------------------------------------------------------------------------
    def load_state_dict(self, model: int, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Mirror of nn.Module.load_state_dict for the library's registry: same missing/unexpected-key
        semantics; shapes are checked by the library (RuntimeError on mismatch, like torch)."""
        expected = self.registry(model)
        missing = [k for k in expected if k not in sd]
        unexpected = [k for k in sd if k not in expected]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: Missing key(s): {missing[:8]}{'...' if len(missing) > 8 else ''}; "
                               f"Unexpected key(s): {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
        st = _stream(self.device)
        for k in expected:
            if k not in sd:
                continue
            t = _f32(sd[k].detach(), self.device)
            shape = (C.c_int64 * t.dim())(*t.shape)
            try:
                check(self.lib.jv_load_tensor(self._h, k.encode(), _ptr(t), shape, t.dim(), 1, st))
            except JvError as e:
                raise RuntimeError(e.msg) from None
        torch.cuda.synchronize(self.device)
        if not missing:
            check(self.lib.jv_finalize(self._h, model, st))
            self._loaded[model] = True
        return missing, unexpected

    def load_noise(self, noise: torch.Tensor):
        t = _f32(noise, self.device)
        check(self.lib.jv_load_noise(self._h, _ptr(t), t.numel(), 1, _stream(self.device)))
        torch.cuda.synchronize(self.device)

    # ---- flow ---------------------------------------------------------------------------------------
    def set_streaming(self, chunk_frames: int = spec.EST_STATIC_CHUNK):
        """chunk-causal estimator attention (the reference's streaming=True); 0 = full attention"""
        check(self.lib.jv_flow_set_streaming(self._h, int(chunk_frames)))

    def set_step_graph(self, on: bool = True):
        """replay the Euler step of cfm_solve as a captured hipGraph (default off: measured no faster; same results)"""
        check(self.lib.jv_flow_set_graph(self._h, 1 if on else 0))

    def set_exact_range(self, on: bool = True):
        """True: bf16x6 for every contraction; False (default): fp16x3 on the estimator linears whose input range is proven
        at load time (jv_flow_set_contraction)"""
        check(self.lib.jv_flow_set_contraction(self._h, 1 if on else 0))

    def contraction_info(self) -> dict:
        """which of the estimator's transformer layers have a usable load-time bound (fp16x3) and which stay on bf16x6
        (jv_flow_contraction_info)"""
        out = (C.c_int32 * 4)()
        check(self.lib.jv_flow_contraction_info(self._h, out, 4))
        return {"blocks": out[0], "blocks_all_h3": out[1], "linears_h3": out[2], "attention_h3": out[3]}

    def flow_estimator(self, x, mask_lens, mu, t, spks, cond):
        """[B2,80,T] tensors on the device; mask_lens int32 [B2] or None."""
        B2, _, T = x.shape
        x, mu, cond = (_f32(v, self.device) for v in (x, mu, cond))
        t, spks = _f32(t, self.device), _f32(spks, self.device)
        lens = None if mask_lens is None else mask_lens.to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty_like(x)
        check(self.lib.jv_flow_estimator_step(self._h, _ptr(x), _ptr(lens), _ptr(mu), _ptr(t), _ptr(spks), _ptr(cond), B2, T,
                                              _ptr(out), _stream(self.device)))
        return out

    def cfm_solve(self, mu, lens, spks, cond, n_timesteps, temperature=1.0, t_span=None):
        B, _, T = mu.shape
        mu, cond, spks = _f32(mu, self.device), _f32(cond, self.device), _f32(spks, self.device)
        lens_d = None if lens is None else lens.to(device=self.device, dtype=torch.int32).contiguous()
        mel = torch.empty_like(mu)
        ts = None
        if t_span is not None:
            ts_host = t_span.detach().to("cpu", torch.float32).contiguous()
            ts = (C.c_float * ts_host.numel())(*ts_host.tolist())
        check(self.lib.jv_cfm_solve(self._h, _ptr(mu), _ptr(lens_d), _ptr(spks), _ptr(cond), B, T, int(n_timesteps),
                                    float(temperature), ts, _ptr(mel), _stream(self.device)))
        return mel

    # ---- prompt mel front-end --------------------------------------------------------------------------
    def load_mel_basis(self, basis: torch.Tensor):
        t = basis.detach().to("cpu", torch.float32).contiguous()
        check(self.lib.jv_load_mel_basis(self._h, t.data_ptr(), t.numel(), 0, _stream(self.device)))

    def mel_spectrogram(self, wav):
        """utils/audio.py:18-63 with extract_speech_feat's parameters: wav [B, n] -> log-mel [B, 80, 1 + (n - 480) // 480]"""
        w = _f32(wav, self.device)
        B, n = w.shape
        T = 1 + (n - 480) // 480
        mel = torch.empty(B, spec.N_FEATS, max(T, 0), device=self.device)
        check(self.lib.jv_mel_spectrogram(self._h, _ptr(w), B, n, _ptr(mel), _stream(self.device)))
        return mel

    # ---- prompt branch --------------------------------------------------------------------------------
    def prompt_encoder(self, token, token_len):
        """FlowEncoder.forward (infer.py:66-83): token [B,Tk] int64, token_len [B] -> prompt_h [B, 2*Tk, 80]"""
        B, Tk = token.shape
        tok = token.to(device=self.device, dtype=torch.int64).contiguous()
        tl = token_len.to(device=self.device, dtype=torch.int64).contiguous()
        h = torch.empty(B, 2 * Tk, spec.N_FEATS, device=self.device)
        check(self.lib.jv_prompt_encoder_fwd(self._h, _ptr(tok), _ptr(tl), B, Tk, _ptr(h), _stream(self.device)))
        return h

    # ---- encoder --------------------------------------------------------------------------------------
    def encoder(self, x, x_lengths, lang, tone, word_pos, syllable_pos, spk_embed):
        B, Tt = x.shape
        ids = [v.to(device=self.device, dtype=torch.int64).contiguous() for v in (x, lang, tone, word_pos, syllable_pos)]
        xl = x_lengths.to(device=self.device, dtype=torch.int64).contiguous()
        spk = _f32(spk_embed, self.device)
        h = torch.empty(B, spec.ENC_HIDDEN, Tt, device=self.device)
        mu_x = torch.empty(B, spec.N_FEATS, Tt, device=self.device)
        logw = torch.empty(B, 1, Tt, device=self.device)
        c = torch.empty(B, spec.N_FEATS, device=self.device)
        check(self.lib.jv_encoder_fwd(self._h, *[_ptr(v) for v in ids], _ptr(xl), _ptr(spk), B, Tt, _ptr(h), _ptr(mu_x),
                                      _ptr(logw), _ptr(c), _stream(self.device)))
        return h, mu_x, logw, c

    def length_regulate(self, logw, x_lengths, mu_x, length_scale=1.0):
        B, _, Tt = logw.shape
        xl = x_lengths.to(device=self.device, dtype=torch.int64).contiguous()
        w_ceil = torch.empty(B, 1, Tt, device=self.device)
        y_lengths = torch.empty(B, dtype=torch.int64, device=self.device)
        st = _stream(self.device)
        check(self.lib.jv_length_regulate(self._h, _ptr(logw), _ptr(xl), _ptr(mu_x), B, Tt, float(length_scale), _ptr(w_ceil),
                                          _ptr(y_lengths), 0, None, None, st))
        ty = int(y_lengths.max().item())       # the reference's one host sync (jyutvoice_tts.py:187)
        attn = torch.empty(B, Tt, ty, device=self.device)
        mu_y = torch.empty(B, spec.N_FEATS, ty, device=self.device)
        check(self.lib.jv_length_regulate(self._h, _ptr(logw), _ptr(xl), _ptr(mu_x), B, Tt, float(length_scale), _ptr(w_ceil),
                                          _ptr(y_lengths), ty, _ptr(attn), _ptr(mu_y), st))
        return w_ceil, y_lengths, attn, mu_y

    # ---- HiFT -----------------------------------------------------------------------------------------
    def hift_f0(self, mel, lens=None):
        B, _, T = mel.shape
        mel = _f32(mel, self.device)
        lens_d = None if lens is None else lens.to(device=self.device, dtype=torch.int32).contiguous()
        f0 = torch.empty(B, T, device=self.device)
        check(self.lib.jv_hift_f0(self._h, _ptr(mel), _ptr(lens_d), B, T, _ptr(f0), _stream(self.device)))
        return f0

    def hift_source(self, f0, phase, noise):
        B, T = f0.shape
        f0, phase, noise = _f32(f0, self.device), _f32(phase, self.device), _f32(noise, self.device)
        s = torch.empty(B, 1, T * spec.HIFT_UPSAMPLE_TOTAL, device=self.device)
        check(self.lib.jv_hift_source(self._h, _ptr(f0), _ptr(phase), _ptr(noise), B, T, _ptr(s), _stream(self.device)))
        return s

    def hift_source_seeded(self, f0, phase, seed: int, call: int):
        """the source signal with its N(0,1) noise drawn inside the kernel from (seed, call) -- no [B,9,480T] noise tensor"""
        B, T = f0.shape
        f0, phase = _f32(f0, self.device), _f32(phase, self.device)
        s = torch.empty(B, 1, T * spec.HIFT_UPSAMPLE_TOTAL, device=self.device)
        check(self.lib.jv_hift_source_seeded(self._h, _ptr(f0), _ptr(phase), C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint32(call & 0xFFFFFFFF),
                                             B, T, _ptr(s), _stream(self.device)))
        return s

    def hift_decode(self, mel, s, lens=None):
        B, _, T = mel.shape
        mel, s = _f32(mel, self.device), _f32(s, self.device)
        lens_d = None if lens is None else lens.to(device=self.device, dtype=torch.int32).contiguous()
        wav = torch.empty(B, T * spec.HIFT_UPSAMPLE_TOTAL, device=self.device)
        check(self.lib.jv_hift_decode(self._h, _ptr(mel), _ptr(s), _ptr(lens_d), B, T, _ptr(wav), _stream(self.device)))
        return wav


# ---- operator-level helpers for the parity tests (same kernels the stages launch) ------------------------
def op_conv_gemm(A, W, bias=None, ntaps=1, tap_row0=0, dil=1, M=None, act="none", prologue="none", alpha=None, slope=0.0,
                 ln=None, ln_eps=1e-5, rowmask=None, res=None):
    """A [rows, Cin] cuda fp32; W [N, ntaps*Cin]; returns out [M, N]."""
    lib = _lib.load()
    rows, cin = A.shape
    N = W.shape[0]
    M = rows if M is None else M
    out = torch.empty(M, N, device=A.device)
    g, b = (ln if ln is not None else (None, None))
    check(lib.jv_op_conv_gemm(_ptr(A), rows, M, cin, ntaps, tap_row0, dil, _ptr(W), N, _ptr(bias), _lib.ACT[act],
                              _lib.PRO[prologue], _ptr(alpha), float(slope), _ptr(g), _ptr(b), float(ln_eps), _ptr(rowmask),
                              _ptr(res), _ptr(out), _stream(A.device)))
    return out


def op_linear_h3(A, W, bias=None, act="none", res=None, a_bound=None, presplit=0):
    """fp16x3 main loop (jv_flow_set_contraction): A [rows, K], W [N, K]; a_bound >= max |A| (default: measured);
    presplit=1: A goes through fp16 planes and LDS-DMA as in the estimator (2: reuse the previous planes, timing only)."""
    lib = _lib.load()
    rows, K = A.shape
    N = W.shape[0]
    out = torch.empty(rows, N, device=A.device)
    bound = float(A.abs().max()) if a_bound is None else float(a_bound)
    check(lib.jv_op_linear_h3(_ptr(A), rows, rows, K, _ptr(W), N, _ptr(bias), _lib.ACT[act], _ptr(res), bound, int(presplit),
                              _ptr(out), _stream(A.device)))
    return out


def op_attention(qkv, lens, B, G, S, L):
    lib = _lib.load()
    out = torch.zeros(qkv.shape[0], 512, device=qkv.device)
    check(lib.jv_op_attention(_ptr(qkv), _ptr(lens), B, G, S, L, _ptr(out), _stream(qkv.device)))
    return out


def op_conv_h3_measured(A, W, bias=None, ntaps=1, tap_row0=0, dil=1, M=None, act="none", prologue="none", alpha=None,
                        slope=0.0, rowmask=None, res=None, amax_in=None, a_extra=0.0, amax_out=None):
    """conv_gemm with the fp16x3 scale derived on the device from amax_in (a 1-element cuda tensor >= max |A|) + a_extra;
    amax_out (1-element cuda tensor, zeroed by the caller) receives max |out|.  amax_in=None: bf16x6, tracking only."""
    lib = _lib.load()
    rows, cin = A.shape
    N = W.shape[0]
    M = rows if M is None else M
    out = torch.empty(M, N, device=A.device)
    check(lib.jv_op_conv_h3_measured(_ptr(A), rows, M, cin, ntaps, tap_row0, dil, _ptr(W), N, _ptr(bias), _lib.ACT[act],
                                     _lib.PRO[prologue], _ptr(alpha), float(slope), _ptr(rowmask), _ptr(res), _ptr(amax_in),
                                     float(a_extra), _ptr(amax_out), _ptr(out), _stream(A.device)))
    return out


def op_attention_h3(qkv, lens, B, G, S, L, bounds=None):
    """fp16x3 attention kernel; bounds = (|q|, |k|, |v|) maxima the caller vouches for (default: measured)"""
    lib = _lib.load()
    out = torch.zeros(qkv.shape[0], 512, device=qkv.device)
    if bounds is None:
        bounds = tuple(float(qkv[:, o:o + 512].abs().max()) for o in (0, 512, 1024))
    check(lib.jv_op_attention_h3(_ptr(qkv), _ptr(lens), B, G, S, L, float(bounds[0]), float(bounds[1]), float(bounds[2]),
                                 _ptr(out), _stream(qkv.device)))
    return out


def op_rowgemm(A, W, bias=None, epi="plain", res=None, ln=None, a_bound=None, out2_scale=1.0, presplit=0, amax_out=None):
    """the row-owning fp16x3 GEMM (rowgemm_kernel.h): returns out fp32 [M,N] (plain / res), the fp16 planes as an fp32
    tensor h + l (gelu), or (out, planes) for res+ln.  Planes come back un-scaled (divided by out2_scale)."""
    lib = _lib.load()
    rows, K = A.shape
    N = W.shape[0]
    code = {"plain": 0, "gelu": 1, "res": 2, "res_ln": 3}[epi]
    bound = float(A.abs().max()) if a_bound is None else float(a_bound)
    out = torch.empty(rows, N, device=A.device) if code != 1 else None
    pc = 256 if code == 3 else N
    out2 = torch.empty(2, rows, pc, dtype=torch.float16, device=A.device) if code in (1, 3) else None
    g, b = (ln if ln is not None else (None, None))
    check(lib.jv_op_rowgemm(_ptr(A), rows, rows, K, _ptr(W), N, _ptr(bias), code, _ptr(res), _ptr(g), _ptr(b), bound,
                            float(out2_scale), int(presplit), _ptr(out), _ptr(out2), _ptr(amax_out), _stream(A.device)))
    if presplit == 2:      # timing call: no post-processing
        return out
    planes = None if out2 is None else (out2[0].double() + out2[1].double()) / out2_scale
    if code == 1:
        return planes
    return (out, planes) if code == 3 else out


def op_attention_planes(qkv, lens, B, G, S, L, bounds=None, chunk=0, planes_out=False, out2_scale=256.0):
    """attention_pl.hip (K / V as fp16 planes by LDS-DMA); planes_out: the result through the fp16-plane output path, returned
    un-scaled as fp64"""
    lib = _lib.load()
    rows = qkv.shape[0]
    if bounds is None:
        bounds = tuple(float(qkv[:, o:o + 512].abs().max()) for o in (0, 512, 1024))
    out = torch.zeros(rows, 512, device=qkv.device)
    out2 = torch.zeros(2, rows, 512, dtype=torch.float16, device=qkv.device) if planes_out else None
    check(lib.jv_op_attention_planes(_ptr(qkv), rows, _ptr(lens), B, G, S, L, float(bounds[0]), float(bounds[1]), float(bounds[2]),
                                     int(chunk), float(out2_scale), _ptr(out), _ptr(out2), _stream(qkv.device)))
    if planes_out:
        return (out2[0].double() + out2[1].double()) / out2_scale
    return out


def op_rowconv(A, W, bias=None, ln=None, act="none", rowmask=None, rowvec=None, res=None, amax_in=None, amax_out=None):
    """rowconv_kernel.h: causal k = 3 conv to 256 channels + LayerNorm / act / mask / + rowvec / + res; W [256, 3 * Cin]"""
    lib = _lib.load()
    rows, cin = A.shape
    out = torch.empty(rows, 256, device=A.device)
    g, b = (ln if ln is not None else (None, None))
    if amax_in is None:
        amax_in = A.abs().max().reshape(1)
    check(lib.jv_op_rowconv(_ptr(A), rows, rows, cin, _ptr(W), _ptr(bias), _ptr(g), _ptr(b), _lib.ACT[act], _ptr(rowmask),
                            _ptr(rowvec), _ptr(res), _ptr(amax_in), _ptr(amax_out), _ptr(out), _stream(A.device)))
    return out


def op_hiftconv(A, W, bias, alpha, ntaps, dil, rowmask=None, res1=None, res2=None, out_scale=1.0, prev=None, amax_out=None):
    """hiftconv_kernel.h: the vocoder's ResBlock convolution on a [rows, C] row buffer (C = 64 / 128 / 256); W [C, ntaps * C]
    tap-major; prev: accumulate onto this tensor"""
    lib = _lib.load()
    rows, C = A.shape
    out = prev.clone() if prev is not None else torch.empty(rows, C, device=A.device)
    masked = A if rowmask is None else A * rowmask[:, None].to(A.dtype)
    amax_in = torch.nan_to_num(masked).abs().max().reshape(1)
    extra = float((1.0 / (alpha + 1e-9)).max())
    check(lib.jv_op_hiftconv(_ptr(A), rows, C, int(ntaps), int(dil), _ptr(W), _ptr(bias), _ptr(alpha), _ptr(rowmask), _ptr(res1),
                             _ptr(res2), float(out_scale), 1 if prev is not None else 0, _ptr(amax_in), extra, _ptr(amax_out),
                             _ptr(out), _stream(A.device)))
    return out


def op_layernorm(x, g, b, eps=1e-5):
    lib = _lib.load()
    out = torch.empty_like(x)
    check(lib.jv_op_layernorm(_ptr(x), _ptr(g), _ptr(b), float(eps), x.shape[0], x.shape[1], _ptr(out), _stream(x.device)))
    return out


def profile_enable(on: bool) -> None:
    check(_lib.load().jv_profile_enable(1 if on else 0))


def profile_report() -> dict:
    """per-kernel {launches, ms, flops, bytes} since the last report (synchronises the device)"""
    import json
    buf = C.create_string_buffer(1 << 16)
    check(_lib.load().jv_profile_report(buf, len(buf)))
    return json.loads(buf.value.decode())
