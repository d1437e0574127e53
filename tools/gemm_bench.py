#!/usr/bin/env python3
"""Micro-benchmark of the conv_gemm / attention kernels on the estimator's shapes at C3 (M = 19.3K rows).
    JV_TILE=0|1|2 python tools/gemm_bench.py      (tuning aid; prints TFLOP/s per shape)

Switches: JV_ONLY=qkv|ff1|ff2|out|res|conv3|attn (shape filter), JV_ATTN_L=frames (one attention length), JV_M=rows, JV_OP_X6=1 (bf16x6 main loop), JV_OP_H3=1 (fp16x3 main loop, linears), JV_TILE (force tile),
JV_ATTN_FP32=1.  With a tuning build (JV_TUNING=1 python -m jyutvoice_amd.build --force): JV_ABLATE=bits (1 no global loads in the
loop, 2 no LDS stores, 4 no barrier, 16 no epilogue, 64 no output stores, 256 slab epilogue) and JV_STAMPS=1 (per-workgroup
s_memtime phase breakdown on stderr)."""
import math
import os
import sys

os.environ.setdefault("JV_DYNAMIC_ENV", "1")

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jyutvoice_amd.engine import op_attention_planes, op_attention, op_attention_h3, op_conv_gemm, op_linear_h3, op_rowconv, op_rowgemm  # noqa: E402

dev = torch.device("cuda:0")
M = int(os.environ.get("JV_M", 4 + 64 * 304))
g = torch.Generator().manual_seed(0)


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


shapes = [("qkv   K256 N1536", 256, 1536, 1, None), ("ff1   K256 N1024", 256, 1024, 1, None),
          ("ff2   K1024 N256", 1024, 256, 1, None), ("out   K512 N256", 512, 256, 1, None),
          ("res   K256 N256", 256, 256, 1, None), ("conv3 K768 N256", 256, 256, 3, None),
          ("conv3+LN K768 N256", 256, 256, 3, "ln"), ("conv3+LN K960 N256", 320, 256, 3, "ln")]
only = os.environ.get("JV_ONLY")
for name, cin, n, taps, ln in shapes:
    if only and not name.startswith(only):
        continue
    A = torch.randn(M + 64, cin, generator=g).to(dev)
    W = (torch.randn(n, taps * cin, generator=g) / math.sqrt(taps * cin)).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    kw = {}
    if ln:
        kw["ln"] = (torch.ones(n, device=dev), torch.zeros(n, device=dev))
        kw["act"] = "mish"
    if os.environ.get("JV_OP_ROWCONV") and taps == 3 and n == 256:      # row-owning causal conv (rowconv_kernel.h)
        A = A[:M].contiguous()
        bound = A.abs().max().reshape(1)
        kw3 = dict(ln=kw["ln"], act="mish") if ln else {}
        t = timeit(lambda: op_rowconv(A, W, b, amax_in=bound, **kw3))
    elif os.environ.get("JV_OP_ROWGEMM") and taps == 1 and not ln and n % 256 == 0:      # row-owning fp16x3 GEMM (rowgemm_kernel.h)
        A = A[:M].contiguous()
        epi = os.environ.get("JV_ROWGEMM_EPI", "plain")
        kw2 = {}
        if epi in ("res", "res_ln"):
            if n != 256:
                continue
            kw2["res"] = torch.randn(M, n, generator=g).to(dev)
        if epi == "res_ln":
            kw2["ln"] = (torch.ones(n, device=dev), torch.zeros(n, device=dev))
        op_rowgemm(A, W, b, epi=epi, a_bound=8.0, **kw2)
        t = timeit(lambda: op_rowgemm(A, W, b, epi=epi, a_bound=8.0, presplit=2, **kw2))
    elif os.environ.get("JV_OP_H3") and taps == 1 and not ln:      # fp16x3 main loop (linears only)
        A = A[:M].contiguous()
        if os.environ.get("JV_H3_DMA_A"):       # A pre-split into fp16 planes once, both operands by LDS-DMA
            op_linear_h3(A, W, b, a_bound=8.0, presplit=1)
            t = timeit(lambda: op_linear_h3(A, W, b, a_bound=8.0, presplit=2))
        else:
            t = timeit(lambda: op_linear_h3(A, W, b, a_bound=8.0))
    else:
        t = timeit(lambda: op_conv_gemm(A, W, b, ntaps=taps, tap_row0=-(taps - 1), M=M, **kw))
    print(f"{name:22s} {t * 1e6:8.1f} us  {2.0 * M * n * taps * cin / t / 1e12:7.1f} TF")
for L in ((int(os.environ["JV_ATTN_L"]),) if os.environ.get("JV_ATTN_L") else (300, 512)):
    if only and only != "attn":
        continue
    B = 64 if L == 300 else 16
    S = L + 4
    qkv = torch.randn(4 + B * S + 8, 1536, generator=g).to(dev)
    lens = torch.full((B,), int(os.environ.get("JV_ATTN_LEN", L)), dtype=torch.int32, device=dev)      # JV_ATTN_LEN: valid keys (fewer key tiles, same queries)
    if os.environ.get("JV_OP_ATTN_PL"):      # K / V as planes by LDS-DMA (attention_pl.hip)
        bounds = tuple(8.0 * float(qkv[:, o:o + 512].abs().max()) for o in (0, 512, 1024))
        po = bool(os.environ.get("JV_ATTN_PLANES_OUT"))      # the result as fp16 planes, as the pipeline takes it (timing includes the test-side conversion: read the kernel from a trace)
        t = timeit(lambda: op_attention_planes(qkv, lens, B, 4, S, L, bounds, planes_out=po))
    elif os.environ.get("JV_OP_H3"):      # fp16x3 attention with bounds as a load-time L1-norm bound would give them
        bounds = tuple(8.0 * float(qkv[:, o:o + 512].abs().max()) for o in (0, 512, 1024))
        t = timeit(lambda: op_attention_h3(qkv, lens, B, 4, S, L, bounds))
    else:
        t = timeit(lambda: op_attention(qkv, lens, B, 4, S, L))
    print(f"attention L={L} B'={B:3d}   {t * 1e6:8.1f} us  {4.0 * B * 8 * L * L * 64 / t / 1e12:7.1f} TF")
