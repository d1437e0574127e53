"""GPU: operator-level parity of the HIP kernels (called through the C ABI) against fp64 PyTorch on the CPU."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    return torch.device("cuda:0")


def pack_conv(w):
    """[Cout, Cin, k] -> [Cout, k*Cin] tap-major (the library's GEMM operand layout)"""
    return w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()


def conv_rows_ref(A, w, bias, tap_row0, dil):
    """reference for the row-buffer conv: out[m] = sum_j A[m + tap_row0 + j*dil] @ w[:, :, j].T, zero outside"""
    A = A.double()
    w = w.double()
    rows = A.shape[0]
    out = torch.zeros(rows, w.shape[0], dtype=torch.float64)
    for j in range(w.shape[2]):
        off = tap_row0 + j * dil
        src = torch.zeros_like(A)
        lo, hi = max(0, -off), min(rows, rows - off)
        if hi > lo:
            src[lo:hi] = A[lo + off:hi + off]
        out += src @ w[:, :, j].T
    if bias is not None:
        out += bias.double()
    return out


def rel_err(got, want):
    return float((got.double().cpu() - want).abs().max() / (want.abs().max() + 1e-12))


def row_err(got, want, scale):
    """worst row of |got - want| against that ROW's own scale (`scale` [rows]: what the row's sum is made of, e.g.
    (|A| @ |W|^T).max + |res|.max) -- a global maximum in the denominator would let a wrong small-magnitude row hide
    behind a large one"""
    e = (got.double().cpu() - want).abs().amax(dim=1)
    return float((e / scale.clamp_min(1e-30)).max())


@pytest.mark.parametrize("M,K,N", [(300, 256, 1536), (19, 1024, 256), (2, 320, 1024), (1000, 512, 80), (129, 256, 18),
                                   (4100, 256, 256)])
def test_linear(dev, M, K, N):
    from jyutvoice_amd.engine import op_conv_gemm
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    out = op_conv_gemm(A.to(dev), W.to(dev), b.to(dev))
    want = A.double() @ W.double().T + b.double()
    assert rel_err(out, want) < 2e-6


@pytest.mark.parametrize("act", ["relu", "gelu", "mish", "elu", "silu"])
def test_activations(dev, act):
    from jyutvoice_amd.engine import op_conv_gemm
    g = torch.Generator().manual_seed(7)
    A = torch.randn(200, 64, generator=g) * 2
    W = torch.randn(96, 64, generator=g) / 4
    out = op_conv_gemm(A.to(dev), W.to(dev), None, act=act)
    z = A.double() @ W.double().T
    want = {"relu": torch.relu, "gelu": F.gelu, "mish": F.mish, "elu": F.elu, "silu": F.silu}[act](z)
    assert float((out.double().cpu() - want).abs().max()) < 5e-6


def test_causal_conv_ln_mish_mask(dev):
    from jyutvoice_amd.engine import op_conv_gemm
    g = torch.Generator().manual_seed(11)
    rows, cin = 333, 320
    A = torch.randn(rows, cin, generator=g)
    w = torch.randn(256, cin, 3, generator=g) / math.sqrt(3 * cin)
    b = torch.randn(256, generator=g) * 0.1
    lg, lb = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    mask = (torch.rand(rows, generator=g) > 0.2).to(torch.uint8)
    res = torch.randn(rows, 256, generator=g)
    out = op_conv_gemm(A.to(dev), pack_conv(w).to(dev), b.to(dev), ntaps=3, tap_row0=-2, act="mish",
                       ln=(lg.to(dev), lb.to(dev)), rowmask=mask.to(dev), res=res.to(dev))
    z = conv_rows_ref(A * mask[:, None], w, b, -2, 1)
    z = F.layer_norm(z, (256,), lg.double(), lb.double(), 1e-5)
    want = F.mish(z) * mask[:, None].double() + res.double()
    assert float((out.double().cpu() - want).abs().max()) < 2e-5


@pytest.mark.parametrize("k,dil,C", [(11, 5, 64), (7, 3, 128), (3, 1, 256), (5, 1, 192)])
def test_dilated_conv_snake_residual(dev, k, dil, C):
    from jyutvoice_amd.engine import op_conv_gemm
    g = torch.Generator().manual_seed(k * 100 + dil)
    rows = 700
    A = torch.randn(rows, C, generator=g)
    w = torch.randn(C, C, k, generator=g) / math.sqrt(k * C)
    b = torch.randn(C, generator=g) * 0.1
    alpha = 1 + 0.1 * torch.randn(C, generator=g).abs()
    res = torch.randn(rows, C, generator=g)
    pad = dil * (k - 1) // 2
    out = op_conv_gemm(A.to(dev), pack_conv(w).to(dev), b.to(dev), ntaps=k, tap_row0=-pad, dil=dil, prologue="snake",
                       alpha=alpha.to(dev), res=res.to(dev))
    Ad = A.double()
    sn = Ad + (1.0 / (alpha.double() + 1e-9)) * torch.sin(Ad * alpha.double()) ** 2
    want = conv_rows_ref(sn, w, b, -pad, dil) + res.double()
    assert float((out.double().cpu() - want).abs().max()) < 2e-5
    # same thing expressed as torch's own conv1d (ties the row-offset convention to F.conv1d's padding)
    want2 = F.conv1d(sn.T[None], w.double(), b.double(), dilation=dil, padding=pad)[0].T + res.double()
    assert float((want - want2).abs().max()) < 1e-9


def test_snake_argument_ranges(dev):
    """Snake's sin^2 takes the short reduction for |alpha x| <= 2^15 and sinf beyond (csrc/jv_device.h sin2_small): both
    ranges, and rows mixing them, against fp64 sin of the same fp32 product"""
    from jyutvoice_amd.engine import op_conv_gemm
    g = torch.Generator().manual_seed(77)
    rows, C = 300, 64
    A = torch.randn(rows, C, generator=g)
    A[:100] *= 3000.0                       # |alpha x| up to ~1e4: still the short path
    A[100:200] *= 2.0e5                     # far beyond 2^15: sinf path
    A[200:, ::7] *= 1.0e5                   # mixed inside one 8-value group
    w = torch.zeros(C, C, 1)
    w[:, :, 0] = torch.eye(C)               # identity 1x1 conv: the output is the prologue itself
    alpha = 1 + 0.1 * torch.randn(C, generator=g).abs()
    out = op_conv_gemm(A.to(dev), pack_conv(w).to(dev), torch.zeros(C).to(dev), ntaps=1, tap_row0=0, prologue="snake",
                       alpha=alpha.to(dev)).double().cpu()
    arg = (A * alpha).double()              # the fp32 product both implementations take the sine of
    want = A.double() + (1.0 / (alpha + 1e-9)).double() * torch.sin(arg) ** 2
    err = (out - want).abs()
    assert float((err / want.abs().clamp_min(1.0)).max()) < 1e-6


def test_lrelu_prologue(dev):
    from jyutvoice_amd.engine import op_conv_gemm
    g = torch.Generator().manual_seed(5)
    A = torch.randn(130, 96, generator=g)
    w = torch.randn(512, 96, 7, generator=g) / math.sqrt(7 * 96)
    out = op_conv_gemm(A.to(dev), pack_conv(w).to(dev), None, ntaps=7, tap_row0=-3, prologue="lrelu", slope=0.1)
    want = conv_rows_ref(F.leaky_relu(A.double(), 0.1), w, None, -3, 1)
    assert float((out.double().cpu() - want).abs().max()) < 2e-5


@pytest.mark.parametrize("B,L,lens", [(3, 77, [77, 40, 1]), (2, 300, [300, 257]), (1, 512, [512]), (2, 33, [33, 32])])
def test_attention(dev, B, L, lens):
    from jyutvoice_amd.engine import op_attention
    g = torch.Generator().manual_seed(B * 1000 + L)
    G, gap = 4, 4
    S = L + gap
    rows = G + B * S + 8
    qkv = torch.randn(rows, 1536, generator=g)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    out = op_attention(qkv.to(dev), lens_t.to(dev), B, G, S, L).cpu()
    for b in range(B):
        blk = qkv[G + b * S: G + b * S + L].double()
        q, k, v = (blk[:, i * 512:(i + 1) * 512].view(L, 8, 64).transpose(0, 1) for i in range(3))
        s = q @ k.transpose(1, 2) / 8.0
        s[:, :, lens[b]:] = -1e10
        o = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(L, 512)
        got = out[G + b * S: G + b * S + L].double()
        assert float((got - o).abs().max()) < 5e-6, b


@pytest.mark.parametrize("C,eps", [(256, 1e-5), (576, 1e-4), (192, 1e-4)])
def test_layernorm(dev, C, eps):
    from jyutvoice_amd.engine import op_layernorm
    g = torch.Generator().manual_seed(C)
    x = torch.randn(1001, C, generator=g) * 3 + 1
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    out = op_layernorm(x.to(dev), w.to(dev), b.to(dev), eps)
    want = F.layer_norm(x.double(), (C,), w.double(), b.double(), eps)
    assert float((out.double().cpu() - want).abs().max()) < 1e-5


@pytest.fixture
def x6(monkeypatch):
    """route jv_op_conv_gemm through the bf16x6 main loop (three bf16 planes per operand, six MFMA products)"""
    monkeypatch.setenv("JV_OP_X6", "1")
    monkeypatch.delenv("JV_NO_X6", raising=False)


@pytest.mark.parametrize("M,K,N", [(300, 256, 1536), (19, 1024, 256), (1000, 512, 80), (4100, 256, 256)])
def test_x6_linear_fp32_accuracy(dev, x6, M, K, N):
    """bf16x6 must be as accurate as the fp32-MFMA kernel: same bound against fp64"""
    from jyutvoice_amd.engine import op_conv_gemm
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))      # rows of very different scale
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    out = op_conv_gemm(A.to(dev), W.to(dev), b.to(dev))
    want = A.double() @ W.double().T + b.double()
    assert rel_err(out, want) < 2e-6
    row_scale = A.double().abs().max(dim=1, keepdim=True).values
    assert float(((out.double().cpu() - want).abs() / row_scale).max()) < 2e-6


@pytest.mark.parametrize("k,dil,C", [(11, 5, 64), (7, 3, 128), (3, 1, 256)])
def test_x6_dilated_conv_snake_residual(dev, x6, k, dil, C):
    test_dilated_conv_snake_residual(dev, k, dil, C)


def test_x6_causal_conv_ln_mish_mask(dev, x6):
    test_causal_conv_ln_mish_mask(dev)


def test_x6_gelu_and_generic_epilogues(dev, x6):
    test_activations(dev, "gelu")
    test_activations(dev, "elu")
    test_lrelu_prologue(dev)


@pytest.mark.parametrize("tile", ["0", "1", "2", "3", "4"])
def test_x6_every_tile_variant(dev, x6, monkeypatch, tile):
    """each tile shape of the bf16x6 kernel (128x128, 64x128, 64x64, 160x128, 128x64) on one problem with ragged edges in M and N,
    plain / GELU / residual epilogues and a 3-tap causal window"""
    from jyutvoice_amd.engine import op_conv_gemm
    monkeypatch.setenv("JV_TILE", tile)
    g = torch.Generator().manual_seed(40 + int(tile))
    M, K, N = 1000, 256, 392
    A = torch.randn(M + 8, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    want = A[:M].double() @ W.double().T + b.double()
    assert rel_err(op_conv_gemm(A.to(dev), W.to(dev), b.to(dev), M=M), want) < 2e-6
    assert rel_err(op_conv_gemm(A.to(dev), W.to(dev), b.to(dev), M=M, act="gelu"), F.gelu(want)) < 2e-6
    assert rel_err(op_conv_gemm(A.to(dev), W.to(dev), b.to(dev), M=M, res=res.to(dev)), want + res.double()) < 2e-6
    w3 = torch.randn(N, K, 3, generator=g) / math.sqrt(3 * K)
    out = op_conv_gemm(A.to(dev), pack_conv(w3).to(dev), b.to(dev), ntaps=3, tap_row0=-2, M=M)
    assert rel_err(out, conv_rows_ref(A, w3, b, -2, 1)[:M]) < 2e-6


@pytest.mark.parametrize("tile,K", [("0", 1024), ("2", 1024), ("3", 256), ("4", 512)])
def test_x6_weight_dma_race_screen(dev, x6, monkeypatch, tile, K):
    """the weight planes reach LDS by LDS-DMA, ordered only by vmcnt + barrier (csrc/conv_gemm_x6_kernel.h): a misplaced read
    shows up as rare wrong tiles, so every variant is replayed on a long K loop and must reproduce bit for bit -- and
    match fp64 -- while other work keeps the memory system busy"""
    from jyutvoice_amd.engine import op_conv_gemm
    monkeypatch.setenv("JV_TILE", tile)
    g = torch.Generator().manual_seed(int(tile) * 7 + K)
    M, N = 3000, 384
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    first = op_conv_gemm(A, W, b)
    assert rel_err(first, A.double().cpu() @ W.double().cpu().T + b.double().cpu()) < 2e-6
    noise = torch.empty(64 << 20, device=dev)
    for i in range(40):
        noise.normal_()                                   # streaming writes in flight beside the GEMM
        assert torch.equal(op_conv_gemm(A, W, b), first), i


# ---- fp16x3 main loop (jv_flow_set_contraction): two fp16 planes per operand after an exact power-of-two scaling chosen
# from a proven bound on |A|; three MFMA products; the estimator's LayerNorm / attention / GELU fed linears use it -------

@pytest.mark.parametrize("M,K,N", [(300, 256, 1536), (19, 1024, 256), (1000, 512, 80), (4100, 256, 256), (5000, 256, 1024)])
def test_h3_linear_fp32_accuracy(dev, M, K, N):
    """LayerNorm-like input (|x| <= 16 max|g| + max|b| is the bound the library proves): same bound against fp64 as the
    fp32-MFMA and bf16x6 kernels"""
    from jyutvoice_amd.engine import op_linear_h3
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K) * torch.exp(torch.randn(N, 1, generator=g))   # rows of different scale
    b = torch.randn(N, generator=g)
    out = op_linear_h3(A.to(dev), W.to(dev), b.to(dev), a_bound=16 * 1.35 + 0.1)
    want = A.double() @ W.double().T + b.double()
    assert rel_err(out, want) < 2e-6
    col_scale = W.double().abs().max(dim=1).values.clamp_min(1e-30) * math.sqrt(K) + b.double().abs()
    assert float(((out.double().cpu() - want).abs() / col_scale).max()) < 2e-6


@pytest.mark.parametrize("tile", ["0", "1", "2", "3", "4"])
def test_h3_every_tile_variant(dev, monkeypatch, tile):
    """each tile shape of the main loop with two planes, ragged M and N, plain / GELU / residual epilogues"""
    from jyutvoice_amd.engine import op_linear_h3
    monkeypatch.setenv("JV_TILE", tile)
    g = torch.Generator().manual_seed(140 + int(tile))
    M, K, N = 1000, 256, 392
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    want = A.double() @ W.double().T + b.double()
    assert rel_err(op_linear_h3(A.to(dev), W.to(dev), b.to(dev)), want) < 2e-6
    assert rel_err(op_linear_h3(A.to(dev), W.to(dev), b.to(dev), act="gelu"), F.gelu(want)) < 2e-6
    assert rel_err(op_linear_h3(A.to(dev), W.to(dev), b.to(dev), res=res.to(dev)), want + res.double()) < 2e-6
    assert rel_err(op_linear_h3(A.to(dev), W.to(dev), b.to(dev), act="silu"), F.silu(want)) < 2e-6      # generic epilogue


def test_h3_range_contract(dev):
    """values AT the proven bound stay finite and exact to the same tolerance (the scale leaves fp16's 65504 a margin);
    values far below it degrade gracefully: absolute error <= 2^-25 / scale per element, i.e. ~1e-12 of the bound"""
    from jyutvoice_amd.engine import op_linear_h3
    g = torch.Generator().manual_seed(7)
    M, K, N = 512, 256, 256
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bound = 700.0
    A = (torch.rand(M, K, generator=g) * 2 - 1) * bound
    A[0, :] = bound
    A[1, :] = -bound
    out = op_linear_h3(A.to(dev), W.to(dev), a_bound=bound)
    want = A.double() @ W.double().T
    assert torch.isfinite(out).all()
    assert rel_err(out, want) < 2e-6
    for shrink in (1e-2, 1e-4, 1e-6):          # whole tensor far below the bound the scale was chosen for
        As = A * shrink
        out = op_linear_h3(As.to(dev), W.to(dev), a_bound=bound)
        want = As.double() @ W.double().T
        err = float((out.double().cpu() - want).abs().max())
        assert err < 2e-6 * float(want.abs().max()) + 1e-9 * bound * 1e-3, (shrink, err)


def test_h3_weight_dma_race_screen(dev, monkeypatch):
    """as test_x6_weight_dma_race_screen, for the two-plane layout with two weight buffers on the 64x64 tile"""
    from jyutvoice_amd.engine import op_linear_h3
    monkeypatch.setenv("JV_TILE", "2")
    g = torch.Generator().manual_seed(99)
    M, K, N = 3000, 1024, 384
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    first = op_linear_h3(A, W, b, a_bound=8.0)
    assert rel_err(first, A.double().cpu() @ W.double().cpu().T + b.double().cpu()) < 2e-6
    noise = torch.empty(64 << 20, device=dev)
    for i in range(40):
        noise.normal_()
        assert torch.equal(op_linear_h3(A, W, b, a_bound=8.0), first), i


@pytest.mark.parametrize("B,L,lens", [(2, 300, [300, 211]), (3, 64, [64, 1, 33]), (1, 512, [512]), (2, 150, [97, 150])])
def test_h3_attention(dev, B, L, lens):
    """the fp16x3 attention kernel against fp64, same tolerance as test_attention; bounds loose by 8x as a load-time
    L1-norm bound would be"""
    from jyutvoice_amd.engine import op_attention_h3
    g = torch.Generator().manual_seed(B * 1000 + L + 5)
    G, gap = 4, 4
    S = L + gap
    rows = G + B * S + 8
    qkv = torch.randn(rows, 1536, generator=g)
    qkv[:, 512:1024] *= 3.0          # sharper softmax rows
    lens_t = torch.tensor(lens, dtype=torch.int32)
    bounds = tuple(8.0 * float(qkv[:, o:o + 512].abs().max()) for o in (0, 512, 1024))
    out = op_attention_h3(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds).cpu()
    for b in range(B):
        blk = qkv[G + b * S: G + b * S + L].double()
        q, k, v = (blk[:, i * 512:(i + 1) * 512].view(L, 8, 64).transpose(0, 1) for i in range(3))
        s = q @ k.transpose(1, 2) / 8.0
        s[:, :, lens[b]:] = -1e10
        o = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(L, 512)
        got = out[G + b * S: G + b * S + L].double()
        assert float((got - o).abs().max()) < 5e-6, b
    # values AT the bound: finite, same tolerance
    qkv2 = qkv.clone()
    big = [float(qkv[:, o:o + 512].abs().max()) for o in (0, 512, 1024)]
    out2 = op_attention_h3(qkv2.to(dev), lens_t.to(dev), B, G, S, L, tuple(big)).cpu()
    assert torch.isfinite(out2).all()
    assert float((out2 - out).abs().max()) < 5e-6


@pytest.mark.parametrize("tile", ["0", "2", "3"])
def test_h3_presplit_operand_is_bit_identical(dev, monkeypatch, tile):
    """A written as fp16 planes by a producer kernel and fetched by LDS-DMA (what the estimator does) gives the same bits as
    the in-kernel split, on ragged shapes, for the tiles the estimator uses; plus the weight/operand DMA race screen"""
    from jyutvoice_amd.engine import op_linear_h3
    monkeypatch.setenv("JV_TILE", tile)
    g = torch.Generator().manual_seed(240 + int(tile))
    M, K, N = 1000, 512, 392
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    for kw in ({}, {"act": "gelu"}, {"res": res}):
        ref = op_linear_h3(A, W, b, a_bound=8.0, **kw)
        got = op_linear_h3(A, W, b, a_bound=8.0, presplit=1, **kw)
        assert torch.equal(ref, got), kw
    want = A.double().cpu() @ W.double().cpu().T + b.double().cpu()
    first = op_linear_h3(A, W, b, a_bound=8.0, presplit=1)
    assert rel_err(first, want) < 2e-6
    noise = torch.empty(64 << 20, device=dev)
    for i in range(30):
        noise.normal_()
        assert torch.equal(op_linear_h3(A, W, b, a_bound=8.0, presplit=1), first), i


@pytest.mark.parametrize("k,dil,C,scale", [(11, 5, 64, 1.0), (7, 3, 128, 30.0), (3, 1, 256, 1e-3)])
def test_h3_measured_bound_chain(dev, k, dil, C, scale):
    """the vocoder's / trunk's form of fp16x3: a producing convolution tracks max |out| on the device (amax_out), the
    consuming Snake convolution derives its scale from that slot + what Snake can add -- no host involvement, no overflow
    possible at any input magnitude.  Checked: the tracked maximum is exact, and the consumer meets the tolerance of the
    bf16x6 test above on the same problem, at three input scales"""
    from jyutvoice_amd.engine import op_conv_h3_measured
    g = torch.Generator().manual_seed(k * 100 + dil + 7)
    rows = 700
    X = torch.randn(rows, C, generator=g) * scale
    w0 = torch.randn(C, C, 3, generator=g) / math.sqrt(3 * C)
    w = torch.randn(C, C, k, generator=g) / math.sqrt(k * C)
    b = torch.randn(C, generator=g) * 0.1 * scale
    alpha = 1 + 0.1 * torch.randn(C, generator=g).abs()
    res = torch.randn(rows, C, generator=g) * scale
    pad = dil * (k - 1) // 2
    slot = torch.zeros(1, device=dev)
    # producer (bf16x6, tracking only): A = conv3(X)
    A = op_conv_h3_measured(X.to(dev), pack_conv(w0).to(dev), None, ntaps=3, tap_row0=-1, amax_out=slot)
    assert float(slot) == float(A.abs().max())
    # consumer: snake -> dilated conv -> + residual, scale from the slot
    extra = float((1.0 / (alpha + 1e-9)).max())
    slot2 = torch.zeros(1, device=dev)
    out = op_conv_h3_measured(A, pack_conv(w).to(dev), b.to(dev), ntaps=k, tap_row0=-pad, dil=dil, prologue="snake",
                              alpha=alpha.to(dev), res=res.to(dev), amax_in=slot, a_extra=extra, amax_out=slot2)
    Ad = A.double().cpu()
    sn = Ad + (1.0 / (alpha.double() + 1e-9)) * torch.sin(Ad * alpha.double()) ** 2
    want = conv_rows_ref(sn, w, b, -pad, dil) + res.double()
    assert torch.isfinite(out).all()
    assert float((out.double().cpu() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    assert float(slot2) == float(out.abs().max())
    # leaky-relu consumer (the up-sampling convolutions): |lrelu(x)| <= |x|, nothing added
    out = op_conv_h3_measured(A, pack_conv(w).to(dev), None, ntaps=k, tap_row0=-pad, dil=dil, prologue="lrelu", slope=0.1,
                              amax_in=slot, a_extra=0.0)
    want = conv_rows_ref(F.leaky_relu(Ad, 0.1), w, None, -pad, dil)
    assert float((out.double().cpu() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


def test_h3_measured_bound_extremes(dev):
    """a tensor with one huge element (5e4 beside O(1) values) and an all-zero tensor: finite, correct"""
    from jyutvoice_amd.engine import op_conv_h3_measured
    g = torch.Generator().manual_seed(3)
    M, K, N = 256, 256, 128
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    A = torch.randn(M, K, generator=g)
    A[17, 5] = 5.0e4
    slot = torch.tensor([float(A.abs().max())], device=dev)
    out = op_conv_h3_measured(A.to(dev), W, None, amax_in=slot)
    want = A.double() @ W.double().cpu().T
    assert torch.isfinite(out).all()
    err = (out.double().cpu() - want).abs()
    assert float(err.max()) < 2e-6 * float(want.abs().max())
    # rows without the outlier: the scale chosen for 5e4 leaves O(1) values an absolute error of ~1e-8 per element
    assert float(err[torch.arange(M) != 17].max()) < 1e-5
    Z = torch.zeros(M, K, device=dev)
    out = op_conv_h3_measured(Z, W, None, amax_in=torch.zeros(1, device=dev))
    assert float(out.abs().max()) == 0.0


def test_h3_measured_bound_not_finite(dev):
    """a producer that already wrote inf or NaN leaves that bit pattern in the slot (integer max of bit patterns); the
    consumer then returns NaN for the rows of that slot ON PURPOSE (scale 0: jv_device.h h3_scale_dev) instead of scaling
    the finite rows into fp16 overflow with a scale derived from a NaN"""
    from jyutvoice_amd.engine import op_conv_h3_measured
    g = torch.Generator().manual_seed(4)
    M, K, N = 128, 128, 64
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    A = torch.randn(M, K, generator=g).to(dev)
    for bad in (float("nan"), float("inf")):
        out = op_conv_h3_measured(A, W, None, amax_in=torch.tensor([bad], device=dev))
        assert torch.isnan(out).all(), bad
    # tracking propagates a NaN written by the producer into the slot
    A2 = A.clone()
    A2[5, 7] = float("nan")
    slot = torch.zeros(1, device=dev)
    op_conv_h3_measured(A2, W, None, amax_out=slot)
    assert not torch.isfinite(slot).all()


# ---- row-owning fp16x3 GEMM (rowgemm_kernel.h): the estimator's transformer linears at chip-filling batch sizes --------
@pytest.mark.parametrize("rt", ["2", "3", "4", "5"])
@pytest.mark.parametrize("K,N", [(512, 256), (256, 1024), (64, 512)])
def test_rowgemm_plain_every_tile_height(dev, monkeypatch, rt, K, N):
    """fp32-level accuracy against fp64 on ragged row counts (last workgroup partly empty), every tile height, one and
    several 256-column chunks, K of one to sixteen steps; and the same result as the tile kernels' fp16x3 on the same operands
    up to the summation order inside an MFMA (16x16x32 here, 32x32x16 there)"""
    from jyutvoice_amd.engine import op_linear_h3, op_rowgemm
    monkeypatch.setenv("JV_ROWGEMM_RT", rt)
    g = torch.Generator().manual_seed(int(rt) * 1000 + K + N)
    M = 1000 + 7 * int(rt)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    out = op_rowgemm(A, W, b, a_bound=8.0)
    want = A.double().cpu() @ W.double().cpu().T + b.double().cpu()
    assert rel_err(out, want) < 2e-6
    assert rel_err(out, op_linear_h3(A, W, b, a_bound=8.0).double().cpu()) < 1e-6


def test_rowgemm_epilogues(dev, monkeypatch):
    """GELU -> planes, + residual (with the measured-bound tracking), + residual -> LayerNorm -> planes: each against fp64"""
    from jyutvoice_amd.engine import op_rowgemm
    g = torch.Generator().manual_seed(77)
    M, K = 2100, 256
    A = torch.randn(M, K, generator=g).to(dev)
    W1 = (torch.randn(1024, K, generator=g) / math.sqrt(K)).to(dev)
    b1 = torch.randn(1024, generator=g).to(dev)
    Ad, W1d, b1d = A.double().cpu(), W1.double().cpu(), b1.double().cpu()
    # exact GELU, written as the next GEMM's operand: h + l planes reproduce it to 22 bits
    got = op_rowgemm(A, W1, b1, epi="gelu", a_bound=8.0, out2_scale=64.0).cpu()
    want = F.gelu(Ad @ W1d.T + b1d)
    assert float((got - want).abs().max()) < 2e-6 * float(want.abs().max())
    # + residual, tracked maximum exact
    W2 = (torch.randn(256, K, generator=g) / math.sqrt(K)).to(dev)
    b2 = torch.randn(256, generator=g).to(dev)
    res = torch.randn(M, 256, generator=g).to(dev)
    slot = torch.zeros(1, device=dev)
    out = op_rowgemm(A, W2, b2, epi="res", res=res, a_bound=8.0, amax_out=slot)
    want = Ad @ W2.double().cpu().T + b2.double().cpu() + res.double().cpu()
    assert rel_err(out, want) < 2e-6
    assert float(slot) == float(out.abs().max())
    # row by row, with residual rows spread over six decades (a row whose residual is tiny next to one whose residual is
    # huge): every row within 2e-6 of ITS OWN scale -- the operand bound times the weights' row sums (what the fp16x3 split
    # guarantees, rowgemm_kernel.h) plus its residual
    decades = 10.0 ** (torch.arange(M) % 7 - 3).double()
    res_w = (res.double().cpu() * decades[:, None]).float().to(dev)
    out_w = op_rowgemm(A, W2, b2, epi="res", res=res_w, a_bound=8.0)
    want_w = Ad @ W2.double().cpu().T + b2.double().cpu() + res_w.double().cpu()
    scale = 8.0 * W2.double().cpu().abs().sum(dim=1).max() + res_w.double().cpu().abs().amax(dim=1)
    assert row_err(out_w, want_w, scale) < 2e-6
    out_ln, planes_w = op_rowgemm(A, W2, b2, epi="res_ln", res=res_w, ln=(torch.ones(256, device=dev), torch.zeros(256, device=dev)),
                                  a_bound=8.0, out2_scale=1024.0)
    assert row_err(out_ln, want_w, scale) < 2e-6
    # the LayerNorm planes of those rows are unit-scale whatever the row's magnitude: an absolute bound IS per-row relative
    ln_w = F.layer_norm(want_w, (256,), None, None, 1e-5)
    assert float((planes_w.double().cpu() - ln_w).abs().max()) < 1e-5
    # in place (res aliases out, as the estimator runs it) + LayerNorm of the new row -> planes
    lg = (1 + 0.1 * torch.randn(256, generator=g)).to(dev)
    lb = (0.02 * torch.randn(256, generator=g)).to(dev)
    out, planes = op_rowgemm(A, W2, b2, epi="res_ln", res=res, ln=(lg, lb), a_bound=8.0, out2_scale=1024.0)
    assert rel_err(out, want) < 2e-6
    ln_want = F.layer_norm(want, (256,), lg.double().cpu(), lb.double().cpu(), 1e-5)
    assert float((planes.cpu() - ln_want).abs().max()) < 5e-6
    # the DMA ring under memory pressure: identical bits over repeated launches beside a streaming writer
    first = op_rowgemm(A, W2, b2, epi="res", res=res, a_bound=8.0)
    noise = torch.empty(64 << 20, device=dev)
    for i in range(20):
        noise.normal_()
        assert torch.equal(op_rowgemm(A, W2, b2, epi="res", res=res, a_bound=8.0), first), i


@pytest.mark.parametrize("B,L,lens", [(2, 300, [300, 217]), (3, 70, [70, 1, 33]), (1, 512, [512]), (2, 40, [40, 40]),
                                      (2, 129, [129, 64])])
def test_attention_planes(dev, B, L, lens):
    """attention_pl.hip (K / V taken pre-split by LDS-DMA, V read transposed from a row-major LDS image) against fp64, on
    every workgroup size (2 / 4 / 8 waves), ragged key masks, fp32 and fp16-plane output, and the chunk-causal mask"""
    from jyutvoice_amd.engine import op_attention_planes
    g = torch.Generator().manual_seed(B * 1000 + L + 9)
    G, gap = 4, 4
    S = L + gap
    rows = G + B * S + 8
    qkv = torch.randn(rows, 1536, generator=g)
    qkv[:, 512:1024] *= 3.0
    lens_t = torch.tensor(lens, dtype=torch.int32)
    bounds = tuple(4.0 * float(qkv[:, o:o + 512].abs().max()) for o in (0, 512, 1024))

    def ref(chunk):
        outs = []
        for b in range(B):
            blk = qkv[G + b * S: G + b * S + L].double()
            q, k, v = (blk[:, i * 512:(i + 1) * 512].view(L, 8, 64).transpose(0, 1) for i in range(3))
            s = q @ k.transpose(1, 2) / 8.0
            s[:, :, lens[b]:] = -1e10
            if chunk:
                i = torch.arange(L)
                s = s.masked_fill(~(i[None, :] < ((i // chunk + 1) * chunk)[:, None])[None], -1e10)
            outs.append((torch.softmax(s, -1) @ v).transpose(0, 1).reshape(L, 512))
        return outs

    for chunk in (0, 50):
        want = ref(chunk)
        out = op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds, chunk=chunk).cpu()
        pl = op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds, chunk=chunk, planes_out=True).cpu()
        for b in range(B):
            sl = slice(G + b * S, G + b * S + L)
            assert float((out[sl].double() - want[b]).abs().max()) < 5e-6, (b, chunk)
            assert float((pl[sl] - want[b]).abs().max()) < 5e-6, (b, chunk)
    # the DMA ring under memory pressure: identical bits over repeated launches beside a streaming writer
    first = op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds)
    noise = torch.empty(64 << 20, device=dev)
    for i in range(10):
        noise.normal_()
        assert torch.equal(op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds), first), i


@pytest.mark.parametrize("B,L,lens", [(2, 300, [300, 217]), (3, 70, [70, 1, 33]), (1, 512, [512]), (2, 40, [40, 40]),
                                      (2, 129, [129, 64]), (2, 330, [330, 321]), (1, 1100, [1093])])
@pytest.mark.parametrize("qt", [None, "2", "3"])
@pytest.mark.parametrize("form", ["JV_OP_ATTN_ROWS", "JV_OP_ATTN_SINGLE"])
def test_attention_rows(dev, monkeypatch, B, L, lens, qt, form):
    """attention_r.hip (one workgroup per head and 64 QT queries, a wave's K / V^T fragments used for QT 16-query tiles,
    v_mfma_f32_16x16x32_f16, per-lane row sums) against fp64: its own choice of QT and forced smaller ones (several
    workgroups per head, idle waves), ragged key masks incl. a last key tile that straddles the mask, fp32 and fp16-plane
    output; identical bits over repeated launches beside a streaming writer (the DMA ring)"""
    from jyutvoice_amd.engine import op_attention_planes
    monkeypatch.setenv(form, "1")      # attention_r.hip (80 queries per wave) / attention_s.hip (160, one wave per SIMD)
    if qt:
        monkeypatch.setenv("JV_ATTN_QT", qt)
    g = torch.Generator().manual_seed(B * 1000 + L + 17)
    G, gap = 4, 4
    S = L + gap
    rows = G + B * S + 8
    qkv = torch.randn(rows, 1536, generator=g)
    qkv[:, 512:1024] *= 3.0
    lens_t = torch.tensor(lens, dtype=torch.int32)
    bounds = tuple(4.0 * float(qkv[:, o:o + 512].abs().max()) for o in (0, 512, 1024))
    out = op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds).cpu()
    pl = op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds, planes_out=True).cpu()
    for b in range(B):
        blk = qkv[G + b * S: G + b * S + L].double()
        q, k, v = (blk[:, i * 512:(i + 1) * 512].view(L, 8, 64).transpose(0, 1) for i in range(3))
        s = q @ k.transpose(1, 2) / 8.0
        s[:, :, lens[b]:] = -1e10
        want = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(L, 512)
        sl = slice(G + b * S, G + b * S + L)
        assert float((out[sl].double() - want).abs().max()) < 5e-6, b
        assert float((pl[sl] - want).abs().max()) < 5e-6, b
    first = op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds)
    noise = torch.empty(64 << 20, device=dev)
    for i in range(6):
        noise.normal_()
        assert torch.equal(op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds), first), i


@pytest.mark.parametrize("B,L,lens", [(2, 300, [300, 217]), (3, 70, [70, 1, 33]), (2, 129, [129, 64]), (2, 320, [320, 299])])
def test_attention_single_equals_planes_bit_for_bit(dev, monkeypatch, B, L, lens):
    """attn64_s (attention_s.hip, one wave per SIMD) and attn64_pl (attention_pl.hip, the DMA-ring form) are chosen by batch
    size (flow.hip: attention64_single_fits): an utterance's mel must not depend on which one its batch got, so the two must
    agree BIT FOR BIT -- same lazy reference maximum, same summation order -- on the fp32 output and on the fp16-plane output
    the pipeline takes, with ragged key masks."""
    from jyutvoice_amd.engine import op_attention_planes
    g = torch.Generator().manual_seed(B * 977 + L)
    G, gap = 4, 4
    S = L + gap
    rows = G + B * S + 8
    qkv = torch.randn(rows, 1536, generator=g)
    qkv[:, 512:1024] *= 3.0
    lens_t = torch.tensor(lens, dtype=torch.int32)
    bounds = tuple(4.0 * float(qkv[:, o:o + 512].abs().max()) for o in (0, 512, 1024))
    monkeypatch.delenv("JV_OP_ATTN_SINGLE", raising=False)
    monkeypatch.delenv("JV_OP_ATTN_ROWS", raising=False)
    pl32 = op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds).cpu()
    pl16 = op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds, planes_out=True).cpu()
    monkeypatch.setenv("JV_OP_ATTN_SINGLE", "1")
    s32 = op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds).cpu()
    s16 = op_attention_planes(qkv.to(dev), lens_t.to(dev), B, G, S, L, bounds, planes_out=True).cpu()
    for b in range(B):
        sl = slice(G + b * S, G + b * S + L)      # (rows between the utterances are not written by either form)
        assert torch.equal(pl32[sl], s32[sl]), b
        assert torch.equal(pl16[sl], s16[sl]), b


@pytest.mark.parametrize("k,dil,C,rows", [(11, 5, 64, 1500), (7, 3, 128, 700), (3, 1, 256, 333), (11, 5, 256, 250), (7, 1, 64, 321),
                                          (3, 5, 128, 161), (11, 1, 128, 40)])
def test_hiftconv(dev, k, dil, C, rows):
    """hiftconv_kernel.h (the vocoder's ResBlock convolutions, row-owning: the whole Snake'd window in LDS, weights in fragment
    order) against fp64 and torch's own conv1d: every kernel size / dilation / channel count of generator.py:90-97, several
    workgroups with a ragged last one and windows that reach past both ends of the buffer, masked rows holding NaN (selected,
    never multiplied), both residuals + scaling + accumulation, and the measured-bound tracking of what is stored"""
    from jyutvoice_amd.engine import op_hiftconv
    g = torch.Generator().manual_seed(k * 1000 + dil * 100 + C)
    A = torch.randn(rows, C, generator=g) * 2
    w = torch.randn(C, C, k, generator=g) / math.sqrt(k * C)
    b = torch.randn(C, generator=g) * 0.1
    alpha = 1 + 0.1 * torch.randn(C, generator=g).abs()
    res1, res2, prev = (torch.randn(rows, C, generator=g) for _ in range(3))
    mask = (torch.rand(rows, generator=g) > 0.1).to(torch.uint8)
    Abad = A.clone()
    Abad[mask == 0] = float("nan")                      # what a masked row holds must not matter
    pad = dil * (k - 1) // 2
    Ad = (A * mask[:, None]).double()
    sn = Ad + (1.0 / (alpha.double() + 1e-9)) * torch.sin(Ad * alpha.double()) ** 2
    sn = sn * mask[:, None].double()                    # masked rows read as zero AFTER the activation too (Snake(0) = 0)
    conv = F.conv1d(sn.T[None], w.double(), b.double(), dilation=dil, padding=pad)[0].T
    # plain
    out = op_hiftconv(A.to(dev), pack_conv(w).to(dev), b.to(dev), alpha.to(dev), k, dil, rowmask=mask.to(dev))
    assert float((out.double().cpu() - conv).abs().max()) < 2e-5
    # NaN in masked rows, residuals, scale, accumulation, tracking
    amax = torch.zeros(1, device=dev)
    out = op_hiftconv(Abad.to(dev), pack_conv(w).to(dev), b.to(dev), alpha.to(dev), k, dil, rowmask=mask.to(dev), res1=res1.to(dev),
                      res2=res2.to(dev), out_scale=1.0 / 3.0, prev=prev.to(dev), amax_out=amax)
    want = ((conv + res1.double()) + res2.double()) / 3.0 + prev.double()
    assert float((out.double().cpu() - want).abs().max()) < 2e-5
    tracked = out.cpu()[mask.bool()].abs().max()
    assert float(amax.cpu()) == float(tracked)          # exactly the largest stored magnitude over unmasked rows
    # in place on a residual (c2 of a ResBlock writes the buffer it adds)
    buf = res1.clone().to(dev)
    inpl = op_hiftconv(A.to(dev), pack_conv(w).to(dev), b.to(dev), alpha.to(dev), k, dil, rowmask=mask.to(dev), res1=buf)
    assert float((inpl.double().cpu() - (conv + res1.double())).abs().max()) < 2e-5


@pytest.mark.parametrize("rt", ["2", "3", "5"])
@pytest.mark.parametrize("cin", [256, 512, 64])
def test_rowconv(dev, monkeypatch, rt, cin):
    """rowconv_kernel.h: the estimator's causal k = 3 convolution with its row-wise tail in the epilogue, against fp64:
    conv -> LayerNorm -> Mish -> mask -> + time-embedding vector -> + residual (CausalBlock1D + the resnet's additions,
    decoder.py:110-115, 784-788), the bare convolution (down / up convs), the tracked maximum, and the DMA-ring race screen"""
    from jyutvoice_amd.engine import op_rowconv
    monkeypatch.setenv("JV_ROWGEMM_RT", rt)
    g = torch.Generator().manual_seed(int(rt) * 100 + cin)
    rows = 1500 + 11 * int(rt)
    A = torch.randn(rows, cin, generator=g) * 2.5
    w = torch.randn(256, cin, 3, generator=g) / math.sqrt(3 * cin)
    b = torch.randn(256, generator=g) * 0.1
    lg, lb = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    mask = (torch.rand(rows, generator=g) > 0.2).to(torch.uint8)
    res = torch.randn(rows, 256, generator=g)
    vec = torch.randn(256, generator=g)
    slot = torch.zeros(1, device=dev)
    out = op_rowconv(A.to(dev), pack_conv(w).to(dev), b.to(dev), ln=(lg.to(dev), lb.to(dev)), act="mish", rowmask=mask.to(dev),
                     rowvec=vec.to(dev), res=res.to(dev), amax_out=slot)
    z = conv_rows_ref(A * mask[:, None], w, b, -2, 1)
    z = F.layer_norm(z, (256,), lg.double(), lb.double(), 1e-5)
    want = F.mish(z) * mask[:, None].double() + vec.double() + res.double()
    assert float((out.double().cpu() - want).abs().max()) < 2e-5
    assert float(slot) == float(out[mask.bool().to(dev)].abs().max())          # masked rows are not tracked
    # the bare convolution, bound well above the data (a stale maximum from an earlier, larger state of the buffer)
    plain = op_rowconv(A.to(dev), pack_conv(w).to(dev), b.to(dev), amax_in=torch.tensor([40.0], device=dev))
    want = conv_rows_ref(A, w, b, -2, 1)
    assert float((plain.double().cpu() - want).abs().max()) < 2e-6 * float(want.abs().max()) * 8
    noise = torch.empty(64 << 20, device=dev)
    for i in range(10):
        noise.normal_()
        assert torch.equal(op_rowconv(A.to(dev), pack_conv(w).to(dev), b.to(dev), amax_in=torch.tensor([40.0], device=dev)), plain), i
