"""GPU: the estimator behind the reference's operator seam and the streaming pieces around the
path (8(f) rank 2): the TensorRT-shaped plug-in object, the reference-side ctypes stub of INTEGRATION.md executed verbatim,
chunk-by-chunk synthesis with `streaming=True`, the vocoder's `cache_source` continuation and `fade_in_out`."""
import ctypes
import os
import re

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def md(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


@pytest.fixture(scope="module")
def eng(tts_sd, hift_sd, noise):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    from jyutvoice_amd.engine import JV_MODEL_HIFT, JV_MODEL_TTS, Engine
    e = Engine("cuda:0", max_batch=2, max_frames=256, max_tokens=64)
    e.load_state_dict(JV_MODEL_TTS, tts_sd)
    e.load_state_dict(JV_MODEL_HIFT, hift_sd)
    e.load_noise(noise)
    yield e
    e.close()


def cfg_inputs(T, seed, lens=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 80, T, generator=g)
    mu = torch.randn(2, 80, T, generator=g)
    mu[1] = 0
    spks = torch.randn(2, 80, generator=g)
    spks[1] = 0
    cond = torch.zeros(2, 80, T)
    t = torch.full((2,), 0.37)
    lens = lens or [T, T]
    mask = (torch.arange(T)[None, None] < torch.tensor(lens)[:, None, None]).float()
    return x, mask, mu, t, spks, cond


def test_trt_seam_protocol(eng, tts_sd):
    """the call sequence of ConditionalCFM.forward_estimator's non-Module branch (flow_matching.py:270-297) against
    jyutvoice_amd.flow.estimator.HipEstimator: acquire -> enter the stream -> declare shapes -> bind seven addresses by the
    engine's tensor names, the output onto x -> execute -> synchronise -> release; the result is read from x"""
    from jyutvoice_amd.flow.estimator import HipEstimator
    from oracle import flow as oflow
    est = HipEstimator(eng)
    assert not isinstance(est, torch.nn.Module)          # what selects the branch
    for T, lens in ((57, None), (40, [40, 23])):
        x, mask, mu, t, spks, cond = (v.cuda() for v in cfg_inputs(T, 300 + T, lens))
        want = oflow.estimator(tts_sd, x.cpu(), mask.cpu(), mu.cpu(), t.cpu(), spks.cpu(), cond.cpu())
        [context, stream], engine = est.acquire_estimator()
        with stream:
            assert context.set_input_shape("x", (2, 80, x.size(2)))
            assert context.set_input_shape("mask", (2, 1, x.size(2)))
            assert context.set_input_shape("mu", (2, 80, x.size(2)))
            assert context.set_input_shape("t", (2,))
            assert context.set_input_shape("spks", (2, 80))
            assert context.set_input_shape("cond", (2, 80, x.size(2)))
            ptrs = [v.contiguous().data_ptr() for v in (x, mask, mu, t, spks, cond)] + [x.data_ptr()]
            for i, ptr in enumerate(ptrs):
                assert context.set_tensor_address(engine.get_tensor_name(i), ptr)
            assert context.execute_async_v3(torch.cuda.current_stream().cuda_stream) is True
            torch.cuda.current_stream().synchronize()
        est.release_estimator(context, stream)
        assert md(x, want) <= 1e-4, T                   # out aliased x
    # the pool hands the one context back
    assert est.trt_context_pool.qsize() == 1
    ctx2, _ = est.acquire_estimator()
    with pytest.raises(RuntimeError, match="unbound"):
        from jyutvoice_amd.flow.estimator import HipEstimatorContext
        HipEstimatorContext(eng).execute_async_v3(0)
    est.release_estimator(*ctx2)


def test_trt_seam_orders_pool_stream_after_producer(eng, tts_sd):
    """solve_euler fills x / mu / t on the caller's current stream and forward_estimator then launches on the
    pool's own non-blocking stream.  acquire_estimator() must order that stream behind the producer: here the inputs are
    produced on the current stream BEHIND a long-running kernel (so they are certainly not ready when the host reaches
    execute_async_v3), and the result must still be the estimator of the finished inputs.  The pool also serialises: a
    second acquire blocks until release."""
    import threading

    from jyutvoice_amd.flow.estimator import HipEstimator
    from oracle import flow as oflow
    est = HipEstimator(eng, trt_concurrent=2)
    T = 48
    host = cfg_inputs(T, 900)
    want = oflow.estimator(tts_sd, *host)
    x, mask, mu, t, spks, cond = (torch.empty_like(v, device="cuda") for v in host)
    pinned = [v.pin_memory() for v in host]
    big = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    for _ in range(6):
        big = big @ big * 1e-3                      # ~ms of queued work ahead of the producers
    for dst, src in zip((x, mask, mu, t, spks, cond), pinned):
        dst.copy_(src, non_blocking=True)           # the producers: queued on the current stream, not finished
    [context, stream], engine = est.acquire_estimator()
    got_second = []
    th = threading.Thread(target=lambda: got_second.append(est.acquire_estimator()))
    th.start()
    with stream:
        for name, v in zip(("x", "mask", "mu", "t", "spks", "cond"), (x, mask, mu, t, spks, cond)):
            assert context.set_input_shape(name, tuple(v.shape))
        ptrs = [v.data_ptr() for v in (x, mask, mu, t, spks, cond)] + [x.data_ptr()]
        for i, ptr in enumerate(ptrs):
            assert context.set_tensor_address(engine.get_tensor_name(i), ptr)
        assert context.execute_async_v3(torch.cuda.current_stream().cuda_stream) is True
        torch.cuda.current_stream().synchronize()
    th.join(timeout=0.2)
    assert th.is_alive() and not got_second         # the second taker waits for the release
    est.release_estimator(context, stream)
    th.join(timeout=10)
    assert not th.is_alive() and len(got_second) == 1
    est.release_estimator(*got_second[0][0])
    assert md(x, want) <= 1e-4
    with pytest.raises(ValueError):
        HipEstimator(eng, trt_concurrent=0)


def test_integration_md_stub_runs_verbatim(eng, tts_sd, monkeypatch):
    """the reference-side binding, executed exactly as printed"""
    from jyutvoice_amd import _lib
    from oracle import flow as oflow
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Operator-level seam"):text.index("## 3. C ABI summary")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    stub = [b for b in blocks if "reference-side stub" in b]
    assert len(stub) == 1
    real = ctypes.CDLL
    monkeypatch.setattr(ctypes, "CDLL", lambda name, *a, **k: real(_lib.LIB_PATH if name == "libjyutvoice_hip.so" else name, *a, **k))
    ns = {}
    exec(compile(stub[0], "INTEGRATION.md#2b", "exec"), ns)
    mod = ns["HipEstimator"](eng._h)
    assert isinstance(mod, torch.nn.Module)
    x, mask, mu, t, spks, cond = (v.cuda() for v in cfg_inputs(64, 11, [64, 41]))
    for streaming in (False, True):
        got = mod(x, mask, mu, t, spks, cond, streaming=streaming)
        want = oflow.estimator(tts_sd, x.cpu(), mask.cpu(), mu.cpu(), t.cpu(), spks.cpu(), cond.cpu(), streaming=streaming)
        assert md(got, want) <= 1e-4, streaming
    eng.set_streaming(0)


def test_streaming_two_chunks(eng, tts_sd, hift_sd, noise):
    """chunk-by-chunk synthesis as CosyVoice2-style streaming drives the path: the flow decoder with streaming=True on the
    first 100 frames, then on all 150 -- chunk-causal attention (decoder.py:951-954) and causal convolutions make the
    finished chunks' frames independent of what arrives later -- each vocoded with HiFTGenerator.inference, the second call
    continuing the first one's source signal through cache_source (generator.py:462-465), and the speech overlap cross-faded
    with fade_in_out (utils/common.py:181-191).  Every stage against the oracle doing the same."""
    import jyutvoice_amd
    from jyutvoice_amd import spec
    from jyutvoice_amd.utils.common import fade_in_out
    from oracle import flow as oflow
    from oracle import hift as ohift
    from oracle import stream as ostream
    tts, hift = jyutvoice_amd.build_default("cuda:0")
    tts.load_state_dict(tts_sd)
    hift.load_state_dict(hift_sd)
    g = torch.Generator().manual_seed(8)
    T1, T2, n = 100, 150, 4
    mu = torch.randn(1, 80, T2, generator=g)
    spks = torch.randn(1, 80, generator=g)
    cond = torch.zeros(1, 80, T2)
    # ---- flow decoder, streaming=True, through the mirror of `self.decoder(...)` (flow_matching.py:356-401)
    mel1, _ = tts.decoder(mu=mu[:, :, :T1].cuda(), mask=torch.ones(1, 1, T1).cuda(), n_timesteps=n, temperature=1.0,
                          spks=spks.cuda(), cond=cond[:, :, :T1].cuda(), streaming=True)
    mel2, _ = tts.decoder(mu=mu.cuda(), mask=torch.ones(1, 1, T2).cuda(), n_timesteps=n, temperature=1.0, spks=spks.cuda(),
                          cond=cond.cuda(), streaming=True)
    stream_est = lambda *a, **k: oflow.estimator(*a, streaming=True, **k)
    want1 = oflow.cfm_solve(tts_sd, noise, mu[:, :, :T1], torch.ones(1, 1, T1), spks, cond[:, :, :T1], n, est=stream_est)
    want2 = oflow.cfm_solve(tts_sd, noise, mu, torch.ones(1, 1, T2), spks, cond, n, est=stream_est)
    assert md(mel1, want1) <= 1e-3 and md(mel2, want2) <= 1e-3
    assert md(mel1, want1) <= 3e-4 and md(mel2, want2) <= 3e-4
    # the streaming property itself: frames of the finished chunks do not move when the utterance grows ...
    assert md(mel2[:, :, :T1], mel1) <= 2e-5
    # ... which full attention does not give
    full2, _ = tts.decoder(mu=mu.cuda(), mask=torch.ones(1, 1, T2).cuda(), n_timesteps=n, temperature=1.0, spks=spks.cuda(),
                           cond=cond.cuda(), streaming=False)
    assert md(full2[:, :, :T1], mel1) > 1e-3
    # ---- vocoder: chunk 1, then chunk 2 = the new frames plus an overlap, continuing chunk 1's source signal
    w = ohift.fold_weight_norm(hift_sd)
    overlap = 8                                             # mel frames re-synthesised at the seam
    hift.manual_seed(5)
    wav1, s1 = hift.inference(mel1)
    start = T1 - overlap
    cache = s1[:, :, start * spec.HIFT_UPSAMPLE_TOTAL:]     # the source signal of the overlap, kept from chunk 1
    wav2, s2 = hift.inference(mel2[:, :, start:], cache_source=cache)
    assert torch.equal(s2[:, :, :cache.shape[2]], cache)
    o1 = ohift.decode(w, mel1.cpu(), s1.cpu())
    assert float((wav1.cpu() - o1).pow(2).mean().sqrt()) <= 1e-4
    # the oracle's inference on the same fresh source (s2 outside the overwritten head is what the source module produced)
    o2, os2 = ostream.hift_inference(w, mel2[:, :, start:].cpu(), s2.cpu(), cache.cpu())
    assert torch.equal(os2, s2.cpu())
    assert float((wav2.cpu() - o2).pow(2).mean().sqrt()) <= 1e-4
    # ---- cross-fade of the speech overlap (Hamming window of twice the overlap, as CosyVoice2 does)
    n_ov = overlap * spec.HIFT_UPSAMPLE_TOTAL
    window = torch.hamming_window(2 * n_ov, periodic=False)
    got = fade_in_out(wav2, wav1, window)
    want = ostream.fade_in_out(wav2.cpu(), wav1.cpu(), window)
    assert got.device == wav2.device and md(got, want) <= 1e-7
    assert torch.equal(got[..., n_ov:], wav2[..., n_ov:])
    # mel-domain use (the reference's argument names)
    mwin = torch.hamming_window(2 * overlap, periodic=False)
    assert md(fade_in_out(mel2[:, :, start:], mel1, mwin), ostream.fade_in_out(mel2[:, :, start:].cpu(), mel1.cpu(), mwin)) <= 1e-7
