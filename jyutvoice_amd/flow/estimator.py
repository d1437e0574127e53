"""The estimator as a plug-in for the reference's TensorRT seam (SURVEY.md 8(b) "Operator seam", 8(f) rank 4).

`ConditionalCFM.forward_estimator` (jyutvoice/flow/flow_matching.py:267-297) takes its second branch for any estimator
that is not an `nn.Module`: it calls `acquire_estimator()` for `([context, stream], engine)`, enters `stream`, declares the
input shapes on the context, binds seven raw device addresses by the engine's tensor names -- x, mask, mu, t, spks, cond and,
as the seventh, `x.data_ptr()` again for the output -- runs `execute_async_v3(stream_handle)`, synchronises and hands the
context back with `release_estimator()`; the result is read from `x`.  The reference fills that seam with
`TrtContextWrapper` (jyutvoice/utils/common.py:219-238) around an engine built from the ONNX export
(scripts/export_onnx.py:228-283).  `HipEstimator` is the same three-object surface around libjyutvoice_hip.so, so

    cfm.estimator = HipEstimator(engine)          # engine: jyutvoice_amd.engine.Engine with the TTS weights loaded

leaves the reference's own Euler/CFG solver (solve_euler, :215-265) running unmodified on the HIP estimator.  No ONNX
export is needed (or offered): the weights go in through load_state_dict, there is no engine file to build.

Contract of the seam, stated because raw addresses carry none of it:
  * fp32 only: the six inputs and the output are read and written as contiguous float32 (the reference's TRT profile,
    scripts/export_onnx.py:343-346, and its `x.contiguous().data_ptr()` calls); a half-precision caller must cast first.
  * stream order: the reference fills x / mask / mu / t on the caller's current stream and then launches on the pool's own
    non-blocking stream with nothing ordering the two (a latent race of the reference's seam).  Here `acquire_estimator()`
    records an event on the caller's current stream and makes the pool stream wait for it, so the estimator never reads
    an input its producer has not finished; the reference's own `current_stream().synchronize()` orders the way back.
  * one library context has one workspace: however many pool entries exist, calls are serialised -- `acquire_estimator()`
    takes a lock that `release_estimator()` returns after the reference has synchronised the stream.
"""
from __future__ import annotations

import ctypes as C
import queue
import threading
from typing import Dict, Tuple

import torch

from .. import spec
from .._lib import check

TENSOR_NAMES = ("x", "mask", "mu", "t", "spks", "cond", "estimator_out")      # binding order of scripts/export_onnx.py


class HipEstimatorEngine:
    """what forward_estimator uses of a tensorrt.ICudaEngine: binding names by index"""

    num_io_tensors = len(TENSOR_NAMES)

    def get_tensor_name(self, index: int) -> str:
        return TENSOR_NAMES[index]


class HipEstimatorContext:
    """what forward_estimator uses of a tensorrt.IExecutionContext"""

    def __init__(self, engine):
        self._engine = engine
        self._shape: Dict[str, Tuple[int, ...]] = {}
        self._addr: Dict[str, int] = {}

    def set_input_shape(self, name: str, shape) -> bool:
        if name not in TENSOR_NAMES[:6]:
            return False
        self._shape[name] = tuple(int(v) for v in shape)
        return True

    def set_tensor_address(self, name: str, ptr: int) -> bool:
        if name not in TENSOR_NAMES:
            return False
        self._addr[name] = int(ptr)
        return True

    def execute_async_v3(self, stream_handle: int) -> bool:
        missing = [n for n in TENSOR_NAMES if n not in self._addr] + [n for n in TENSOR_NAMES[:6] if n not in self._shape]
        if missing:
            raise RuntimeError(f"HipEstimatorContext: unbound tensors {missing}")
        b2, c, t = self._shape["x"]
        want = {"x": (b2, spec.N_FEATS, t), "mask": (b2, 1, t), "mu": (b2, spec.N_FEATS, t), "t": (b2,),
                "spks": (b2, spec.N_FEATS), "cond": (b2, spec.N_FEATS, t)}
        if c != spec.N_FEATS or any(self._shape[k] != v for k, v in want.items()):
            raise RuntimeError(f"HipEstimatorContext: shapes {self._shape} do not describe one estimator call")
        a = self._addr
        p = lambda n: C.c_void_p(a[n])
        eng = self._engine
        check(eng.lib.jv_flow_estimator_masked(eng._h, p("x"), p("mask"), p("mu"), p("t"), p("spks"), p("cond"), b2, t,
                                               p("estimator_out"), C.c_void_p(int(stream_handle))))
        return True


class HipEstimator:
    """`TrtContextWrapper`-shaped pool (utils/common.py:219-238): `trt_concurrent` contexts, each with its own torch stream
    context manager.  All of them drive ONE library context (one workspace), so the pool hands out one at a time."""

    def __init__(self, engine, trt_concurrent: int = 1, device=None):
        if trt_concurrent < 1:
            raise ValueError("trt_concurrent must be >= 1")
        self.engine = engine
        self.trt_engine = HipEstimatorEngine()
        self.device = engine.device if device is None else torch.device(device)
        self.trt_context_pool: "queue.Queue" = queue.Queue(maxsize=trt_concurrent)
        self._busy = threading.Lock()
        for _ in range(trt_concurrent):
            stream = torch.cuda.Stream(self.device)
            ctx = torch.cuda.stream(stream)
            ctx.jv_stream = stream          # the raw stream behind the context manager, for the event wait below
            self.trt_context_pool.put([HipEstimatorContext(engine), ctx])

    def acquire_estimator(self):
        self._busy.acquire()                # released by release_estimator(), i.e. after the caller synchronised the stream
        try:
            entry = self.trt_context_pool.get()
            # order the pool stream behind whatever produced the inputs on the caller's stream
            entry[1].jv_stream.wait_event(torch.cuda.current_stream(self.device).record_event())
        except BaseException:
            self._busy.release()
            raise
        return entry, self.trt_engine

    def release_estimator(self, context, stream):
        self.trt_context_pool.put([context, stream])
        self._busy.release()
