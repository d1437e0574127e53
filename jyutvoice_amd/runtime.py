"""Per-GPU runtime shared by the mirror classes: one library context holding both models' packed weights and
a workspace that grows on demand.  Growth (jv_reserve) re-creates the activation buffers only -- weights are uploaded and
packed once per load_state_dict -- but it drains the device and reallocates, so a service should call
`get_runtime(dev).ensure(max_batch, max_frames, max_tokens)` once up front, as bench.py does."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import synth
from .engine import JV_MODEL_HIFT, JV_MODEL_TTS, Engine


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class Runtime:
    def __init__(self, device):
        self.device = torch.device(device)
        self.engine: Optional[Engine] = None
        self.caps = (0, 0, 0)
        self.sds: Dict[int, Dict[str, torch.Tensor]] = {}
        self.noise = None

    def set_weights(self, model: int, sd: Dict[str, torch.Tensor]):
        """validate + upload a state-dict (raises like nn.Module.load_state_dict)"""
        fresh = model not in self.sds
        self.sds[model] = {k: v.detach() for k, v in sd.items()}
        if self.engine is None:
            self.ensure(1, 512, 256)
        elif fresh:
            self.engine.load_state_dict(model, self.sds[model], strict=True)      # the other models' weights stay in place
        else:
            self._rebuild(self.caps)      # re-loading a finalized model: the C ABI wants a new context

    def ensure(self, batch: int, frames: int, tokens: int) -> Engine:
        need = (batch, frames, tokens)
        if self.engine is not None and all(n <= c for n, c in zip(need, self.caps)):
            return self.engine
        caps = (max(batch, self.caps[0]), _round_up(max(frames, self.caps[1]), 64), _round_up(max(tokens, self.caps[2]), 32))
        if self.engine is None:
            self._rebuild(caps)
        else:
            # workspace only: the weights of both models stay where they are.  jv_reserve is failure-atomic: if the larger
            # workspace does not fit it restores the old one and the error propagates with self.caps unchanged; if even
            # that fails the context refuses every call (JV_ERR_STATE), so it is dropped here and rebuilt on the next use.
            try:
                self.engine.reserve(*caps)
            except Exception:
                if self.engine.broken():
                    self.engine.close()
                    self.engine = None
                    self.caps = (0, 0, 0)
                raise
            self.caps = caps
        return self.engine

    def _rebuild(self, caps):
        """new context + upload of every known state-dict: only when WEIGHTS change (the C ABI finalizes a model once)"""
        if self.engine is not None:
            self.engine.close()
            self.engine = None
            torch.cuda.empty_cache()
        eng = Engine(self.device, max_batch=caps[0], max_frames=caps[1], max_tokens=caps[2])
        for model, sd in self.sds.items():
            eng.load_state_dict(model, sd, strict=True)
        if self.noise is None:
            self.noise = synth.rand_noise()
        eng.load_noise(self.noise)
        self.engine = eng
        self.caps = caps


_runtimes: Dict[str, Runtime] = {}


def get_runtime(device="cuda:0") -> Runtime:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("jyutvoice_amd runs on an AMD GPU only; there is no CPU path")
    key = f"cuda:{dev.index if dev.index is not None else torch.cuda.current_device()}"
    if key not in _runtimes:
        _runtimes[key] = Runtime(key)
    return _runtimes[key]
