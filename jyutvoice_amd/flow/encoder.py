"""`FlowEncoder` drop-in (the class the reference defines in infer.py:35-83 for the prompt / voice-cloning branch):
speech tokens -> `prompt_h`, the hidden sequence `JyutVoiceTTS.synthesise` prepends to `mu` (jyutvoice_tts.py:213-225).

Same constructor keywords, state-dict key names (`input_embedding.weight`, `encoder.*` of UpsampleConformerEncoder,
`encoder_proj.*`) and `forward(token, token_len) -> (h [B, 2*Tk, 80], h_lengths)`.  All arithmetic runs in
libjyutvoice_hip.so (jv_prompt_encoder_fwd); the only host-side arithmetic is the positional encoding's 256 frequencies,
computed with the reference's own torch expression (jyutvoice/transformer/embedding.py:239-242) and handed to the library."""
from __future__ import annotations

import math
from typing import Dict

import torch

from .. import spec
from ..engine import JV_MODEL_PROMPT
from ..runtime import get_runtime

DIV_TERM_KEY = "pos_enc.div_term"      # not a state-dict entry of the reference: `pe` is a plain attribute there


def div_term() -> torch.Tensor:
    d = spec.PROMPT_DIM
    return torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))


class FlowEncoder:
    def __init__(self, vocab_size: int = 6561, input_size: int = 512, output_size: int = 80, device="cuda:0"):
        got, want = (vocab_size, input_size, output_size), (spec.PROMPT_VOCAB, spec.PROMPT_DIM, spec.N_FEATS)
        if got != want:
            raise NotImplementedError(f"libjyutvoice_hip is built for the CosyVoice2 flow encoder {want}; got {got}")
        self.device = torch.device(device)
        self._loaded = False

    def to(self, device):
        self.device = torch.device(device)
        return self

    def eval(self):
        return self

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in spec.PROMPT_INVENTORY if k not in state_dict]
        unexpected = [k for k in state_dict if k not in spec.PROMPT_INVENTORY]
        if missing or (strict and unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for FlowEncoder: Missing key(s): {missing[:6]}; "
                               f"Unexpected key(s): {unexpected[:6]}")
        for k, shape in spec.PROMPT_INVENTORY.items():
            if tuple(state_dict[k].shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(state_dict[k].shape)} from "
                                   f"checkpoint, the shape in current model is {tuple(shape)}.")
        sd = {k: state_dict[k] for k in spec.PROMPT_INVENTORY}
        sd[DIV_TERM_KEY] = div_term()
        get_runtime(self.device).set_weights(JV_MODEL_PROMPT, sd)
        self._loaded = True
        return missing, unexpected

    @torch.inference_mode()
    def forward(self, token: torch.Tensor, token_len: torch.Tensor):
        """infer.py:66-83 -> (h [B, 2*Tk, 80], h_lengths [B]); rows beyond 2*token_len[b] are zero"""
        if not self._loaded:
            raise RuntimeError("FlowEncoder: load_state_dict() has not been called")
        B, Tk = token.shape
        eng = get_runtime(self.device).ensure(B, 2 * Tk, 1)
        h = eng.prompt_encoder(token, token_len)
        return h, token_len.to(self.device) * spec.PROMPT_UP_STRIDE

    __call__ = forward


def extract_flow_weights(state_dict: Dict[str, torch.Tensor]):
    """Split a CosyVoice2 `flow.pt` state-dict the way scripts/download_pretrain_weights.py:168-214 does:
    -> (flow-encoder part: `encoder.*`, `input_embedding.*`, `encoder_proj.*`;  decoder part: `decoder.*`,
    `spk_embed_affine_layer.*`).  The first loads into `FlowEncoder`, the second into `JyutVoiceTTS.load_pretrain`."""
    enc = {k: v for k, v in state_dict.items() if k.startswith(("encoder.", "input_embedding.", "encoder_proj."))}
    dec = {k: v for k, v in state_dict.items() if k.startswith(("decoder.", "spk_embed_affine_layer."))}
    return enc, dec


def load_flow_encoder(flow_encoder_path, device="cuda:0"):
    """infer.py:209-230: `FlowEncoder` with the weights of `flow_encoder.pt`, or None when no path is given"""
    if flow_encoder_path is None:
        return None
    enc = FlowEncoder(device=device)
    enc.load_state_dict(torch.load(flow_encoder_path, map_location="cpu", weights_only=True))
    return enc.eval()
