"""Configuration holder mirroring `jyutvoice.flow.flow_matching.CausalConditionalCFM` (flow_matching.py:343-401).

Unlike the reference constructor this one does not reseed the global torch / numpy / python RNGs
(flow_matching.py:353): the fixed noise tensor is drawn from a private seed-0 generator (synth.rand_noise)."""
import torch

from .. import spec


def _get(params, name):
    return params[name] if isinstance(params, dict) else getattr(params, name)


class CausalConditionalCFM:
    def __init__(self, in_channels, cfm_params, n_spks=1, spk_emb_dim=64, estimator=None):
        self.t_scheduler = _get(cfm_params, "t_scheduler")
        self.inference_cfg_rate = _get(cfm_params, "inference_cfg_rate")
        if self.t_scheduler != "cosine" or abs(self.inference_cfg_rate - spec.CFG_RATE) > 1e-12 or spk_emb_dim != spec.N_FEATS:
            raise NotImplementedError("libjyutvoice_hip is built for t_scheduler='cosine', inference_cfg_rate=0.7, "
                                      "spk_emb_dim=80 (configs/base.yaml:76-87)")
        self.estimator = estimator
        self.device = torch.device("cuda:0")      # JyutVoiceTTS sets it to its own device

    @torch.inference_mode()
    def forward(self, mu, mask, n_timesteps, temperature=1.0, spks=None, cond=None, streaming=False):
        """flow_matching.py:356-401 -> (mel [B,80,T], None): fixed noise prefix * temperature, cosine schedule, the Euler/CFG
        loop (solve_euler, :215-265) -- all of it inside jv_cfm_solve.  streaming=True: the estimator's chunk-causal
        attention (decoder.py:951-954) with this decoder's static_chunk_size, so frames of finished chunks do not change
        when more of the utterance arrives.  B > 1 is the batched extension (mask rows = per-utterance lengths)."""
        from ..runtime import get_runtime
        B, _, T = mu.shape
        eng = get_runtime(self.device).ensure(B, T, 1)
        chunk = getattr(self.estimator, "static_chunk_size", spec.EST_STATIC_CHUNK) if streaming else 0
        eng.set_streaming(chunk)
        try:
            lens = None if mask is None else mask.reshape(B, -1).ne(0).sum(dim=1)
            if cond is None:
                cond = torch.zeros_like(mu)
            t_span = 1 - torch.cos(torch.linspace(0, 1, n_timesteps + 1) * 0.5 * torch.pi)
            mel = eng.cfm_solve(mu, lens, spks, cond, n_timesteps, temperature, t_span=t_span)
        finally:
            eng.set_streaming(0)
        return mel, None

    __call__ = forward
