"""Configuration holder mirroring `jyutvoice.flow.flow_matching.CausalConditionalCFM` (flow_matching.py:343-401).

Unlike the reference constructor this one does not reseed the global torch / numpy / python RNGs
(flow_matching.py:353): the fixed noise tensor is drawn from a private seed-0 generator (synth.rand_noise)."""
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
