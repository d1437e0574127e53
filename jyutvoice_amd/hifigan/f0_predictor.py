"""Configuration holder mirroring `jyutvoice.hifigan.f0_predictor.ConvRNNF0Predictor` (f0_predictor.py:8-50)."""
from .. import spec


class ConvRNNF0Predictor:
    def __init__(self, num_class: int = 1, in_channels: int = 80, cond_channels: int = 512):
        if (num_class, in_channels, cond_channels) != (1, spec.N_FEATS, spec.HIFT_F0_CH):
            raise NotImplementedError("libjyutvoice_hip is built for ConvRNNF0Predictor(1, 80, 512)")
        self.num_class = num_class
