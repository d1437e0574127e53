"""Configuration holder mirroring `jyutvoice.models.duration_predictor.DurationPredictor` (duration_predictor.py:26-46)."""
from .. import spec


class DurationPredictor:
    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, gin_channels):
        got = (in_channels, filter_channels, kernel_size, gin_channels)
        want = (spec.ENC_HIDDEN, spec.DP_FILTER, spec.DP_KERNEL, spec.SPK_EMBED_DIM)
        if got != want:
            raise NotImplementedError(f"libjyutvoice_hip is built for DurationPredictor{want}; got {got}")
        self.in_channels, self.filter_channels, self.p_dropout = in_channels, filter_channels, p_dropout
