"""Configuration holder mirroring `jyutvoice.flow.decoder.CausalConditionalDecoder` (decoder.py:798-915)."""
from .. import spec


class CausalConditionalDecoder:
    def __init__(self, in_channels, out_channels, channels=(256, 256), dropout=0.05, attention_head_dim=64, n_blocks=1,
                 num_mid_blocks=2, num_heads=4, act_fn="snake", static_chunk_size=50, num_decoding_left_chunks=2):
        got = (in_channels, out_channels, tuple(channels), attention_head_dim, n_blocks, num_mid_blocks, num_heads, act_fn)
        want = (spec.EST_IN, spec.N_FEATS, (spec.EST_CH,), 64, spec.EST_N_BLOCKS, spec.EST_N_MID, spec.EST_HEADS, "gelu")
        if got != want:
            raise NotImplementedError(f"libjyutvoice_hip is built for the base.yaml estimator {want}; got {got}")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.static_chunk_size, self.num_decoding_left_chunks = static_chunk_size, num_decoding_left_chunks
