"""Configuration holder mirroring `jyutvoice.models.text_encoder.TextEncoder` (text_encoder.py:340-404).

The arithmetic lives in libjyutvoice_hip.so (jv_encoder_fwd), which is specialised to the architecture of
configs/base.yaml:51-67; this class only checks that the YAML asks for that architecture, so that editing
the dotted path `jyutvoice.` -> `jyutvoice_amd.` is the whole integration."""
from .. import spec


def _get(params, name):
    return params[name] if isinstance(params, dict) else getattr(params, name)


class TextEncoder:
    def __init__(self, encoder_type, encoder_params, n_vocab, n_lang, n_tone=7):
        self.encoder_type = encoder_type
        self.n_vocab = n_vocab
        self.n_feats = _get(encoder_params, "n_feats")
        self.n_channels = _get(encoder_params, "n_channels")
        self.hidden_channels = self.n_channels * 2 + _get(encoder_params, "gin_channels")
        want = dict(n_feats=spec.N_FEATS, n_channels=spec.ENC_CH, filter_channels=spec.ENC_FILTER, n_heads=spec.ENC_HEADS,
                    n_layers=spec.ENC_LAYERS, kernel_size=spec.ENC_KERNEL, gin_channels=spec.SPK_EMBED_DIM, prenet=True)
        got = {k: _get(encoder_params, k) for k in want}
        if got != want or (n_vocab, n_lang, n_tone) != (spec.ENC_N_VOCAB, spec.ENC_N_LANG, spec.ENC_N_TONE):
            raise NotImplementedError(f"libjyutvoice_hip is built for the base.yaml text encoder {want}, "
                                      f"n_vocab/n_lang/n_tone=(97,4,7); got {got}, {(n_vocab, n_lang, n_tone)}")

    def output_size(self):
        return self.hidden_channels
