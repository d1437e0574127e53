"""jyutvoice_amd: MI355X-native (gfx950) implementation of JyutVoice's synthesise() + HiFT hot path.

Import surface mirrors the reference's dotted paths (configs/base.yaml `!new:` constructors):
    jyutvoice_amd.models.jyutvoice_tts.JyutVoiceTTS      jyutvoice_amd.hifigan.generator.HiFTGenerator
    jyutvoice_amd.models.text_encoder.TextEncoder        jyutvoice_amd.hifigan.f0_predictor.ConvRNNF0Predictor
    jyutvoice_amd.models.duration_predictor.DurationPredictor
    jyutvoice_amd.flow.flow_matching.CausalConditionalCFM jyutvoice_amd.flow.decoder.CausalConditionalDecoder
The numerical work is in libjyutvoice_hip.so (hand-written HIP, C ABI in include/jyutvoice_hip.h); importing this
package does not load it, using any model does, and fails loudly if it is missing.
"""
__version__ = "0.1.0"


def build_default(device="cuda:0"):
    """No-YAML factory with the base.yaml constants baked in -> (tts, hift), weights not yet loaded."""
    from types import SimpleNamespace as NS

    from .flow.decoder import CausalConditionalDecoder
    from .flow.flow_matching import CausalConditionalCFM
    from .hifigan.f0_predictor import ConvRNNF0Predictor
    from .hifigan.generator import HiFTGenerator
    from .models.duration_predictor import DurationPredictor
    from .models.jyutvoice_tts import JyutVoiceTTS
    from .models.text_encoder import TextEncoder

    enc = TextEncoder("RoPE Encoder", NS(n_feats=80, n_channels=192, filter_channels=768, filter_channels_dp=256, n_heads=2,
                                         n_layers=6, kernel_size=3, p_dropout=0.1, gin_channels=192, prenet=True),
                      n_vocab=97, n_lang=4, n_tone=7)
    dp = DurationPredictor(576, 256, 3, 0.1, 192)
    est = CausalConditionalDecoder(in_channels=320, out_channels=80, channels=[256], dropout=0.0, attention_head_dim=64,
                                   n_blocks=4, num_mid_blocks=12, num_heads=8, act_fn="gelu", static_chunk_size=50,
                                   num_decoding_left_chunks=-1)
    cfm = CausalConditionalCFM(in_channels=240, n_spks=1, spk_emb_dim=80,
                               cfm_params=NS(sigma_min=1e-6, solver="euler", t_scheduler="cosine", training_cfg_rate=0.2,
                                             inference_cfg_rate=0.7, reg_loss_type="l1"), estimator=est)
    tts = JyutVoiceTTS(enc, cfm, dp, output_size=80, spk_embed_dim=192, device=device)
    hift = HiFTGenerator(in_channels=80, base_channels=512, nb_harmonics=8, sampling_rate=24000, nsf_alpha=0.1,
                         nsf_sigma=0.003, nsf_voiced_threshold=10, upsample_rates=[8, 5, 3],
                         upsample_kernel_sizes=[16, 11, 7], istft_params={"n_fft": 16, "hop_len": 4},
                         resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
                         source_resblock_kernel_sizes=[7, 7, 11], source_resblock_dilation_sizes=[[1, 3, 5]] * 3,
                         lrelu_slope=0.1, audio_limit=0.99, f0_predictor=ConvRNNF0Predictor(1, 80, 512), device=device)
    return tts, hift
