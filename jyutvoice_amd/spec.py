"""Frozen constants and the state-dict inventory of the JyutVoice hot path.

The HIP library is specialised to one architecture: the one `configs/base.yaml:1-110` of the
reference instantiates.  This module is the single host-side statement of that architecture:
hyper-parameters, and for every checkpoint tensor its name and shape.  `load_state_dict` on the
mirror classes validates against it, `synth.py` fills it with synthetic weights, and the C-ABI
library's own registry (csrc/registry.hip) must agree with it (tests/test_abi.py checks that).

Key namespaces follow the reference checkpoints:
  tts:  encoder.* (117)  dp.* (12)  decoder.estimator.* (910)  spk_embed_affine_layer.* (2)
  hift: 328 keys, weight-norm in two spellings (parametrizations.weight.original0/1 in
        jyutvoice/hifigan/generator.py:26, weight_g/weight_v in jyutvoice/hifigan/f0_predictor.py:16)
"""
from __future__ import annotations

from collections import OrderedDict

# ---- configs/base.yaml:1-24 -------------------------------------------------------------------
N_FEATS = 80
SAMPLE_RATE = 24000
HOP_LENGTH = 480
SPK_EMBED_DIM = 192

# ---- text encoder (configs/base.yaml:51-67, text_encoder.py:340-404) ---------------------------
ENC_N_VOCAB = 97
ENC_N_LANG = 4
ENC_N_TONE = 7
ENC_N_WORD_POS = 4
ENC_N_SYL_POS = 4
ENC_CH = 192
ENC_HIDDEN = 576           # n_channels*2 + gin_channels
ENC_FILTER = 768
ENC_HEADS = 2
ENC_HEAD_DIM = 288
ENC_ROPE_DIM = 144         # int(288*0.5), text_encoder.py:203
ENC_LAYERS = 6
ENC_KERNEL = 3
ENC_PRENET_KERNEL = 5
ENC_PRENET_LAYERS = 3
ENC_LN_EPS = 1e-4

# ---- duration predictor (configs/base.yaml:69-74) ----------------------------------------------
DP_FILTER = 256
DP_KERNEL = 3

# ---- CFM + estimator (configs/base.yaml:76-99) -------------------------------------------------
CFG_RATE = 0.7
NOISE_FRAMES = 50 * 300    # flow_matching.py:354
EST_IN = 320
EST_CH = 256
EST_TIME_DIM = 1024
EST_HEADS = 8
EST_HEAD_DIM = 64
EST_INNER = EST_HEADS * EST_HEAD_DIM  # 512
EST_FF = 1024
EST_N_BLOCKS = 4
EST_N_MID = 12
EST_LN_EPS = 1e-5
EST_STATIC_CHUNK = 50      # chunk_size * token_mel_ratio (streaming only)

# ---- HiFT (configs/base.yaml:26-48) -------------------------------------------------------------
HIFT_BASE_CH = 512
HIFT_NB_HARMONICS = 8
HIFT_NSF_ALPHA = 0.1
HIFT_NSF_SIGMA = 0.003
HIFT_VOICED_THRESHOLD = 10.0
HIFT_UP_RATES = (8, 5, 3)
HIFT_UP_KERNELS = (16, 11, 7)
HIFT_NFFT = 16
HIFT_HOP = 4
HIFT_RB_KERNELS = (3, 7, 11)
HIFT_RB_DILATIONS = (1, 3, 5)
HIFT_SRC_RB_KERNELS = (7, 7, 11)
HIFT_LRELU_SLOPE = 0.1
HIFT_AUDIO_LIMIT = 0.99
HIFT_F0_CH = 512
# source_downs: (kernel, stride, padding) for the three fusion points (generator.py:308-332)
HIFT_SRC_DOWNS = ((30, 15, 7), (6, 3, 1), (1, 1, 0))
HIFT_UPSAMPLE_TOTAL = 480  # 8*5*3*4


def _tts_inventory() -> "OrderedDict[str, tuple]":
    inv: "OrderedDict[str, tuple]" = OrderedDict()

    # encoder.* ------------------------------------------------------------------------------
    e = "encoder."
    inv[e + "emb.weight"] = (ENC_N_VOCAB, ENC_CH)
    inv[e + "lang_emb.weight"] = (ENC_N_LANG, ENC_CH)
    inv[e + "tone_emb.weight"] = (ENC_N_TONE, ENC_CH)
    inv[e + "word_pos_emb.weight"] = (ENC_N_WORD_POS, ENC_CH)
    inv[e + "syllable_pos.weight"] = (ENC_N_SYL_POS, ENC_CH)
    for i in range(ENC_PRENET_LAYERS):
        inv[e + f"prenet.conv_layers.{i}.weight"] = (ENC_CH, ENC_CH, ENC_PRENET_KERNEL)
        inv[e + f"prenet.conv_layers.{i}.bias"] = (ENC_CH,)
        inv[e + f"prenet.norm_layers.{i}.gamma"] = (ENC_CH,)
        inv[e + f"prenet.norm_layers.{i}.beta"] = (ENC_CH,)
    inv[e + "prenet.proj.weight"] = (ENC_CH, ENC_CH, 1)
    inv[e + "prenet.proj.bias"] = (ENC_CH,)
    for i in range(ENC_LAYERS):
        for n in "qkvo":
            inv[e + f"encoder.attn_layers.{i}.conv_{n}.weight"] = (ENC_HIDDEN, ENC_HIDDEN, 1)
            inv[e + f"encoder.attn_layers.{i}.conv_{n}.bias"] = (ENC_HIDDEN,)
        inv[e + f"encoder.norm_layers_1.{i}.gamma"] = (ENC_HIDDEN,)
        inv[e + f"encoder.norm_layers_1.{i}.beta"] = (ENC_HIDDEN,)
        inv[e + f"encoder.ffn_layers.{i}.conv_1.weight"] = (ENC_FILTER, ENC_HIDDEN, ENC_KERNEL)
        inv[e + f"encoder.ffn_layers.{i}.conv_1.bias"] = (ENC_FILTER,)
        inv[e + f"encoder.ffn_layers.{i}.conv_2.weight"] = (ENC_HIDDEN, ENC_FILTER, ENC_KERNEL)
        inv[e + f"encoder.ffn_layers.{i}.conv_2.bias"] = (ENC_HIDDEN,)
        inv[e + f"encoder.norm_layers_2.{i}.gamma"] = (ENC_HIDDEN,)
        inv[e + f"encoder.norm_layers_2.{i}.beta"] = (ENC_HIDDEN,)
    inv[e + "proj.weight"] = (N_FEATS, ENC_HIDDEN, 1)
    inv[e + "proj.bias"] = (N_FEATS,)

    # dp.* -----------------------------------------------------------------------------------
    d = "dp."
    inv[d + "conv_1.weight"] = (DP_FILTER, ENC_HIDDEN, DP_KERNEL)
    inv[d + "conv_1.bias"] = (DP_FILTER,)
    inv[d + "norm_1.gamma"] = (DP_FILTER,)
    inv[d + "norm_1.beta"] = (DP_FILTER,)
    inv[d + "conv_2.weight"] = (DP_FILTER, DP_FILTER, DP_KERNEL)
    inv[d + "conv_2.bias"] = (DP_FILTER,)
    inv[d + "norm_2.gamma"] = (DP_FILTER,)
    inv[d + "norm_2.beta"] = (DP_FILTER,)
    inv[d + "proj.weight"] = (1, DP_FILTER, 1)
    inv[d + "proj.bias"] = (1,)
    inv[d + "cond.weight"] = (ENC_HIDDEN, SPK_EMBED_DIM, 1)
    inv[d + "cond.bias"] = (ENC_HIDDEN,)

    # decoder.estimator.* --------------------------------------------------------------------
    p = "decoder.estimator."
    inv[p + "time_mlp.linear_1.weight"] = (EST_TIME_DIM, EST_IN)
    inv[p + "time_mlp.linear_1.bias"] = (EST_TIME_DIM,)
    inv[p + "time_mlp.linear_2.weight"] = (EST_TIME_DIM, EST_TIME_DIM)
    inv[p + "time_mlp.linear_2.bias"] = (EST_TIME_DIM,)

    def resnet(prefix: str, cin: int) -> None:
        inv[prefix + "mlp.1.weight"] = (EST_CH, EST_TIME_DIM)
        inv[prefix + "mlp.1.bias"] = (EST_CH,)
        inv[prefix + "block1.block.0.weight"] = (EST_CH, cin, 3)
        inv[prefix + "block1.block.0.bias"] = (EST_CH,)
        inv[prefix + "block1.block.2.weight"] = (EST_CH,)
        inv[prefix + "block1.block.2.bias"] = (EST_CH,)
        inv[prefix + "block2.block.0.weight"] = (EST_CH, EST_CH, 3)
        inv[prefix + "block2.block.0.bias"] = (EST_CH,)
        inv[prefix + "block2.block.2.weight"] = (EST_CH,)
        inv[prefix + "block2.block.2.bias"] = (EST_CH,)
        inv[prefix + "res_conv.weight"] = (EST_CH, cin, 1)
        inv[prefix + "res_conv.bias"] = (EST_CH,)

    def btb(prefix: str) -> None:
        inv[prefix + "norm1.weight"] = (EST_CH,)
        inv[prefix + "norm1.bias"] = (EST_CH,)
        inv[prefix + "attn1.to_q.weight"] = (EST_INNER, EST_CH)
        inv[prefix + "attn1.to_k.weight"] = (EST_INNER, EST_CH)
        inv[prefix + "attn1.to_v.weight"] = (EST_INNER, EST_CH)
        inv[prefix + "attn1.to_out.0.weight"] = (EST_CH, EST_INNER)
        inv[prefix + "attn1.to_out.0.bias"] = (EST_CH,)
        inv[prefix + "norm3.weight"] = (EST_CH,)
        inv[prefix + "norm3.bias"] = (EST_CH,)
        inv[prefix + "ff.net.0.proj.weight"] = (EST_FF, EST_CH)
        inv[prefix + "ff.net.0.proj.bias"] = (EST_FF,)
        inv[prefix + "ff.net.2.weight"] = (EST_CH, EST_FF)
        inv[prefix + "ff.net.2.bias"] = (EST_CH,)

    resnet(p + "down_blocks.0.0.", EST_IN)
    for j in range(EST_N_BLOCKS):
        btb(p + f"down_blocks.0.1.{j}.")
    inv[p + "down_blocks.0.2.weight"] = (EST_CH, EST_CH, 3)
    inv[p + "down_blocks.0.2.bias"] = (EST_CH,)
    for i in range(EST_N_MID):
        resnet(p + f"mid_blocks.{i}.0.", EST_CH)
        for j in range(EST_N_BLOCKS):
            btb(p + f"mid_blocks.{i}.1.{j}.")
    resnet(p + "up_blocks.0.0.", 2 * EST_CH)
    for j in range(EST_N_BLOCKS):
        btb(p + f"up_blocks.0.1.{j}.")
    inv[p + "up_blocks.0.2.weight"] = (EST_CH, EST_CH, 3)
    inv[p + "up_blocks.0.2.bias"] = (EST_CH,)
    inv[p + "final_block.block.0.weight"] = (EST_CH, EST_CH, 3)
    inv[p + "final_block.block.0.bias"] = (EST_CH,)
    inv[p + "final_block.block.2.weight"] = (EST_CH,)
    inv[p + "final_block.block.2.bias"] = (EST_CH,)
    inv[p + "final_proj.weight"] = (N_FEATS, EST_CH, 1)
    inv[p + "final_proj.bias"] = (N_FEATS,)

    inv["spk_embed_affine_layer.weight"] = (N_FEATS, SPK_EMBED_DIM)
    inv["spk_embed_affine_layer.bias"] = (N_FEATS,)
    return inv


def _hift_inventory() -> "OrderedDict[str, tuple]":
    inv: "OrderedDict[str, tuple]" = OrderedDict()

    def wn(prefix: str, shape: tuple) -> None:
        # torch.nn.utils.parametrizations.weight_norm, dim=0 (generator.py:26)
        inv[prefix + "bias"] = (shape[0],)
        inv[prefix + "parametrizations.weight.original0"] = (shape[0], 1, 1)
        inv[prefix + "parametrizations.weight.original1"] = shape

    inv["m_source.l_linear.weight"] = (1, HIFT_NB_HARMONICS + 1)
    inv["m_source.l_linear.bias"] = (1,)
    wn("conv_pre.", (HIFT_BASE_CH, N_FEATS, 7))
    for i, k in enumerate(HIFT_UP_KERNELS):
        cin = HIFT_BASE_CH >> i
        cout = HIFT_BASE_CH >> (i + 1)
        # ConvTranspose1d weight is [in, out, k]; weight_norm dim=0 is over the in-channel axis
        inv[f"ups.{i}.bias"] = (cout,)
        inv[f"ups.{i}.parametrizations.weight.original0"] = (cin, 1, 1)
        inv[f"ups.{i}.parametrizations.weight.original1"] = (cin, cout, k)
    for i, (k, _s, _p) in enumerate(HIFT_SRC_DOWNS):
        cout = HIFT_BASE_CH >> (i + 1)
        inv[f"source_downs.{i}.weight"] = (cout, HIFT_NFFT + 2, k)
        inv[f"source_downs.{i}.bias"] = (cout,)

    def resblock(prefix: str, ch: int, k: int) -> None:
        for which in ("convs1", "convs2"):
            for j in range(3):
                wn(f"{prefix}{which}.{j}.", (ch, ch, k))
        for which in ("activations1", "activations2"):
            for j in range(3):
                inv[f"{prefix}{which}.{j}.alpha"] = (ch,)

    for i, k in enumerate(HIFT_SRC_RB_KERNELS):
        resblock(f"source_resblocks.{i}.", HIFT_BASE_CH >> (i + 1), k)
    for i in range(3):
        for j, k in enumerate(HIFT_RB_KERNELS):
            resblock(f"resblocks.{3 * i + j}.", HIFT_BASE_CH >> (i + 1), k)
    wn("conv_post.", (HIFT_NFFT + 2, HIFT_BASE_CH >> 3, 7))
    for n in range(5):
        cin = N_FEATS if n == 0 else HIFT_F0_CH
        pre = f"f0_predictor.condnet.{2 * n}."
        # legacy torch.nn.utils.weight_norm spelling (f0_predictor.py:16)
        inv[pre + "bias"] = (HIFT_F0_CH,)
        inv[pre + "weight_g"] = (HIFT_F0_CH, 1, 1)
        inv[pre + "weight_v"] = (HIFT_F0_CH, cin, 3)
    inv["f0_predictor.classifier.weight"] = (1, HIFT_F0_CH)
    inv["f0_predictor.classifier.bias"] = (1,)
    return inv


# ---- prompt (voice-cloning) branch: FlowEncoder of infer.py:35-83 = Embedding(6561, 512) ->
# UpsampleConformerEncoder (jyutvoice/transformer/upsample_encoder.py:137-375, CosyVoice2 settings: 512-d, 8 heads,
# FFN 2048, 6 + 4 pre-LN rel-pos blocks, no conv module, no macaron, static_chunk_size 25) -> Linear(512, 80)
PROMPT_VOCAB = 6561
PROMPT_DIM = 512
PROMPT_HEADS = 8
PROMPT_FFN = 2048
PROMPT_BLOCKS = 6
PROMPT_UP_BLOCKS = 4
PROMPT_LOOKAHEAD = 3
PROMPT_UP_STRIDE = 2
PROMPT_STATIC_CHUNK = 25


def _prompt_inventory() -> "OrderedDict[str, tuple]":
    inv: "OrderedDict[str, tuple]" = OrderedDict()
    D, F = PROMPT_DIM, PROMPT_FFN
    inv["input_embedding.weight"] = (PROMPT_VOCAB, D)

    def embed(pre):
        inv[pre + "out.0.weight"] = (D, D)
        inv[pre + "out.0.bias"] = (D,)
        inv[pre + "out.1.weight"] = (D,)
        inv[pre + "out.1.bias"] = (D,)

    def block(pre):
        inv[pre + "self_attn.pos_bias_u"] = (PROMPT_HEADS, D // PROMPT_HEADS)
        inv[pre + "self_attn.pos_bias_v"] = (PROMPT_HEADS, D // PROMPT_HEADS)
        for n in ("q", "k", "v", "out"):
            inv[pre + f"self_attn.linear_{n}.weight"] = (D, D)
            inv[pre + f"self_attn.linear_{n}.bias"] = (D,)
        inv[pre + "self_attn.linear_pos.weight"] = (D, D)
        inv[pre + "feed_forward.w_1.weight"] = (F, D)
        inv[pre + "feed_forward.w_1.bias"] = (F,)
        inv[pre + "feed_forward.w_2.weight"] = (D, F)
        inv[pre + "feed_forward.w_2.bias"] = (D,)
        for n in ("norm_ff", "norm_mha"):
            inv[pre + n + ".weight"] = (D,)
            inv[pre + n + ".bias"] = (D,)

    embed("encoder.embed.")
    inv["encoder.after_norm.weight"] = (D,)
    inv["encoder.after_norm.bias"] = (D,)
    inv["encoder.pre_lookahead_layer.conv1.weight"] = (D, D, PROMPT_LOOKAHEAD + 1)
    inv["encoder.pre_lookahead_layer.conv1.bias"] = (D,)
    inv["encoder.pre_lookahead_layer.conv2.weight"] = (D, D, 3)
    inv["encoder.pre_lookahead_layer.conv2.bias"] = (D,)
    for i in range(PROMPT_BLOCKS):
        block(f"encoder.encoders.{i}.")
    inv["encoder.up_layer.conv.weight"] = (D, D, 2 * PROMPT_UP_STRIDE + 1)
    inv["encoder.up_layer.conv.bias"] = (D,)
    embed("encoder.up_embed.")
    for i in range(PROMPT_UP_BLOCKS):
        block(f"encoder.up_encoders.{i}.")
    inv["encoder_proj.weight"] = (N_FEATS, D)
    inv["encoder_proj.bias"] = (N_FEATS,)
    return inv


TTS_INVENTORY = _tts_inventory()
HIFT_INVENTORY = _hift_inventory()
PROMPT_INVENTORY = _prompt_inventory()

# known-answer structural checks (README.md:171,233 of the reference)
assert len(TTS_INVENTORY) == 117 + 12 + 910 + 2, len(TTS_INVENTORY)
assert len(HIFT_INVENTORY) == 328, len(HIFT_INVENTORY)
assert len(PROMPT_INVENTORY) == 1 + 206 + 2, len(PROMPT_INVENTORY)   # the instantiated reference encoder has 206 tensors


def numel(shape: tuple) -> int:
    n = 1
    for s in shape:
        n *= s
    return n


EST_PARAMS = sum(numel(s) for k, s in TTS_INVENTORY.items() if k.startswith("decoder.estimator."))
assert EST_PARAMS == 71_302_480, EST_PARAMS
