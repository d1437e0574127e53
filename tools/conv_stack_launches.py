"""which conv-stack kernels does a 32-utterance solve launch?  (the built-in profiler around one warm solve: rowres_h3 = whole resnets in one launch; JV_NO_RES_PAIR=1 shows the two-launch form)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jyutvoice_amd
from jyutvoice_amd import synth, engine

sd = synth.tts_state_dict(fixed_duration=1.5)
keys = ("x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos", "spk_embed")
b = synth.batch(32, 150)
tts, _ = jyutvoice_amd.build_default("cuda:0")
tts.load_state_dict(sd)
tts.synthesise(*[b[k] for k in keys], None, n_timesteps=2, batched=True)
engine.profile_enable(True)
mel = tts.synthesise(*[b[k] for k in keys], None, n_timesteps=2, batched=True)["mel"]
rep = engine.profile_report()
engine.profile_enable(False)
for k, v in sorted(rep.items()):
    if "row" in k or "conv" in k:
        print(k, v)
print("finite", bool(torch.isfinite(mel).all()))
