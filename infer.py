#!/usr/bin/env python3
"""`infer.py`-compatible CLI for the MI355X-native synthesis path.

Flag names and defaults follow the reference CLI (infer.py:272-329 of indiejoseph/JyutVoice): --text/--lang/--phone/
--ref_audio/--output/--config/--tts_checkpoint/--flow_encoder/--speech_tokenizer/--campplus/--hift/--n_timesteps/
--length_scale (default 0.9).  What is in scope here is the call sequence of infer.py:341-351 and :419-441 --
load the two state-dicts, `tts.synthesise(...)`, `hift.inference(mel)`, write a 24 kHz wav.

The reference's front-ends are NOT part of this build: text -> ids needs its G2P stack
(pycantonese / pypinyin / g2p_en), and --ref_audio needs two external ONNX models (speech tokenizer, speaker embedding) and a
mel extractor.  Instead this CLI takes their outputs directly:

    --tokens tokens.json     {"x": [...], "lang": [...], "tone": [...], "word_pos": [...], "syllable_pos": [...]}
                             (equal-length int lists = the output contract of jyutvoice/text/__init__.py:20-35 after
                             `intersperse`; with "interspersed": false the raw text_to_sequence lists, which get their
                             blanks here -- jyutvoice_amd/utils/text.py validates either form), optionally
                             "spk_embed": [192 floats], and for voice cloning
                             "prompt_token": [speech-token ids] + either "prompt_feat": [[80 floats] per frame] or
                             "prompt_wav_24k": "ref_24k.wav" (16-bit mono; its mel is extracted on the GPU as
                             infer.py:386 does) -- what infer.py:386-392 gets from --ref_audio; the prompt encoder
                             (--flow_encoder) then runs on the GPU exactly as infer.py:390-392 runs it
    --synthetic N            no checkpoint / no tokens: N synthetic tokens, key-hashed weights (smoke / demo)
    --synthetic-prompt K     with --synthetic: also a synthetic K-token voice prompt through the prompt encoder

With --text and no --tokens it explains what is missing instead of guessing.
"""
import argparse
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def write_wav(path: str, wav, sample_rate: int = 24000) -> None:
    """16-bit PCM mono wav (torchaudio is not available in this image; infer.py:441 uses torchaudio.save)"""
    import torch
    pcm = (wav.detach().cpu().flatten().clamp(-1, 1) * 32767.0).round().to(torch.int16).numpy().tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " +
                struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16) + b"data" + struct.pack("<I", len(pcm)))
        f.write(pcm)


def read_wav_24k(path: str):
    """16-bit PCM mono 24 kHz wav -> [1, n] float in [-1, 1] (torchaudio.load + resampling of infer.py:368-382 are front-end)"""
    import torch
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise SystemExit(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHIIHH", data[pos + 8:pos + 24])
        elif tag == b"data":
            pcm = data[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None or fmt[0] != 1 or fmt[1] != 1 or fmt[2] != 24000 or fmt[5] != 16:
        raise SystemExit(f"{path}: need 16-bit PCM, mono, 24 kHz (resample with the reference's front-end first)")
    return (torch.frombuffer(bytearray(pcm), dtype=torch.int16).float() / 32768.0).unsqueeze(0)


def main(argv=None):
    p = argparse.ArgumentParser(description="JyutVoice TTS inference on MI355X (jyutvoice_amd)")
    p.add_argument("--text", default=None, help="Text to synthesize (needs the reference's G2P front-end; see --tokens)")
    p.add_argument("--lang", default=None, choices=["en", "zh", "yue", "multilingual"], help="Language of the text")
    p.add_argument("--phone", default=None, help="Phonetic transcription (for Cantonese, optional)")
    p.add_argument("--ref_audio", default=None, help="Reference audio (needs the reference's ONNX front-ends; see --tokens)")
    p.add_argument("--output", required=True, help="Output audio file path")
    p.add_argument("--config", default="configs/base.yaml", help="accepted for compatibility; the base.yaml constants are built in")
    p.add_argument("--tts_checkpoint", default="pretrained_models/epoch=0-step=55872.ckpt", help="Path to TTS model checkpoint")
    p.add_argument("--flow_encoder", default="pretrained_models/flow_encoder.pt", help="Path to flow encoder weights (voice prompt)")
    p.add_argument("--speech_tokenizer", default="pretrained_models/speech_tokenizer_v2.onnx", help="unused (front-end)")
    p.add_argument("--campplus", default="pretrained_models/campplus.onnx", help="unused (front-end)")
    p.add_argument("--hift", default="pretrained_models/hift.pt", help="Path to HiFT vocoder weights")
    p.add_argument("--n_timesteps", type=int, default=10, help="Number of diffusion timesteps")
    p.add_argument("--length_scale", type=float, default=0.9, help="Length scale for speech duration control")
    p.add_argument("--tokens", default=None, help="JSON with the five id lists (and optionally spk_embed)")
    p.add_argument("--synthetic", type=int, default=0, help="use N synthetic tokens and synthetic weights")
    p.add_argument("--synthetic-prompt", type=int, default=0, help="with --synthetic: K synthetic prompt tokens (voice-cloning path)")
    p.add_argument("--seed", type=int, default=0, help="seed of the vocoder's source-noise draws")
    args = p.parse_args(argv)

    import torch

    import jyutvoice_amd
    from jyutvoice_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("no AMD GPU visible: jyutvoice_amd has no CPU path")
    device = torch.device("cuda:0")
    print(f"Using device: {device} ({torch.cuda.get_device_name(0)})")
    tts, hift = jyutvoice_amd.build_default(device)

    prompt_feat = prompt_h = None
    if args.synthetic:
        tts.load_state_dict(synth.tts_state_dict())
        hift.load_state_dict(synth.hift_state_dict())
        u = synth.batch(1, args.synthetic)
        ids = {k: u[k] for k in ("x", "lang", "tone", "word_pos", "syllable_pos")}
        spk = u["spk_embed"]
        if args.synthetic_prompt:
            from jyutvoice_amd.flow.encoder import FlowEncoder
            flow_encoder = FlowEncoder(device=device)
            flow_encoder.load_state_dict(synth.prompt_state_dict())
            ptok, plen = synth.prompt_tokens(1, args.synthetic_prompt)
            prompt_h, _ = flow_encoder(ptok, plen)
            prompt_feat = torch.randn(1, 2 * args.synthetic_prompt, 80, generator=torch.Generator().manual_seed(args.seed))
    else:
        if not args.tokens:
            raise SystemExit("--text/--ref_audio need the reference's G2P and ONNX front-ends, which are outside this build; "
                             "pass --tokens tokens.json (five id lists [+ spk_embed]) or --synthetic N")
        print(f"Loading TTS model from {args.tts_checkpoint}...")
        ckpt = torch.load(args.tts_checkpoint, map_location="cpu", weights_only=False)
        tts.load_state_dict(ckpt["state_dict"] if "state_dict" in ckpt else ckpt)
        print(f"Loading HiFT vocoder from {args.hift}...")
        hift.load_state_dict(torch.load(args.hift, map_location="cpu"))
        from jyutvoice_amd.utils.text import load_tokens_json
        tok = json.load(open(args.tokens))
        try:      # the contract of get_text (infer.py:189-206): equal lengths, ids inside the embedding tables, blanks in place
            ids = load_tokens_json(tok)
        except ValueError as e:
            raise SystemExit(str(e))
        spk = torch.tensor(tok["spk_embed"], dtype=torch.float32).view(1, 192) if "spk_embed" in tok else torch.randn(1, 192)
        if "prompt_token" in tok and ("prompt_feat" in tok or "prompt_wav_24k" in tok):   # infer.py:386-392
            from jyutvoice_amd.flow.encoder import load_flow_encoder
            print(f"Loading flow encoder from {args.flow_encoder}...")
            flow_encoder = load_flow_encoder(args.flow_encoder, device)
            ptok = torch.tensor(tok["prompt_token"], dtype=torch.int64).view(1, -1)
            prompt_h, _ = flow_encoder(ptok, torch.tensor([ptok.shape[1]], dtype=torch.int64))
            if "prompt_wav_24k" in tok:
                from jyutvoice_amd.utils.audio import extract_speech_feat
                prompt_feat, _ = extract_speech_feat(read_wav_24k(tok["prompt_wav_24k"]), device)
            else:
                prompt_feat = torch.tensor(tok["prompt_feat"], dtype=torch.float32).view(1, -1, 80)
    tts = tts.eval().to(device)
    hift = hift.eval().to(device)
    hift.manual_seed(args.seed)
    x_lengths = torch.tensor([ids["x"].shape[1]], dtype=torch.int64)

    print("Running TTS synthesis...")
    start = time.time()
    result = tts.synthesise(x=ids["x"], x_lengths=x_lengths, lang=ids["lang"], tone=ids["tone"], word_pos=ids["word_pos"],
                            syllable_pos=ids["syllable_pos"], prompt_feat=prompt_feat, prompt_h=prompt_h, spk_embed=spk,
                            n_timesteps=args.n_timesteps, length_scale=args.length_scale)
    wav, _ = hift.inference(result["mel"])
    torch.cuda.synchronize()
    print(f"Synthesis time: {time.time() - start:.2f} s (rtf of synthesise(): {result['rtf']:.4f})")
    print(f"Saving audio to {args.output}...")
    write_wav(args.output, wav[0])
    print(f"Generated audio saved to: {args.output}")
    print(f"Audio duration: {wav.shape[1] / 24000:.2f} seconds")


if __name__ == "__main__":
    main()
