#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s and 24 kHz RTF of the full synthesis path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[2], the one the metric is quoted on): full text encoder -> flow decoder (CFM Euler/CFG,
n_timesteps = 10) -> HiFT vocoder, batch 32 utterances per GPU, 150 tokens -> 300 mel frames = 6.0 s of 24 kHz audio
each, synthetic key-hashed weights and seeded inputs; fp32 data, fp32 accumulation, and contractions on
the 16-bit matrix cores at fp32-level accuracy: fp16x3 (operands scaled by an exact power of two and split into two fp16
planes = 22 significant bits, three MFMA products) wherever an upper bound of the operand is proven at load time or
measured on the device, bf16x6 (three bf16 planes = 24 bits, six products) elsewhere.  `value` is measured
in that default mode; `value_exact_range` repeats the timed loop with every contraction on bf16x6
(jv_flow_set_contraction(1)), and `parity` carries the default mode's measured error against the CPU oracle on the
utterances the CPU baseline synthesises anyway.
One "step" = one pass of that path over one batch; inputs are resident in HBM before the timed region.
With N > 1 (launched by torch.distributed.run, one rank per GPU) each rank synthesises its own 32 utterances (weak
scaling, no data-path collective) and the generated mels are all-gathered over RCCL at the end of every step.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel, measured with HIP events on the launch stream
inside the timed region (jv_profile_*); `cpu_baseline` is the CPU oracle (a plain-PyTorch port of the reference path)
timed on this box's host cores on a bounded sample -- reported, not a target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = fp32 vector peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA
HBM_PEAK_GBS = 8000.0


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from a committed PMC pass (tools/profile.sh: rocprofv3 --pmc FETCH_SIZE, doubled as
    MI355X_MICROARCH.md prescribes for gfx950, and --pmc WRITE_SIZE; counters cannot be read from inside this process).
    Quoted only for the shape those passes ran (the default C3 shape, or BASELINE.json configs[1] = --workload c2 --batch 8
    --tokens 256 from profiles/rNN_pmc_traffic_c2.json) AND only when the file was measured on this very
    build: profiles/rNN_pmc_traffic.json carries the hash of the kernel sources (jyutvoice_amd.build.source_hash) and a
    kernel edited since makes `traffic` null instead of silently stale."""
    import glob

    from jyutvoice_amd.build import source_hash
    if args.strong or args.ragged:
        return {"traffic_note": "PMC passes exist for the equal-length weak-scaling shapes only"}
    shape = {"workload": args.workload, "batch": args.batch, "tokens": args.tokens, "timesteps": args.timesteps}
    here, want = os.path.dirname(os.path.abspath(__file__)), source_hash()
    for path in sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        with open(path) as fh:
            d = json.load(fh)
        # (a file without a shape is the default C3 pass; profiles/rNN_pmc_traffic_c2.json is BASELINE.json configs[1])
        if d.get("csrc_sha16") != want or d.get("shape", {"workload": "c3", "batch": 32, "tokens": 150, "timesteps": 10}) != shape:
            continue
        k, note = d["kernels"].get(kernel), ""
        if not k and kernel.endswith(",ln>"):
            # rocprofv3 sees template arguments only: the fused block that also writes the next LayerNorm's planes to HBM
            # (`...,ln>`, the q|k|v-split regime) and the plain one are the same instantiation there, averaged together
            k, note = d["kernels"].get(kernel[:-4] + ">"), " (averaged with the launches of the same instantiation that do not write the LayerNorm planes)"
        if not k or "fetch_bytes" not in k or "write_bytes" not in k:
            continue
        return {"traffic": k["fetch_bytes"] + k["write_bytes"], "traffic_fetch": k["fetch_bytes"], "traffic_write": k["write_bytes"],
                "traffic_source": f"profiles/{os.path.basename(path)} (csrc_sha16 {want}) <- " + d["source"] + note}
    return {"traffic_note": f"no PMC pass under profiles/ was measured on this build (csrc_sha16 {want}): run tools/profile.sh"}


def kernel_peak(name: str):
    """(peak TFLOP/s of algorithmic fp32 work, description) for a profiled kernel family"""
    if name.startswith("conv_gemm_x6") or name.startswith("attn64_x6"):
        # bf16x6: every fp32-accurate multiply-add costs six bf16 MFMA products, so the ceiling for ALGORITHMIC flops is
        # the dense bf16 MFMA peak / 6
        return BF16_MFMA_PEAK_TFLOPS / 6.0, ("fp32 operands split into 3 bf16 planes, 6 x v_mfma_f32_32x32x16_bf16 per product, "
                                             "fp32 accumulate; peak = dense bf16 MFMA (2500) / 6")
    if name.startswith(("conv_gemm_h3", "attn64_h3", "attn64_pl", "rowgemm_h3", "rowconv_h3", "rowres_h3", "rowffn_h3", "rowblock_h3", "hiftconv_h3", "hiftpair_h3", "attn64_r", "attn64_s")):
        # fp16x3: three fp16 MFMA products per fp32-accurate multiply-add (fp16 and bf16 MFMA rates are equal)
        return BF16_MFMA_PEAK_TFLOPS / 3.0, ("fp32 operands scaled by an exact power of two and split into 2 fp16 planes (22 bits), "
                                             "3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; peak = dense fp16 MFMA (2500) / 3")
    return FP32_MFMA_PEAK_TFLOPS, "fp32 inputs, fp32 accumulate (v_mfma_f32_32x32x2_f32); peak = dense fp32 MFMA"


def path_flops(workload: str, utt_tokens, n_steps: int) -> float:
    """Algorithmic FLOP (2 x multiply-add) of one pass, SURVEY.md 8(d): per utterance of Tt tokens -> T = 2 Tt frames,
      flow estimator, per frame and Euler step, CFG x2:  2 * 2 * (7 360 512 convs + 58 720 256 linears + 57 344 T attention)
      HiFT:     612.3 MFLOP per mel frame
      encoder + duration predictor:  (52.354 M + 6 layers * 2 * 2 * 576 * Tt attention) per token  (= 52.7 M at Tt = 150)"""
    total = 0.0
    for tt in utt_tokens:
        t = 2 * tt
        total += n_steps * t * 2.0 * 2.0 * (7360512.0 + 58720256.0 + 57344.0 * t)
        if workload == "c3":
            total += 612.3e6 * t + tt * (52.354e6 + 6 * 2304.0 * tt)
    return total


def host_cores() -> int:
    """CPU threads this process may really use: cgroup quota if any, else the affinity mask, capped at the GPU box's
    documented 16-core share when neither says less."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(n_tokens: int, n_timesteps: int, repeats: int, hip=None):
    """The oracle on the host, as BASELINE.md 3 plans it: B = 1 sequential semantics (what the reference does), end to end,
    for (a) ONE utterance of the benchmarked shape (C3: 150 tokens -> 300 frames) and (b) configs[0] (C1: 64 tokens -> 128
    frames); per stage (encoder + duration predictor + length regulation, CFM loop, HiFT) and total; 1 warm-up, then the
    median of `repeats` runs (each of another utterance of the same shape: utterance i of the batch is the oracle's
    utterance i, same seeded inputs, so the timed runs double as the parity check).
    hip (optional): {"mel" [B,80,T], "wav" [B,480T], "s" [B,1,480T]} CPU tensors of the benchmarked pass -- mel max-abs against
    the oracle's mel, waveform RMS against the oracle's vocoder on the HIP mel with the HIP source signal (outside the timed part)."""
    import math
    import statistics

    import torch

    from jyutvoice_amd import synth
    from oracle import hift as ohift
    from oracle import tts as otts
    torch.set_num_threads(host_cores())
    tts_sd, hift_sd, noise = synth.tts_state_dict(fixed_duration=1.5), synth.hift_state_dict(), synth.rand_noise()
    w = ohift.fold_weight_norm(hift_sd)
    g = torch.Generator().manual_seed(0)
    mels = {}

    def one(i, tokens, keep):
        tm = {}
        u = synth.batch(1, tokens, first_index=i)
        t0 = time.perf_counter()
        res = otts.synthesise(tts_sd, noise, u["x"], u["x_lengths"], u["lang"], u["tone"], u["word_pos"], u["syllable_pos"],
                              u["spk_embed"], None, n_timesteps=n_timesteps, timings=tm)
        mel = res["mel"]
        T = mel.shape[2]
        t1 = time.perf_counter()
        f0 = ohift.f0_predict(w, mel)
        phase = (torch.rand(1, 9, 1, generator=g) * 2 - 1) * math.pi
        phase[:, 0] = 0
        s = ohift.source(w, f0, phase, torch.randn(1, 9, 480 * T, generator=g))
        ohift.decode(w, mel, s)
        t2 = time.perf_counter()
        if keep:
            mels[i] = mel
        return {"frames": T, "encoder_dp": tm["encoder_dp"], "cfm": tm["cfm"], "hift": t2 - t1, "total": t2 - t0}

    def config(tokens, keep):
        one(0, tokens, keep)                      # warm-up
        runs = [one(1 + i, tokens, keep) for i in range(repeats)]
        med = {k: statistics.median(r[k] for r in runs) for k in ("encoder_dp", "cfm", "hift", "total")}
        T = runs[0]["frames"]
        return {"tokens": tokens, "frames": T, "ms": {k: round(1e3 * v, 1) for k, v in med.items()},
                "frames_per_s": round(T / med["total"], 2), "rtf": round(med["total"] / (T * 0.02), 4)}, sum(r["total"] for r in runs)

    with torch.inference_mode():
        c3, cpu_s3 = config(n_tokens, True)
        c1, cpu_s1 = config(64, False)
        parity = None
        if hip is not None:
            mel_err = wav_err = 0.0
            checked = sorted(i for i in mels if i < hip["mel"].shape[0])
            for i in checked:
                mel_err = max(mel_err, float((hip["mel"][i:i + 1] - mels[i]).abs().max()))
                want = ohift.decode(w, hip["mel"][i:i + 1], hip["s"][i:i + 1])
                wav_err = max(wav_err, float((hip["wav"][i:i + 1] - want).pow(2).mean().sqrt()))
            tol = {"mel_max_abs": 1e-3, "wav_rms": 1e-4}
            parity = {"mel_max_abs": float(f"{mel_err:.3e}"), "wav_rms": float(f"{wav_err:.3e}"), "utterances": checked, "tolerance": tol,
                      "ok": bool(checked) and mel_err <= tol["mel_max_abs"] and wav_err <= tol["wav_rms"],
                      "against": "CPU oracle (oracle/, B = 1 per utterance): its mel; its vocoder on the HIP mel with the HIP source signal"}
    out = {"value": c3["frames_per_s"], "unit": "mel-frames/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(), "kind": "port",
           "rtf": c3["rtf"],
           "sample": f"one utterance of the benchmarked shape ({n_tokens} tokens -> {c3['frames']} frames) and one of configs[0] (64 tokens -> "
                     f"{c1['frames']} frames), n_timesteps={n_timesteps}, B=1 sequential, encoder+CFM+HiFT end to end, 1 warm-up then the "
                     f"median of {repeats} runs each ({cpu_s3 + cpu_s1:.1f} s of timed CPU work); `value` is the benchmarked shape's",
           "c3_utterance": c3, "c1": c1}
    return out, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--tokens", type=int, default=150, help="text tokens per utterance (2 mel frames each)")
    ap.add_argument("--timesteps", type=int, default=10)
    ap.add_argument("--workload", choices=("c3", "c2"), default="c3",
                    help="c3 (default, the headline): full encoder -> flow -> HiFT; c2: the CFM Euler loop alone on [B,80,T] "
                         "N(0,1) mu (BASELINE.json configs[1]: --workload c2 --batch 8 --tokens 256)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--profile-steps", type=int, default=1,
                    help="how many of the K timed steps carry the per-launch HIP events (each event pair costs GPU time: all "
                         "steps instrumented lowers `value` by ~5 %%; one step gives ~1700 launches of the dominant kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-range", action="store_true", help="skip the second timed loop (bf16x6 everywhere)")
    ap.add_argument("--cpu-utts", type=int, default=5, help="CPU-baseline runs per configuration (median of)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --total-batch utterances over ALL ranks (BASELINE.json configs[4] as written: 256 over 8 "
                         "GPUs) instead of --batch per rank")
    ap.add_argument("--total-batch", type=int, default=256)
    ap.add_argument("--ragged", action="store_true",
                    help="utterances of seeded, unequal lengths: token counts uniform in [--ragged-min, --tokens] (one of them "
                         "--tokens, so the padded shape is the equal-length run's); `value` counts VALID frames only")
    ap.add_argument("--ragged-min", type=int, default=60)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev_index = local_rank % max(1, torch.cuda.device_count())   # one rank per GPU on the real node; the modulo only matters
    torch.cuda.set_device(dev_index)                             # when rehearsing several ranks on a 1-GPU box (gloo)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("JV_FORCE_DIST") == "1"   # JV_FORCE_DIST: run the collective path with one rank
    coll_dev = device      # where the collectives' tensors live: the GPU under RCCL, the host under gloo (rehearsals on one card)
    if use_dist:
        backend = os.environ.get("JV_DIST_BACKEND", "nccl")      # nccl == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
            coll_dev = "cpu"

    import jyutvoice_amd
    from jyutvoice_amd import dist as jdist
    from jyutvoice_amd import engine, synth
    from jyutvoice_amd.runtime import get_runtime

    if args.strong:      # this rank's contiguous share of the fixed global batch
        lo_s, hi_s = jdist.shard_range(args.total_batch, rank, world)
        args.batch = hi_s - lo_s
    B, Tt, n_steps = args.batch, args.tokens, args.timesteps
    T = 2 * Tt
    tts, hift = jyutvoice_amd.build_default(device)
    get_runtime(device).ensure(B, T, Tt)
    tts.load_state_dict(synth.tts_state_dict(fixed_duration=1.5))     # every token -> ceil(1.5) = 2 frames
    hift.load_state_dict(synth.hift_state_dict())
    hift.manual_seed(1234 + rank)
    lo, _ = jdist.shard_range(args.total_batch if args.strong else B * world, rank, world)
    utt_tokens = [Tt] * B
    if args.ragged:      # real batches are ragged (jyutvoice/utils/mask.py:232-255 pads them to the longest)
        gl = torch.Generator().manual_seed(4321 + rank)
        utt_tokens = torch.randint(args.ragged_min, Tt + 1, (B,), generator=gl).tolist()
        utt_tokens[int(torch.randint(0, B, (1,), generator=gl))] = Tt
    batch = {k: v.to(device) for k, v in synth.batch(B, Tt, first_index=lo, lengths=utt_tokens).items()}
    valid_frames = 2 * sum(utt_tokens)      # fixed_duration = 1.5: two frames per token

    if args.workload == "c2":     # mu ~ N(0,1) [B,80,T], spks ~ N(0,1), cond = 0, full mask
        gen = torch.Generator().manual_seed(1234 + rank)
        c2_mu = torch.randn(B, 80, T, generator=gen).to(device)
        c2_spks = torch.randn(B, 80, generator=gen).to(device)
        c2_cond = torch.zeros(B, 80, T, device=device)
        c2_lens = torch.tensor([2 * t for t in utt_tokens], dtype=torch.int32, device=device) if args.ragged else None
        eng = get_runtime(device).ensure(B, T, Tt)

    src = {}

    def step(gather=True):
        """one pass; gather=False leaves out the end-of-step collective (passes that only SOME ranks run, outside the timed
        region: a collective entered by rank 0 alone would wait for the others forever)"""
        if args.workload == "c2":
            mel = eng.cfm_solve(c2_mu, c2_lens, c2_spks, c2_cond, n_steps, 1.0)
            return {"mel": mel, "mel_lengths": None}, None
        res = tts.synthesise(batch["x"], batch["x_lengths"], batch["lang"], batch["tone"], batch["word_pos"],
                             batch["syllable_pos"], batch["spk_embed"], None, n_timesteps=n_steps, batched=True)
        wav, src["s"] = hift.inference(res["mel"], lengths=res["mel_lengths"] if args.ragged else None)
        if use_dist and gather:
            if coll_dev == "cpu":      # gloo rehearsal: host tensors
                jdist.all_gather_mels(res["mel"].cpu(), res["mel_lengths"].cpu())
            else:
                jdist.all_gather_mels(res["mel"], res["mel_lengths"])
        return res, wav

    def timed_loop(n, profiled_steps=0):
        """barrier + synchronize, exactly n steps, synchronize + barrier; MAX over ranks"""
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(n):
            if profiled_steps and i == profiled_steps:
                engine.profile_enable(False)     # events stay queued on the stream; read back after the timed region
            out = step()
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    for _ in range(args.warmup):
        step()

    profile = not args.no_profile
    torch.cuda.synchronize(device)
    if profile:
        engine.profile_report()          # drop anything recorded during warm-up
        engine.profile_enable(True)
    prof_steps = min(max(1, args.profile_steps), args.steps) if profile else 0
    elapsed, (res, wav) = timed_loop(args.steps, prof_steps)
    assert res["mel"].shape == (B, 80, T) and torch.isfinite(res["mel"]).all(), res["mel"].shape
    if args.workload == "c3":
        assert wav.shape == (B, 480 * T) and torch.isfinite(wav).all(), wav.shape
    kern = {}
    event_overhead_us = None
    if profile:
        engine.profile_enable(False)
        kern = engine.profile_report()
        pair = kern.pop("_empty_event_pair", None)      # what an event pair measures around nothing
        if pair:
            event_overhead_us = 1e3 * pair["ms"]
    hip_out = None
    if args.workload == "c3" and rank == 0:
        hip_out = {"mel": res["mel"].float().cpu(), "wav": wav.float().cpu(), "s": src["s"].float().cpu()}
    # the same loop with every contraction on bf16x6 (24-bit operands): what the fp16x3 engine buys, measured, not claimed
    elapsed_exact = None
    if not args.no_exact_range and not os.environ.get("JV_EXACT_RANGE"):
        rt_eng = get_runtime(device).ensure(B, T, Tt)
        rt_eng.set_exact_range(True)
        step()
        if profile:      # the same conditions as `value`: the first timed step carries the per-launch events (discarded here)
            torch.cuda.synchronize(device)
            engine.profile_enable(True)
        elapsed_exact, (res_x, _) = timed_loop(args.steps, prof_steps)
        if profile:
            engine.profile_enable(False)
            engine.profile_report()
        assert torch.isfinite(res_x["mel"]).all()
        rt_eng.set_exact_range(False)

    # per-stage GPU time (encoder + duration predictor + length regulation | CFM loop | HiFT): three more passes, outside the
    # timed region, with events on the launch stream at the stage boundaries; median
    stage_ms = None
    if args.workload == "c3" and rank == 0:
        import statistics
        runs = []
        for _ in range(3):
            tts.stage_events = []
            step_res, _ = step(gather=False)      # rank 0 alone runs these
            e_end = torch.cuda.Event(enable_timing=True)
            e_end.record(torch.cuda.current_stream(device))
            torch.cuda.synchronize(device)
            e0, e1, e2 = tts.stage_events
            runs.append((e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e_end)))
        tts.stage_events = None
        med = [statistics.median(r[i] for r in runs) for i in range(3)]
        stage_ms = {"encoder_dp_length_regulation": round(med[0], 3), "cfm_loop": round(med[1], 3), "hift": round(med[2], 3),
                    "sum": round(sum(med), 3),
                    "measured": "events on the launch stream at the stage boundaries, 3 passes outside the timed region, median"}

    # The default build runs 13 of the 14 stages' first q | k | v INSIDE the preceding resnet's launch (rowres_kernel<RT, true>), so
    # the profiler's conv-stack group -- per launch -- then holds those products too.  The stack by itself (resnets, down / up /
    # final convolutions, final projection: what the north star's HBM fraction and VERDICT r3's 13 ms are about) is measured
    # here, after everything timed: the same weights re-loaded with JV_NO_RES_QKV=1 (the C ABI builds a new context and reads
    # the switch), one warm pass, one pass under the profiler, then the default context restored.
    cs_alone = None
    if profile and rank == 0 and not os.environ.get("JV_NO_RES_QKV") and not os.environ.get("JV_NO_RES_PAIR"):
        os.environ["JV_NO_RES_QKV"] = "1"
        try:
            tts.load_state_dict(synth.tts_state_dict(fixed_duration=1.5))
            if args.workload == "c2":
                eng = get_runtime(device).ensure(B, T, Tt)
            step(gather=False)
            torch.cuda.synchronize(device)
            engine.profile_enable(True)
            step(gather=False)
            torch.cuda.synchronize(device)
            engine.profile_enable(False)
            cs_alone = engine.profile_report().get("_group:flow_conv_stack")
        finally:
            os.environ.pop("JV_NO_RES_QKV", None)
            tts.load_state_dict(synth.tts_state_dict(fixed_duration=1.5))
            if args.workload == "c2":
                eng = get_runtime(device).ensure(B, T, Tt)

    if rank == 0:
        frames = (args.total_batch if args.strong else world * B) * T * args.steps
        if args.ragged:      # every rank draws its own lengths from the same distribution; rank 0's sum stands for each
            frames = world * valid_frames * args.steps
        out = {
            "metric": "mel_frames_per_sec", "value": round(frames / elapsed, 1), "unit": "mel-frames/s",
            "rtf": round(elapsed / (frames * 0.02), 6), "x_realtime": round(frames * 0.02 / elapsed, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": ("f32 (bf16x6 split: 24-bit operands on the bf16 matrix cores, fp32 accumulate)" if os.environ.get("JV_EXACT_RANGE") else
                      "f32 (fp16x3 split: 22-bit operands on the fp16 matrix cores where a bound exists, bf16x6 = 24-bit elsewhere; fp32 accumulate)"),
            "data": "synthetic",
            "config": {"workload": ("C3: text encoder -> CFM flow decoder (Euler+CFG) -> HiFT vocoder, full synthesise()+inference()"
                                    if args.workload == "c3" else "C2: CFM flow decoder loop alone (Euler+CFG), N(0,1) mu, full mask"),
                       "utterances_per_gpu": B, "global_batch": args.total_batch if args.strong else B * world, "tokens": Tt, "mel_frames": T,
                       "audio_seconds_per_utterance": T * 0.02, "n_timesteps": n_steps, "parallelism": f"utterance-dp{world}",
                       "contraction": ("bf16x6 everywhere (JV_EXACT_RANGE)" if os.environ.get("JV_EXACT_RANGE") else
                                       "fp32-accurate split-plane MFMA: fp16x3 on the estimator's range-proven linears and "
                                       "attention, bf16x6 elsewhere (DESIGN.md 5; JV_EXACT_RANGE=1 forces bf16x6)")},
        }
        if args.ragged:
            out["config"]["ragged"] = {"tokens_min": min(utt_tokens), "tokens_max": max(utt_tokens), "valid_frames_per_gpu": valid_frames,
                                       "padded_frames_per_gpu": B * T, "fill": round(valid_frames / (B * T), 4),
                                       "note": "`value` counts valid frames only; lengths ~ U[--ragged-min, --tokens], seeded"}
        # the whole path against the engine's ceiling: algorithmic FLOP of a pass / wall time / (2500 / 3)
        pf = path_flops(args.workload, utt_tokens, n_steps) * (args.total_batch / B if args.strong else world)
        ptf = pf * args.steps / elapsed / 1e12
        out["roofline_path"] = {"bound": "mfma", "alg_tflop_per_pass": round(pf / 1e12, 3), "achieved": round(ptf, 2),
                                "peak": round(world * BF16_MFMA_PEAK_TFLOPS / 3.0, 1), "unit": "TFLOP/s",
                                "frac": round(ptf / (world * BF16_MFMA_PEAK_TFLOPS / 3.0), 4),
                                "note": "every contraction of the pass (SURVEY.md 8(d) formulas) over the wall time of the timed loop, "
                                        "against the fp16x3 ceiling (dense fp16 MFMA / 3) of the GPUs used"}
        if stage_ms:
            out["stage_ms"] = stage_ms
        groups = {k[len("_group:"):]: kern.pop(k) for k in [k for k in kern if k.startswith("_group:")]}
        cs = groups.get("flow_conv_stack")
        if cs and cs["ms"] > 0:
            # BASELINE.json's north star asks for the HBM fraction of the flow decoder's Conv1d stack.  Algorithmic bytes as
            # SURVEY.md 8(d) defines them: 96 064 B per CFG-sample-frame (each conv reads its input once and writes its output
            # once, elementwise fused = 0) + 29.5 MB of weights per estimator call; time = HIP events over every launch of the
            # stack (resnets, down / up / final convolutions, final projection, their LayerNorm passes) in the profiled step.
            calls = n_steps * prof_steps
            cs_bytes = calls * (96064.0 * 2 * (valid_frames if args.ragged else B * T) + 29.5e6)
            gbs = cs_bytes / (cs["ms"] * 1e-3) / 1e9
            out["conv_stack"] = {"bound": "hbm (as the north star prices it; the kernels themselves are matrix-pipe bound, DESIGN.md 5)",
                                 "alg_bytes_per_pass": round(cs_bytes / prof_steps), "ms_per_pass": round(cs["ms"] / prof_steps, 3),
                                 "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                 "launches_per_pass": cs["launches"] // prof_steps,
                                 "alg_tflops": round(cs["flops"] / (cs["ms"] * 1e-3) / 1e12, 2)}
            fused_qkv = any(k.startswith("rowres_h3") and k.endswith(",qkv>") for k in kern)      # (not in the column-split regime of few row tiles: C2)
            if cs_alone and cs_alone["ms"] > 0 and fused_qkv:
                # (the numbers above are the group as the default build launches it: q | k | v of 13 stages inside; these are the stack alone)
                one = n_steps * (96064.0 * 2 * (valid_frames if args.ragged else B * T) + 29.5e6)
                g1 = one / (cs_alone["ms"] * 1e-3) / 1e9
                out["conv_stack"]["includes"] = ("the first q | k | v of 13 of the 14 stages, computed inside the resnet launch that precedes it "
                                                 "(rowres_kernel<RT, true>): its time and FLOP are in ms_per_pass / alg_tflops, not in alg_bytes_per_pass")
                out["conv_stack"]["stack_alone"] = {
                    "ms_per_pass": round(cs_alone["ms"], 3), "achieved": round(g1, 1), "unit": "GB/s", "frac": round(g1 / HBM_PEAK_GBS, 4),
                    "launches_per_pass": cs_alone["launches"], "alg_tflops": round(cs_alone["flops"] / (cs_alone["ms"] * 1e-3) / 1e12, 2),
                    "how": "one pass under the profiler after the timed region, the same weights re-loaded with JV_NO_RES_QKV=1 "
                           "(q | k | v as launches of their own, outside the group): the Conv1d stack as VERDICT r3 measured it"}
        if kern:
            tot_ms = sum(v["ms"] for v in kern.values())
            name, d = max(((k, v) for k, v in kern.items() if v["flops"] > 0), key=lambda kv: kv[1]["ms"])
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
            peak, precision = kernel_peak(name)
            out["roofline"] = {
                "kernel": name, "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(tf / peak, 4), "traffic": None, "traffic_unit": "bytes per launch (fetch + write)",
                "launches": d["launches"], "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                "alg_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
                "alg_hbm_gbs": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1),
                "share_of_profiled_kernel_time": round(d["ms"] / tot_ms, 3),
                "measured": f"HIP events on the launch stream around every launch of the first {prof_steps} of the {args.steps} timed "
                            "steps, inside the timed region (jv_profile_*)",
                "precision": precision,
            }
            if event_overhead_us is not None:      # the fixed cost inside avg_launch_us; rocprofv3's kernel duration excludes it
                out["roofline"]["event_pair_overhead_us"] = round(event_overhead_us, 2)
                out["roofline"]["avg_launch_us_less_event_overhead"] = round(1e3 * d["ms"] / d["launches"] - event_overhead_us, 2)
            out["roofline"].update(pmc_traffic(name, args))
            def line(k, v):
                e = {"launches": v["launches"], "ms_per_step": round(v["ms"] / prof_steps, 3)}
                if v["ms"] <= 0:
                    return e
                if v["flops"] > 0:      # matrix-pipe kernels: algorithmic TFLOP/s against the precision's dense MFMA peak
                    tfs = v["flops"] / (v["ms"] * 1e-3) / 1e12
                    e.update(bound="mfma", tflops=round(tfs, 2), frac_of_peak=round(tfs / kernel_peak(k)[0], 3))
                else:                   # row-wise kernels: algorithmic GB/s against HBM
                    gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                    e.update(bound="hbm", gbs=round(gbs, 1), frac_of_peak=round(gbs / HBM_PEAK_GBS, 3))
                return e
            out["kernels"] = {k: line(k, v) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])}
            out["profiled_kernel_ms_per_step"] = round(tot_ms / prof_steps, 3)
            out["profiled_steps"] = prof_steps
        if elapsed_exact is not None:
            out["value_exact_range"] = round(frames / elapsed_exact, 1)
            out["ms_per_step_exact_range"] = round(1e3 * elapsed_exact / args.steps, 3)
            out["exact_range_note"] = ("the same timed loop, under the same conditions (first step instrumented), with every "
                                       "contraction on bf16x6 (24-bit operands, jv_flow_set_contraction(1)); `value` is the default fp16x3 mode")
        parity_ok = True
        if world == 1 and not args.no_cpu_baseline and args.cpu_utts > 0 and args.workload == "c3" and not args.ragged:
            out["cpu_baseline"], parity = cpu_baseline(Tt, n_steps, args.cpu_utts, hip_out)
            if parity:
                out["parity"] = parity
                parity_ok = parity["ok"]
        print(json.dumps(out), flush=True)
        if not parity_ok:      # a throughput measured on results outside the tolerance is not a result
            sys.stderr.write("bench.py: parity against the CPU oracle is outside the tolerance -- see `parity` on the JSON line\n")
            if use_dist:
                dist.destroy_process_group()
            sys.exit(3)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
