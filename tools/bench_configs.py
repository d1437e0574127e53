#!/usr/bin/env python3
"""bench.py lines of the other configurations on ONE box (through gpurun, repo root):
    python tools/bench_configs.py [tag]      -> gpurun_out/bench_configs_<tag>.json   (committed as profiles/rNN_bench_configs.json)
Every run is --no-cpu-baseline --no-exact-range --no-profile; a failed run is recorded with its stderr tail."""
import json
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = [
    ("c3", "", "the headline (32 utterances x 150 tokens, n = 10), no instrumentation"),
    ("b1", "--batch 1 --tokens 64", "C1's shape on the GPU: one utterance of 64 tokens"),
    ("c2", "--workload c2 --batch 8 --tokens 256", "BASELINE.json configs[1]: the CFM loop alone, 8 x 512 frames"),
    ("c4", "--timesteps 32", "C3 with n = 32"),
    ("b4", "--batch 4", ""), ("b8", "--batch 8", ""), ("b16", "--batch 16", ""), ("b22", "--batch 22", "as many frames as the ragged batch below holds"),
    ("b64", "--batch 64", "two rounds of the row-owning tiles"),
    ("ragged", "--ragged", "32 utterances of 60 .. 150 tokens (seeded), compact geometry; `value` counts valid frames"),
    ("ragged_uniform", "--ragged", "the same padded to the longest (JV_NO_COMPACT=1)"),
]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r"
    out = {}
    for key, flags, what in CONFIGS:
        env = dict(os.environ)
        if key == "ragged_uniform":
            env["JV_NO_COMPACT"] = "1"
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-exact-range", "--no-profile"] + flags.split()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            out[key] = {"flags": flags, "error": (r.stderr or "")[-400:]}
            print(key, "FAILED", flush=True)
            continue
        j = json.loads(lines[-1])
        out[key] = {"flags": flags + (" (JV_NO_COMPACT=1)" if key == "ragged_uniform" else ""), "what": what, "ms_per_step": j["ms_per_step"],
                    "mel_frames_per_s": j["value"], "x_realtime": j["x_realtime"], "stage_ms": {k: v for k, v in (j.get("stage_ms") or {}).items() if k != "measured"},
                    "roofline_path_frac": (j.get("roofline_path") or {}).get("frac")}
        if "ragged" in j.get("config", {}):
            out[key]["ragged"] = j["config"]["ragged"]
        print(key, j["ms_per_step"], j["value"], flush=True)
    dst = os.path.join(ROOT, "gpurun_out", f"bench_configs_{tag}.json")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as fh:
        json.dump({"note": "bench.py lines of one build on one box, back to back; --no-cpu-baseline --no-exact-range --no-profile", "configs": out}, fh, indent=1)


if __name__ == "__main__":
    main()
