#!/usr/bin/env python3
"""Which GEMM form the estimator should take at each batch size (run on the GPU box through gpurun):
    python tools/regime_sweep.py [--timesteps 2] [--batches 1,2,4,...]  -> gpurun_out/regime_sweep.json + a table on stdout

For every batch size B (utterances of 300 frames) the CFM loop alone (bench.py --workload c2) is timed with the transformer
linears on the tile kernels (JV_NO_ROWGEMM=1) and on the row-owning kernels at every forced tile height (JV_ROWGEMM_RT = 2..5);
the default (what rowgemm_tile() picks) is timed as well, so the table shows at a glance where the cost model is off.
Each configuration is its own process (the context flags are read at jv_create)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(B, T2, n, env_extra):
    env = dict(os.environ, JV_DYNAMIC_ENV="1", **env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--batch", str(B), "--tokens", str(T2), "--timesteps", str(n),
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-exact-range", "--no-profile"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        print(f"  [B={B} {env_extra}] timed out after 300 s", file=sys.stderr, flush=True)
        return {"error": "timeout"}
    if r.returncode == 0:
        for ln in r.stdout.splitlines()[::-1]:
            if ln.startswith("{"):
                return json.loads(ln)["ms_per_step"]
    # a crashed or NaN-asserting bench must not read as "not run": keep the diagnostic
    tail = "\n".join((r.stderr or r.stdout).strip().splitlines()[-6:])
    print(f"  [B={B} {env_extra}] bench failed with exit code {r.returncode}:\n{tail}", file=sys.stderr, flush=True)
    return {"error": f"exit {r.returncode}", "stderr_tail": tail}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timesteps", type=int, default=2)
    ap.add_argument("--batches", default="1,2,4,6,8,12,16,24,32,40,48,64")
    ap.add_argument("--frames", type=int, default=300)
    args = ap.parse_args()
    out = {}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    print(f"{'B':>4} {'default':>9} {'tile':>9} {'rt2':>9} {'rt3':>9} {'rt4':>9} {'rt5':>9}   frames/s (default)", flush=True)
    for B in [int(b) for b in args.batches.split(",")]:
        row = {"default": run(B, args.frames // 2, args.timesteps, {}),
               "tile": run(B, args.frames // 2, args.timesteps, {"JV_NO_ROWGEMM": "1"})}
        for rt in (2, 3, 4, 5):
            if 2 * B * (args.frames + 4) // (16 * rt) >= 24:      # a forced row tile needs a few workgroups to mean anything
                row[f"rt{rt}"] = run(B, args.frames // 2, args.timesteps, {"JV_ROWGEMM_RT": str(rt)})
        out[B] = row
        num = lambda v: isinstance(v, (int, float))
        f = lambda v: f"{v:9.2f}" if num(v) else ("    ERROR" if isinstance(v, dict) else "        -")
        rate = f"{B * args.frames / (row['default'] * 1e-3):10.0f}" if num(row["default"]) else "         -"
        print(f"{B:4d} {f(row['default'])} {f(row['tile'])} {f(row.get('rt2'))} {f(row.get('rt3'))} {f(row.get('rt4'))} {f(row.get('rt5'))}   "
              f"{rate}", flush=True)
        with open(os.path.join(ROOT, "gpurun_out", "regime_sweep.json"), "w") as fh:      # after every row: a late failure loses nothing
            json.dump({"timesteps": args.timesteps, "frames": args.frames, "ms_per_pass": out}, fh, indent=1)


if __name__ == "__main__":
    main()
