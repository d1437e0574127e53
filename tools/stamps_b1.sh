#!/bin/bash
# phase shares of the tile GEMM launches of one single-utterance pass (tuning build, through gpurun, repo root):
#   JV_TUNING=1 JV_BUILD_TAG=tune python -m jyutvoice_amd.build   (here, before the call)
#   bash tools/stamps_b1.sh      -> gpurun_out/stamps_b1.txt
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
JYUTVOICE_HIP_LIB=$PWD/jyutvoice_amd/libjyutvoice_hip.tune.so JV_STAMPS=1 python bench.py --batch 1 --tokens 64 --steps 1 --warmup 1 \
  --no-cpu-baseline --no-exact-range --no-profile > gpurun_out/stamps_bench.json 2> gpurun_out/stamps_raw.log
python - <<'PY' > gpurun_out/stamps_b1.txt
import re, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0, 0.0])
pat = re.compile(r"\[stamps x6\] (\d+x\d+) K (\d+) taps (\d+) split (\d+) N (\d+) grid (\d+): prologue (\S+)\s+loop (\S+)\s+epilogue (\S+) ticks.*span (\S+) ticks, (\S+) us by events \((\S+) ticks/us\)")
for ln in open("gpurun_out/stamps_raw.log"):
    m = pat.search(ln)
    if not m:
        continue
    key = (m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)))
    a = agg[key]
    tpu = float(m.group(12))
    a[0] += 1
    a[1] += float(m.group(7)) / tpu; a[2] += float(m.group(8)) / tpu; a[3] += float(m.group(9)) / tpu
    a[4] += float(m.group(10)) / tpu; a[5] += float(m.group(11))
print(f"{'tile':8s} {'K':>5s} {'taps':>4s} {'split':>5s} {'N':>5s} {'grid':>5s} {'calls':>6s} {'prologue us':>12s} {'loop us':>8s} {'epilogue us':>12s} {'span us':>8s} {'events us':>10s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0] * kv[1][5]):
    n = a[0]
    print(f"{k[0]:8s} {k[1]:5d} {k[2]:4d} {k[3]:5d} {k[4]:5d} {k[5]:5d} {n:6d} {a[1] / n:12.2f} {a[2] / n:8.2f} {a[3] / n:12.2f} {a[4] / n:8.2f} {a[5] / n:10.2f}")
PY
cat gpurun_out/stamps_b1.txt
