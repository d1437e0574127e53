#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel time from the kernel-trace stats and HBM bytes per launch
from the PMC passes (FETCH_SIZE doubled, as MI355X_MICROARCH.md prescribes for gfx950; both counters are in KiB)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    """rocprofv3 kernel name -> the name bench.py's in-library profiler gives the same launches"""
    m = re.search(r"rowgemm_(?:wd_|wa_)?kernel<(\d+), (\d+)>", name)
    if m:
        epi = {"0": "", "1": ",gelu", "2": ",res", "3": ",res,ln", "4": ",qkv"}.get(m.group(2), "")
        return f"rowgemm_h3<{16 * int(m.group(1))}x256{epi}>"
    m = re.search(r"rowblock_kernel<(\d+), (true|false|0|1)(?:, (?:true|false|0|1))?>", name)      # (<RT, QKV, STAG>)
    if m:
        return f"rowblock_h3<{16 * int(m.group(1))}x256{',qkv' if m.group(2) in ('true', '1') else ''}>"
    m = re.search(r"rowres_kernel<(\d+)(?:, (\w+))?>", name)      # a whole resnet in one launch (+ the following block's q | k | v)
    if m:
        return f"rowres_h3<{16 * int(m.group(1))}x256{',qkv' if m.group(2) in ('true', '1') else ''}>"
    m = re.search(r"rowffn_kernel<(\d+)>", name)
    if m:
        return f"rowffn_h3<{16 * int(m.group(1))}x256>"
    m = re.search(r"rowconv_(?:wd_)?kernel<(\d+)(?:, (true|false|0|1))?>", name)      # (<RT, RES>: res_conv folded in)
    if m:
        return f"rowconv_h3<{16 * int(m.group(1))}x256,k3{'+res' if m.group(2) in ('true', '1') else ''}>"
    m = re.search(r"attn64_s_kernel<(\d), \d+(?:, \d+)?>", name)
    if m:
        return f"attn64_s<{64 * int(m.group(1))} q>"
    m = re.search(r"hiftpair_kernel<(\d+), (\d+)(?:, \d+)?>", name)
    if m:
        return f"hiftpair_h3<{80 * int(m.group(2))}x{m.group(1)},snake>"
    m = re.search(r"hiftconv_kernel<(\d+), (\d+)>", name)
    if m:
        return f"hiftconv_h3<{80 * int(m.group(2))}x{m.group(1)},snake>"
    m = re.search(r"attn64_pl_kernel<(\d)(?:, \d)?>", name)
    if m:
        return f"attn64_pl<{m.group(1)} waves>"
    m = re.search(r"conv_gemm_x6_kernel<(\d+), (\d+), \d+, \d+, (\d), (\d), (\d)(?:, (\d+))?(?:, (\d+))?>", name)
    if m:
        pro = {"0": "", "1": ",snake", "2": ",lrelu"}[m.group(3)]
        epi = {"0": "", "1": ",gelu", "9": ",gelu", "2": ",res", "4": ",generic"}.get(m.group(4), "")
        dma = ",dmaA" if m.group(5) == "3" else ""
        fam = "conv_gemm_h3" if m.group(7) == "2" else "conv_gemm_x6"      # last parameter: planes per operand
        return f"{fam}<{m.group(1)}x{m.group(2)}{pro}{dma}{epi}>"
    m = re.search(r"conv_gemm_kernel<(\d+), (\d+), \d+, \d+, \d+, (\d), (\d)>", name)
    if m:
        pro = {"0": "", "1": ",snake", "2": ",lrelu"}[m.group(3)]
        epi = {"0": "", "1": ",gelu", "2": ",res", "4": ",generic"}.get(m.group(4), "")
        return f"conv_gemm<{m.group(1)}x{m.group(2)}{pro}{epi}>"
    m = re.search(r"attn64_x6_kernel<(\d)(?:, \d+)?(?:, (\d+))?>", name)
    if m:
        return f"{'attn64_h3' if m.group(2) == '2' else 'attn64_x6'}<{m.group(1)} waves>"      # last parameter: planes
    m = re.search(r"conv_gemm_kernelILi(\d+)ELi(\d+)ELi\d+ELi\d+ELi\d+ELb([01])ELi(\d)", name)
    if m:
        pro = {"0": "", "1": ",snake", "2": ",lrelu"}[m.group(4)]
        return f"conv_gemm<{m.group(1)}x{m.group(2)}{',ln' if m.group(3) == '1' else ''}{pro}>"
    m = re.search(r"conv_gemm_kernel<(\d+), (\d+), \d+, \d+, \d+, (true|false), (\d)>", name)
    if m:
        pro = {"0": "", "1": ",snake", "2": ",lrelu"}[m.group(4)]
        return f"conv_gemm<{m.group(1)}x{m.group(2)}{',ln' if m.group(3) == 'true' else ''}{pro}>"
    m = re.search(r"attn64_kernel(?:ILi|<)(\d)", name)
    if m:
        return f"attn64<{m.group(1)} waves>"
    m = re.search(r"jv::?(\w+?)(?:_kernel)?(?:E|<|\()", name)
    return (m.group(1) if m else name)[:48]


def main(out):
    print(f"# rocprofv3 summary ({os.path.basename(out)})\n")
    def newest(pattern):      # (a directory merged back from several GPU calls holds every call's files: the last run's count)
        return sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)[-1:]


    stats = newest(os.path.join(out, "trace", "**", "*kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        print("## kernel-trace --stats (bench.py --steps 3 --warmup 1; all 4 passes of the path included)\n")
        print("| kernel | calls | total ms | avg us | % |")
        print("|---|---|---|---|---|")
        by_name = defaultdict(lambda: [0, 0.0])      # template instances that share a short name (kernel sizes of a pair kernel, ...) are one row
        for r in rows:
            a = by_name[short(r["Name"])]
            a[0] += int(r["Calls"])
            a[1] += float(r["TotalDurationNs"])
        for k, (n, t) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:22]:
            print(f"| {k} | {n} | {t / 1e6:.2f} | {t / n / 1e3:.2f} | {100 * t / tot:.1f} |")
        print(f"\ntotal kernel time {tot / 1e6:.1f} ms\n")

    traffic = defaultdict(dict)
    for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = newest(os.path.join(out, tag, "**", "*counter_collection.csv"))
        if not files:
            continue
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(files[0])):
            if r.get("Counter_Name") != counter:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        mult = 2.0 if counter == "FETCH_SIZE" else 1.0
        print(f"## --pmc {counter} (1 step; KiB per dispatch{', x2 gfx950 correction applied' if mult == 2 else ''})\n")
        print("| kernel | dispatches | MB per launch |")
        print("|---|---|---|")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"| {k} | {n} | {mult * v * 1024 / n / 1e6:.2f} |")
        print()
        for k, (n, v) in agg.items():
            traffic[k]["fetch_bytes" if counter == "FETCH_SIZE" else "write_bytes"] = round(mult * v * 1024 / n)

    if traffic:   # per-launch HBM bytes by kernel, read back by bench.py for roofline.traffic
        import json
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from jyutvoice_amd.build import source_hash
        with open(os.path.join(out, "pmc_traffic.json"), "w") as fh:
            extra = os.environ.get("JV_PROFILE_ARGS", "").split()      # e.g. --workload c2 --batch 8 --tokens 256 (tools/profile.sh)
            def arg(name, default):
                return extra[extra.index(name) + 1] if name in extra else default
            json.dump({"csrc_sha16": source_hash(),      # bench.py quotes these figures only for the build they were measured on
                       "shape": {"workload": arg("--workload", "c3"), "batch": int(arg("--batch", 32)), "tokens": int(arg("--tokens", 150)),
                                 "timesteps": int(arg("--timesteps", 10))},
                       "source": f"tools/profile.sh {os.path.basename(out)}: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and --pmc WRITE_SIZE, "
                                 f"separate passes over bench.py --steps 1 {' '.join(extra)}, bytes per launch",
                       "kernels": traffic}, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
