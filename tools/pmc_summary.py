#!/usr/bin/env python3
"""Average each PMC counter per dispatch and kernel over the passes tools/pmc.sh wrote (counter_collection.csv files)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if "jv" not in k:
                continue
            k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("jv::", "")
            k = k.split("(")[0]      # "attn64_s_kernel<5, 2, 0>(jv::AttnArgs)" -> "attn64_s_kernel<5, 2, 0>"
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
def total(cs, name):
    s, n = cs.get(name, (0.0, 0))
    return s / n if n else 0.0


# largest kernels first (by busy cycles), each with the two derived readings the DESIGN quotes
for k, cs in sorted(acc.items(), key=lambda kv: -total(kv[1], "GRBM_GUI_ACTIVE") * max(1, kv[1].get("GRBM_GUI_ACTIVE", (0, 1))[1])):
    print(f"## {k}")
    cyc = total(cs, "GRBM_GUI_ACTIVE") / 8.0      # the counter sums over the 8 XCDs
    mfma, valu, busy = total(cs, "SQ_INSTS_MFMA"), total(cs, "SQ_INSTS_VALU"), total(cs, "SQ_VALU_MFMA_BUSY_CYCLES")
    if cyc and mfma:
        print(f"matrix pipe busy {100.0 * busy / 1024.0 / cyc:.1f} % of {cyc / 1e3:.1f} K cycles; {valu / mfma:.1f} vector instructions per MFMA; "
              f"LDS bank-conflict cycles {total(cs, 'SQ_LDS_BANK_CONFLICT'):,.0f} of {total(cs, 'SQ_LDS_IDX_ACTIVE'):,.0f} active\n")
    print("| counter | per dispatch | dispatches |\n|---|---|---|")
    for c, (s, n) in sorted(cs.items()):
        print(f"| {c} | {s / n:,.0f} | {n} |")
    print()
