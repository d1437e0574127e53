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
            k = k.split("(")[0].replace("void jv::", "")
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for k, cs in acc.items():
    print(f"## {k}")
    print("| counter | per dispatch | dispatches |\n|---|---|---|")
    for c, (s, n) in sorted(cs.items()):
        print(f"| {c} | {s / n:,.0f} | {n} |")
    print()
