#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel: average FETCH_SIZE / WRITE_SIZE per launch.
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, i.e.
exactly half the bytes (MI355X_MICROARCH.md, HBM section) - the corrected column doubles it."""
import csv
import re
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = re.sub(r"\(.*$", "", k)[:60]
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
print(f"{'kernel':62s} {'launches':>8s} {'FETCH KiB/launch':>17s} {'corrected MB':>13s} {'WRITE KiB/launch':>17s}")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", [0, 1])[0]):
    f = c.get("FETCH_SIZE", [0.0, 0])
    w = c.get("WRITE_SIZE", [0.0, 0])
    n = max(f[1], w[1], 1)
    fa = f[0] / f[1] if f[1] else float("nan")
    wa = w[0] / w[1] if w[1] else float("nan")
    print(f"{k:62s} {n:8d} {fa:17.1f} {2 * fa * 1024 / 1e6:13.2f} {wa:17.1f}")
