#!/usr/bin/env python3
"""Effective shader clock per kernel from a `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` pass (MI355X_MICROARCH.md, DVFS
give-back): clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch wall time.  The quotient reads high on dispatches shorter than about
0.3 ms; only dispatches of at least --min-us are averaged.  usage: clock_summary.py <dir> [--min-us 150]"""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 150.0
acc = defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        ns = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        if ns < min_us * 1e3:
            continue
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*$", "", k)[:60]
        a = acc[k]
        a[0] += float(row["Counter_Value"]) / 8.0
        a[1] += ns
        a[2] += 1
print(f"{'kernel':62s} {'dispatches':>10s} {'avg us':>9s} {'clock GHz':>10s} {'MFMA peak at that clock, TFLOP/s':>34s}")
for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    ghz = cyc / ns
    print(f"{k:62s} {n:10d} {ns / n / 1e3:9.1f} {ghz:10.3f} {2500.0 * ghz / 2.4:34.0f}")
