#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from two rocprofv3 PMC passes over bench.py (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; separate
passes as for the traffic counters): busy share = (MFMA busy cycles summed over the chip's 1024 SIMDs / 1024) / (GPU-active
cycles summed over the 8 XCDs / 8), averaged per launch.  north_star asks for "MFMA utilisation against chip peak": this is the
counter-side statement beside the flop-side one of bench.py's roofline (achieved TFLOP/s / 2.5 PFLOP/s).
usage: mfma_busy_summary.py <dir of the MFMA pass> <dir of the GRBM pass>"""
import csv
import glob
import sys
from collections import defaultdict

import re


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*$", "", row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
            a = acc[k]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


busy, act = load(sys.argv[1], "SQ_VALU_MFMA_BUSY_CYCLES"), load(sys.argv[2], "GRBM_GUI_ACTIVE")
rows = []
for k in busy:
    if k in act and act[k][1] and busy[k][1]:
        b, a = busy[k][0] / busy[k][1], act[k][0] / act[k][1]
        rows.append((b / 1024 / (a / 8) if a else 0.0, a / 8, busy[k][1], k))
print(f"{'kernel':70s} {'launches':>8s} {'GPU-active cycles / launch':>27s} {'MFMA pipe busy share':>21s}")
for share, cyc, n, k in sorted(rows, key=lambda r: -r[1] * r[2]):
    if cyc * n < 1e6:
        continue
    print(f"{k:70s} {n:8d} {cyc:27.0f} {share:21.3f}")
