#!/usr/bin/env python3
"""Summarises `rocprofv3 --pmc <group> --kernel-trace` passes over tools/attn_bench.py (one directory per counter group) for the
attention kernels: average per launch and the derived ratios quoted in DESIGN.md.  usage: attn_counters.py <dir> [<dir> ...]"""
import csv
import glob
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            m = re.search(r"(attn2_kernel|attn_kernel)<(\d+), (true|false)", k)
            if not m:
                continue
            name = f"{m.group(1)} D={m.group(2)}{' causal' if m.group(3) == 'true' else ''} grid={row['Grid_Size']}"
            a = acc[name][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for name, c in sorted(acc.items()):
    v = {k: a[0] / a[1] for k, a in c.items()}
    print(f"\n{name}  ({max(a[1] for a in c.values())} launches)")
    for k in sorted(v):
        print(f"  {k:38s} {v[k]:16.0f}")
    if "SQ_INSTS_VALU" in v and "SQ_INSTS_MFMA" in v and v["SQ_INSTS_MFMA"]:
        print(f"  -> vector instructions per MFMA: {v['SQ_INSTS_VALU'] / v['SQ_INSTS_MFMA'] - 1:.1f}   (SQ_INSTS_VALU includes the MFMAs)")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_VALU_MFMA_COEXEC_CYCLES" in v:
        print(f"  -> share of MFMA-busy cycles with a vector instruction co-executing: {v['SQ_VALU_MFMA_COEXEC_CYCLES'] / v['SQ_VALU_MFMA_BUSY_CYCLES']:.2f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        # BUSY_CYCLES is summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
        print(f"  -> MFMA pipe busy share of the launch: {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (v['GRBM_GUI_ACTIVE'] / 8):.2f}")
    if all(k in v for k in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY")):
        w = v["SQ_WAVE_CYCLES"]
        print(f"  -> wave time: issuing {v['SQ_ACTIVE_INST_ANY'] / w:.2f}, waiting (s_waitcnt / barrier) {v['SQ_WAIT_ANY'] / w:.2f}, "
              f"issue-stalled {max(0.0, 1 - (v['SQ_ACTIVE_INST_ANY'] + v['SQ_WAIT_ANY']) / w):.2f}")
