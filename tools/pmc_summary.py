#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel: average FETCH_SIZE / WRITE_SIZE per launch.
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, i.e.
exactly half the bytes (MI355X_MICROARCH.md, HBM section) - the corrected column doubles it."""
import csv
import re
import glob
import sys
from collections import defaultdict

import json
json_out = None
workload = "C2 workload"
args = sys.argv[1:]
if args and args[0] == "--json":
    json_out, args = args[1], args[2:]
if args and args[0] == "--workload":     # what the passes ran (goes into the json's "source" text)
    workload, args = args[1], args[2:]
EPI = ["bias", "bias_gelu", "bias_relu", "scale_res", "silu_mul", "rope_qkv"]


def tag_of(k):
    """kernel name as rocprofv3 prints it -> the library's profiler tag (the key bench.py looks traffic up by)"""
    m = re.match(r"t256::gemm256_kernel<(\d), 0(?:, \d)?>", k)     # <EPI, VAR = 0[, FUSE]> (round 4 added the folded-norm parameter)
    if m:
        return "gemm256_" + EPI[int(m.group(1))]
    m = re.match(r"t256::gemm256p_kernel<(\d)(?:, \d)?>", k)      # persistent form of the same tile kernel (round 3): same profiler tag
    if m:
        return "gemm256_" + EPI[int(m.group(1))]
    m = re.match(r"gemm256f8_kernel<(\d), (true|false)(?:, (true|false))?>", k)      # MXFP8 operands (round 4; round 5: <EPI, OUT8, SPLIT>)
    if m:
        return "gemm256f8s_slices" if m.group(3) == "true" else "gemm256f8_" + EPI[int(m.group(1))]
    m = re.match(r"splitk_finish256f8_kernel<(\d)", k)
    if m:
        return "gemm256f8s_finish_" + EPI[int(m.group(1))]
    if re.match(r"t256::gemm256_kernel<0, 7(?:, \d)?>", k):
        return "gemm256s_slices"        # K-sliced 256-tile launch: fp32 slice images to the workspace
    m = re.match(r"t256::splitk_finish256_kernel<(\d)>", k)
    if m:
        return "gemm256s_finish_" + EPI[int(m.group(1))]
    m = re.match(r"t128::gemm128_kernel<(\d), (true|false)>", k)
    if m:
        return "gemm128_" + EPI[int(m.group(1))] + ("_splitk" if m.group(2) == "true" else "")
    m = re.match(r"t64::gemm_skinny_kernel<(\d)", k)
    if m:
        return "gemm64_" + EPI[int(m.group(1))]
    m = re.match(r"(?:v2::attn2_kernel|attn_kernel)<(\d+), (true|false)", k)
    if m:
        return f"attn_d{m.group(1)}" + ("_causal" if m.group(2) == "true" else "")
    for name in ("layernorm", "rmsnorm", "row_stats", "quantize_mxfp8", "rope_split", "rope_heads", "patchify", "embed_gather", "reward_heads", "cls_rows"):
        if k.startswith(name):
            return name
    return None


acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in args:
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

if json_out:
    per = {}
    for k, c in acc.items():
        t = tag_of(k)
        f, w = c.get("FETCH_SIZE", [0.0, 0]), c.get("WRITE_SIZE", [0.0, 0])
        if t and f[1] and w[1]:
            per[t] = per.get(t, 0) + 0   # several template instances can share a tag (layernorm<2>, <8>): keep the largest
            per[t] = max(per[t], int((2 * f[0] / f[1] + w[0] / w[1]) * 1024))
    # the library profiles a K-sliced 256-tile GEMM (slices launch + finishing launch) under ONE scope gemm256s_<epilogue>:
    # emit the matching combined entry (finishing launch + the slices launch that feeds it)
    for t in [t for t in per if t.startswith("gemm256s_finish_")]:
        per["gemm256s_" + t[len("gemm256s_finish_"):]] = per[t] + per.get("gemm256s_slices", 0)
    import hashlib, os
    h = hashlib.sha1()
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mj-video_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(csrc, fn), "rb").read())
    json.dump({"source_sha1": h.hexdigest(),   # kernel sources these counters were collected on (bench.py checks it)
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_profiles.sh), " + workload + "; "
                         "traffic per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes): the factor 2 is the gfx950 FETCH_SIZE "
                         "correction of MI355X_MICROARCH.md; counted at the L2<->fabric boundary, Infinity-Cache hits included",
               "per_launch_bytes": per}, open(json_out, "w"), indent=1)
