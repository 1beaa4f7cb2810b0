#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one .hip source, from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(CPU only: cross-compiles for gfx950).  usage: kernel_resources.py mj-video_amd/csrc/attention.hip [name filter] [-D...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(src, defines=(), include_root=None):
    inc = include_root or ROOT
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", f"-I{inc}/include",
           f"-I{os.path.dirname(os.path.abspath(src))}", "-c", src, "-o", "/dev/null", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", *defines]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = [], None
    for ln in err.splitlines():
        m = re.search(r"remark: .*?(?:Function Name|Name): (\S+)", ln)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
            out.append(cur)
            continue
        m = re.search(r"remark: .*?    ([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


if __name__ == "__main__":
    src = sys.argv[1]
    flt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    for r in resources(src, defs):
        if flt in r["name"]:
            print(f"{r['name'][:110]:110s} vgpr {r.get('VGPRs', -1):4d} agpr {r.get('AGPRs', -1):4d} spill {r.get('VGPRs Spill', -1):3d} "
                  f"sgpr {r.get('SGPRs', -1):4d} scratch {r.get('ScratchSize [bytes/lane]', -1):5d} occ {r.get('Occupancy [waves/SIMD]', -1)} "
                  f"lds {r.get('LDS Size [bytes/block]', -1)}")
