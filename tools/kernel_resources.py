#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy report of every kernel of the product library (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).
usage: tools/kernel_resources.py [-DPG_F32 ...] [--grep k_solve]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(extra=(), pattern=""):
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", "pg_api.hip", "-o", "/dev/null", *extra]
    out = subprocess.run(cmd, cwd=os.path.join(ROOT, "pigeon.jl_amd", "csrc"), capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip() or v
            cur = {"name": re.sub(r"\(.*", "", name)}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0]] = v
    print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>6s}")
    for r in rows:
        if pattern in r["name"]:
            print(f"{r['name'][:60]:60s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('ScratchSize', '?'):>8s} {r.get('Occupancy', '?'):>4s} {r.get('LDS', '?'):>6s}")
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    pat = ""
    if "--grep" in args:
        i = args.index("--grep"); pat = args[i + 1]; del args[i:i + 2]
    report(args, pat)
