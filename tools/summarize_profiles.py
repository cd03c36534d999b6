"""Condense gpurun_out/prof (written by tools/gpu_profile.sh on the GPU box) into profiles/<tag>/ and profiles/traffic.json.

usage: python tools/summarize_profiles.py r01_final
  kernel_stats.csv  : rocprofv3 --kernel-trace --stats summary, copied as is
  pmc_summary.json  : per-kernel per-launch means of every counter collected in the separate --pmc passes
  ../traffic.json   : HBM bytes per launch of the dominant kernel, (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction,
                      MI355X_MICROARCH.md HBM section); bench.py reports it as roofline.traffic
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "latest"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)

ks = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
assert len(ks) == 1, ks
shutil.copy(ks[0], os.path.join(dst, "kernel_stats.csv"))
ks32 = glob.glob(os.path.join(src, "stats_f32", "*", "*_kernel_stats.csv"))
if len(ks32) == 1:
    shutil.copy(ks32[0], os.path.join(dst, "kernel_stats_f32.csv"))


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if not k.startswith("pg::"):
            continue
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
summary = {k: {c: {"launches": v[1], "mean_per_launch": v[0] / v[1]} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)

# HBM bytes per launch of the dominant kernel (k_solve) and of the bandwidth-bound one (k_hji_lookup)
rows = list(csv.DictReader(open(ks[0])))


def traffic_of(match):
    cand = [r for r in rows if match in r["Name"]]
    if not cand:
        return None
    dom = max(cand, key=lambda r: float(r["TotalDurationNs"]))
    dk = short(dom["Name"])
    if dk not in summary or "FETCH_SIZE" not in summary[dk]:
        return None
    fs = summary[dk]["FETCH_SIZE"]["mean_per_launch"]; ws = summary[dk]["WRITE_SIZE"]["mean_per_launch"]
    return dk, {"hbm_bytes_per_launch": (2 * fs + ws) * 1024, "fetch_size_kib": fs, "write_size_kib": ws, "avg_launch_ns_rocprof": float(dom["AverageNs"]),
                "source": f"profiles/{tag}/pmc_summary.json"}


out = {"formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction for wide coalesced reads, MI355X_MICROARCH.md HBM section; separate --pmc passes)",
       "workload": "bench.py B=4096 coupled fp64 cold (k_solve); 2^20 lookups on the 13x13x9^5 grid (k_hji_lookup)", "kernels": {}}
for m in ("pg::k_solve", "pg::k_hji_lookup", "pg::k_linearize", "pg::k_nodes"):
    t = traffic_of(m)
    if t:
        out["kernels"][t[0]] = t[1]
        if m == "pg::k_solve":
            out.update({"kernel": t[0], **t[1]})
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for r in rows[:8]:
    print(f"{short(r['Name'])[:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']:>6s} %")
print(json.dumps({k: {c: round(v['mean_per_launch']) for c, v in cs.items()} for k, cs in summary.items() if 'k_solve' in k or 'lookup' in k}, indent=1))
