"""Condense gpurun_out/prof (written by tools/gpu_profile.sh on the GPU box, PART = 1, 2, 3) into profiles/<tag>/ and profiles/traffic.json.

usage: python tools/summarize_profiles.py r03_final
  kernel_stats*.csv : rocprofv3 --kernel-trace --stats summaries, copied as they are: kernel_stats.csv = the cold fp64 headline; _dec = + config 5 (lateral N = 50 with the
                      wall rows: k_nodes_dec, k_qp_dec, k_solve_lat); _roll = + the closed-loop rollout (k_nodes_warm, k_advance, warm k_solve); _c3 = config 3 alone (tools/gpu_config3_probe.py f32: fp32 library, safety row,
                      five cold steps); _f32 = fp32 at 8192 per GPU (config 4's share)
  pmc_summary.json  : per-kernel per-launch means of every counter collected in the separate --pmc passes, one section per pass group ("headline", "dec", "f32")
  ../traffic.json   : HBM bytes per launch of the dominant kernels, (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section),
                      matrix-core utilisation where the counters were collected, and the hash of the kernel sources the counters were taken at
                      (bench.py reports traffic_stale: true when the sources have changed since)
"""
import csv
import glob
import hashlib
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


def short(name):
    return name.replace("void ", "").split("(")[0]


stats = {}
for d, out in (("stats", "kernel_stats.csv"), ("stats_dec", "kernel_stats_dec.csv"), ("stats_roll", "kernel_stats_roll.csv"), ("stats_c3", "kernel_stats_c3.csv"),
               ("stats_f32", "kernel_stats_f32.csv"), ("stats_decloop", "kernel_stats_decloop.csv"), ("stats_hji", "kernel_stats_hji.csv")):
    ks = glob.glob(os.path.join(src, d, "*", "*_kernel_stats.csv"))
    # (gpurun MERGES a call's files into gpurun_out/: two campaigns leave two files per directory and their counters would be averaged together -- delete gpurun_out/prof first)
    assert len(ks) <= 1, f"{d}: files of more than one campaign in gpurun_out/prof -- delete it and run tools/gpu_profile.sh again"
    if len(ks) == 1:
        shutil.copy(ks[0], os.path.join(dst, out))
        stats[d] = list(csv.DictReader(open(ks[0])))
assert "stats" in stats, "PART=1 of tools/gpu_profile.sh has not run"

groups = {"headline": ["pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_mfma", "pmc_flops", "pmc_tcc", "pmc_wr", "pmc_wb"], "hji": ["pmc_fetch_hji", "pmc_write_hji"],
          "dec": ["pmc_fetch_dec", "pmc_write_dec", "pmc_sq_dec", "pmc_sq2_dec", "pmc_flops_dec"], "f32": ["pmc_fetch_f32", "pmc_write_f32", "pmc_sq_f32", "pmc_mfma_f32", "pmc_flops_f32"],
          "c3": ["pmc_fetch_c3", "pmc_write_c3", "pmc_sq_c3", "pmc_flops_c3"]}
summary = {}
for g, dirs in groups.items():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
        assert len(fs) <= 1, f"{d}: files of more than one campaign in gpurun_out/prof"
        for f in fs:
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if not k.startswith("pg::"):
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    if acc:
        summary[g] = {k: {c: {"launches": v[1], "mean_per_launch": v[0] / v[1]} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)


def kernel_source_sha16():
    """The hash of the kernel sources the counters were TAKEN at: bench.py prints it into its JSON line, and the profile's own bench log (gpurun_out/prof/bench_stats.log)
    carries that line -- the stamp is the collection's, not whatever the tree holds when this script runs.  Fallback (logs of an older bench.py): the tree's hash."""
    try:
        line = [l for l in open(os.path.join(src, "bench_stats.log")).read().splitlines() if l.startswith("{")][-1]
        return json.loads(line)["kernel_source_sha16"]
    except Exception:
        h = hashlib.sha256()
        for f in ("pg_kernels.hip", "pg_solve_lat.hip", "pg_device.hpp", "pg_api.hip"):
            h.update(open(os.path.join(ROOT, "pigeon.jl_amd", "csrc", f), "rb").read())
        return h.hexdigest()[:16]


def entry(group, stats_key, match):
    rows = stats.get(stats_key, [])
    cand = [r for r in rows if match in r["Name"]]
    S = summary.get(group, {})
    if not cand:
        return None
    dom = max(cand, key=lambda r: float(r["TotalDurationNs"]))
    dk = short(dom["Name"])
    if dk not in S or "FETCH_SIZE" not in S[dk] or "WRITE_SIZE" not in S[dk]:
        return None
    c = {k: v["mean_per_launch"] for k, v in S[dk].items()}
    e = {"hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024, "fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
         "avg_launch_ns_rocprof": float(dom["AverageNs"]), "min_launch_ns_rocprof": float(dom["MinNs"]), "max_launch_ns_rocprof": float(dom["MaxNs"]),
         "stddev_launch_ns_rocprof": float(dom.get("StdDev", 0.0) or 0.0), "launches_rocprof": int(dom["Calls"]), "source": f"profiles/{tag}/pmc_summary.json [{group}]"}
    if "SQ_INSTS_VALU" in c:
        # issue fraction: every VALU wave-instruction occupies its SIMD for 4 cycles (64 lanes on 16), 1024 SIMDs at 2.4 GHz over the launch
        e["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
        e["valu_issue_frac"] = c["SQ_INSTS_VALU"] * 4 / (1024 * float(dom["AverageNs"]) * 1e-9 * 2.4e9)
        for k_, n_ in (("SQ_INSTS_LDS", "lds_insts_per_launch"), ("SQ_INSTS_SALU", "salu_insts_per_launch"), ("SQ_INSTS_VMEM_RD", "vmem_rd_insts_per_launch"), ("SQ_INSTS_VMEM_WR", "vmem_wr_insts_per_launch"),
                       ("SQ_ACTIVE_INST_VALU", "active_inst_valu"), ("SQ_WAVE_CYCLES", "wave_cycles"), ("SQ_WAVES", "waves")):
            if k_ in c:
                e[n_] = c[k_]
    for prec, peak in (("F64", 78.6), ("F32", 157.3)):
        if f"SQ_INSTS_VALU_FMA_{prec}" in c:
            # the hardware's own count of executed arithmetic (round 6): wave-level instruction counts by type; one instruction = 64 lanes whatever EXEC holds, FMA = 2 flops.
            # Against the vector peak of that precision over the launch's average duration; also as a share of ALL counted VALU instructions
            ar = c.get(f"SQ_INSTS_VALU_ADD_{prec}", 0.0) + c.get(f"SQ_INSTS_VALU_MUL_{prec}", 0.0) + c.get(f"SQ_INSTS_VALU_TRANS_{prec}", 0.0) + c[f"SQ_INSTS_VALU_FMA_{prec}"]
            fl = 64.0 * (ar + c[f"SQ_INSTS_VALU_FMA_{prec}"])
            secs = float(dom["AverageNs"]) * 1e-9
            e["hw_flops"] = {"precision": prec.lower(), "add": c.get(f"SQ_INSTS_VALU_ADD_{prec}"), "mul": c.get(f"SQ_INSTS_VALU_MUL_{prec}"), "fma": c[f"SQ_INSTS_VALU_FMA_{prec}"], "trans": c.get(f"SQ_INSTS_VALU_TRANS_{prec}"),
                             "int32": c.get("SQ_INSTS_VALU_INT32"), "int64": c.get("SQ_INSTS_VALU_INT64"), "cvt": c.get("SQ_INSTS_VALU_CVT"),
                             "arith_insts_per_launch": ar, "valu_flops_per_launch": fl, "frac_of_vector_peak": fl / secs / 1e12 / peak, "peak_tflops": peak,
                             "arith_share_of_valu_insts": (ar / c["SQ_INSTS_VALU"]) if c.get("SQ_INSTS_VALU") else None, "source": f"profiles/{tag}/pmc_summary.json [{group}]"}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        # busy cycles of the matrix pipe (summed over SIMDs) over 1024 SIMDs x the launch at 2.4 GHz; fp64 MFMA flops (512 per MOPS unit: MI355X_MICROARCH.md) against the dense fp64 peak
        secs = float(dom["AverageNs"]) * 1e-9
        mops64 = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
        e["mfma"] = {"busy_cycles_per_launch": c["SQ_VALU_MFMA_BUSY_CYCLES"], "sq_busy_cycles_per_launch": c["SQ_BUSY_CYCLES"],
                     "busy_frac_of_simd_time": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * secs * 2.4e9), "mfma_instructions_per_launch": c.get("SQ_INSTS_MFMA"),
                     "mops_f64_per_launch": mops64, "tflops_f64_executed": mops64 * 512 / secs / 1e12, "peak_tflops_f64_matrix": 78.6,
                     "frac_of_peak": mops64 * 512 / secs / 1e12 / 78.6, "source": f"profiles/{tag}/pmc_summary.json [{group}]"}
    return dk, e


out = {"formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction for wide coalesced reads, MI355X_MICROARCH.md HBM section; separate --pmc passes)",
       "workload": "bench.py B=4096 coupled fp64 cold (k_solve, k_nodes_linearize); 2^20 lookups on the 13x13x9^5 grid (k_hji_lookup); config 5 = lateral N = 50 + walls (k_solve_lat)",
       "kernel_source_sha16": kernel_source_sha16(), "kernels": {}}
for group, sk, m in (("headline", "stats", "pg::k_solve<false, false, false, false"), ("headline", "stats", "pg::k_solve<false, false, false, true"), ("hji", "stats_hji", "pg::k_hji_lookup<3>"), ("hji", "stats_hji", "pg::k_hji_lookup<5>"), ("hji", "stats_hji", "pg::k_hji_lookup<7>"),
                     ("headline", "stats", "pg::k_linearize"), ("headline", "stats", "pg::k_nodes"),
                     ("dec", "stats_dec", "pg::k_solve_lat<1, true, true, 16, 1>"), ("dec", "stats_dec", "pg::k_solve_lat<1, true, false, 64, 2>"), ("dec", "stats_dec", "pg::k_solve_lat<1, false, true, 16, 1>"), ("dec", "stats_dec", "pg::k_solve_lat<1, false, false, 64, 2>"), ("dec", "stats_dec", "pg::k_qp_dec"), ("dec", "stats_dec", "pg::k_nodes_dec")):
    t = entry(group, sk, m)
    if t:
        out["kernels"][t[0]] = t[1]
        if m == "pg::k_solve<false, false, false, false":
            out.update({"kernel": t[0], **{k: v for k, v in t[1].items() if k != "mfma"}})
# the fp32 library's kernels (same names): config 3 (4096 + safety row: group "c3") first, config 4's share (8192: group "f32") for what c3 does not carry
out["kernels_f32"] = {}
for group, sk, m in (("c3", "stats_c3", "pg::k_solve<false, false, false, true"), ("c3", "stats_c3", "pg::k_nodes"), ("c3", "stats_c3", "pg::k_hji_lookup"), ("f32", "stats_f32", "pg::k_solve<false, false, false, false"), ("f32", "stats_f32", "pg::k_nodes")):
    t = entry(group, sk, m)
    if t and t[0] not in out["kernels_f32"]:
        out["kernels_f32"][t[0]] = dict(t[1], workload="config 3 (B = 4096, fp32, safety row)" if group == "c3" else "fp32 at 8192 per GPU (config 4's share)")
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for sk, rows in stats.items():
    print("==", sk)
    for r in rows[:8]:
        print(f"{short(r['Name'])[:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']:>6s} %")
print(json.dumps({k: v.get("mfma") for k, v in out["kernels"].items()}, indent=1))
