#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W : MPC QP solves/sec on N MI355X (one process per GPU, RCCL gather of controls).

A "step" = the reference's whole per-timestep callback body (compute_time_steps! -> compute_linearization_nodes! -> update_QP! ->
solve! -> get_next_control; /root/reference/src/ros_integration.jl:96-99,124) for a batch of B = 4096 independent instances per GPU
(BASELINE.json configs[1]: coupled MPC, N = 30, X1, randomised x0 along the skidpadoval test path, fp64), COLD (solved = false),
inputs resident in HBM before the timed region.  Weak scaling: every rank owns its own 4096 instances (seed 12345 + rank).

`python bench.py --gpus N` started as a plain process launches its N ranks itself (torch.distributed.run as a child, before any GPU call).
The JSON line also carries the other BASELINE configs and SURVEY 8(d) items as secondary objects (rank 0, N = 1 only where they need the CPU):
  roofline (+ roofline.valu: algorithmic-FLOP fraction of the fp64 vector peak from live iteration counts), iteration histograms,
  hji_lookup (7-D grid, and the 4-D variant), decoupled_n50 (config 5), fp32 (config 3), cpu_baseline (OSQP-port on the host cores for the
  headline and the decoupled config, config 1 = single-instance closed loop GPU vs CPU, measured accuracy against the exact optimum).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_VALU_PEAK_TF = 78.6                  # MI355X fp64 vector peak (256 CUs x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz); fp32: 157.3
BYTES_PER_SOLVE_COLD_FP64 = 112           # state 6 + control 3 + t0 in, control 3 + status + iters out

# Algorithmic FLOPs of k_solve per instance (FMA = 2; derivation term by term in EXPERIMENTS.md 6): stage-structured interior point with n = 8 states,
# m = 2 inputs, 16 rows per stage.  Per stage: matrix pass 2480 (P [A B c] 1080, B'PB + B'PA 250, S^-1 and K 80, Q + A'PA + F'K 1070),
# one vector pass 180, one roll-out 160, one half-iteration of stage work (assemble + Newton point + step rule) 400.
def solve_flops(N, ipm_iters, polish_rounds):
    per_iter = N * (2480 + 2 * 180 + 2 * 160 + 2 * 400)          # one predictor-corrector iteration: 1 matrix pass, 2 vector passes, 2 roll-outs
    per_polish = N * (2480 + 180 + 160 + 400) + 0.3 * N * (180 + 160 + 400)   # first polish solve (+ a refinement for ~30 % of the rounds)
    return 6000.0 + per_iter * ipm_iters + per_polish * polish_rounds


def kernel_source_sha16():
    """Content hash of the device sources (what profiles/traffic.json was measured on is recorded with the same function by tools/summarize_profiles.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("pg_kernels.hip", "pg_solve_lat.hip", "pg_device.hpp", "pg_api.hip"):
        h.update(open(os.path.join(ROOT, "pigeon.jl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


# Algorithmic FLOPs of k_solve_lat per instance and pass through its loop (an interior-point iteration or a polish round; FMA = 2), 5-state stage, NR rows per stage:
# matrix pass 2 x (P [A B c] 175 + B'M 35 + Q + A'M + F'K 175 + gains 15) = 800, vector pass 2 x 45 = 90, two roll-outs 2 x 2 x 42 = 170, and the three stage-parallel
# passes (barrier terms and slack elimination, Newton point and slacks, step rules and update): ~70 flops per row and pass.
def lat_flops(N, NR, passes):
    return passes * N * (800.0 + 90.0 + 170.0 + 3 * 70.0 * NR)


def linearize_flops_from_isa(B, Ns, Nl, precision):
    """(flops of one linearisation launch, its instruction totals) from profiles/isa_mix.json (tools/isa_mix.py: static counts of the kernel's loop nests x trip counts; per lane =
    per (instance, interval)).  Falls back to the hand count of the same algorithm (832 / 699 flops per ramp / zero-order-hold stage, 40 stages per interval) without the file."""
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", "isa_mix.json")))
        m = mix["kernels"]["k_nodes_linearize<true, 1, 1>" + ("" if precision == "f64" else " [f32]")]
        tot = {k: B * (Ns * m["zoh"]["per_interval"][k] + Nl * m["ramp"]["per_interval"][k]) for k in ("valu", "agpr_moves", "arith", "flops")}
        isa = {"valu_wave_insts": tot["valu"] / 64.0, "agpr_move_wave_insts": tot["agpr_moves"] / 64.0, "arith_wave_insts": tot["arith"] / 64.0,
               "flops_per_stage": {"ramp": m["ramp"]["per_stage"]["flops"], "zoh": m["zoh"]["per_stage"]["flops"]}, "stale": mix.get("kernel_source_sha16") != kernel_source_sha16()}
        return float(tot["flops"]), isa
    except Exception:
        return float(B) * 40.0 * (Ns * 699.0 + Nl * 832.0), None


def roofline_consistency(line):
    """What must hold between the fractions of a bench record whatever the numbers are: a kernel cannot execute more arithmetic than it issues instructions for.  Returns the
    list of violations (empty = consistent); bench.py attaches it to the record, tests/test_abi_and_host.py asserts it empty on the committed record of this round."""
    bad = []
    for e in ((line.get("roofline") or {}).get("kernels") or []):
        issue = e.get("valu_issue_frac")
        for k in ("valu_flop_frac", "executed_fp64_frac", "hw_valu_flop_frac"):
            if issue is not None and e.get(k) is not None and e[k] > issue:
                bad.append(f"{e.get('kernel')}: {k} {e[k]:.3f} > valu_issue_frac {issue:.3f}")
    return bad


def roofline_of(tr, kernel_match, ms, units, bytes_per_unit, flops, precision, note=None):
    """`roofline` object of a secondary benchmark line: HBM fraction from the algorithmic bytes and the live duration; counter traffic, VALU issue fraction (SQ_INSTS_VALU x 4
    cycles over the SIMD time of the launch) and the rocprof duration spread from the committed PMC / stats passes (profiles/traffic.json); flop-model fraction of the vector peak."""
    achieved = units * bytes_per_unit / (ms * 1e-3) / 1e9
    peak_tf = FP64_VALU_PEAK_TF if precision == "f64" else 2 * FP64_VALU_PEAK_TF
    r = {"bound": "hbm", "kernel": kernel_match, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "avg_launch_ms": ms,
         "algorithmic_bytes_per_launch": units * bytes_per_unit, "traffic": None,
         "valu": {"bound": "valu-" + precision, "algorithmic_flops_per_launch": flops, "achieved": flops / (ms * 1e-3) / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                  "frac": flops / (ms * 1e-3) / 1e12 / peak_tf}}
    matches = [kernel_match] if isinstance(kernel_match, str) else list(kernel_match)      # (several: the launches a phase is made of -- k_solve_lat's first launch + the resuming one)
    r["kernel"] = matches[0] if len(matches) == 1 else " + ".join(matches)
    hits = [[v for k, v in (tr or {}).get("kernels", {}).items() if m in k] for m in matches]
    if hits and all(hits):
        es = [h[0] for h in hits]
        r["traffic"] = sum(e.get("hbm_bytes_per_launch") or 0.0 for e in es); r["traffic_source"] = es[0].get("source")
        for k in ("valu_insts_per_launch", "avg_launch_ns_rocprof"):
            if all(k in e for e in es):
                r[k] = sum(e[k] for e in es)
        if len(es) == 1:
            for k in ("valu_issue_frac", "min_launch_ns_rocprof", "max_launch_ns_rocprof", "stddev_launch_ns_rocprof", "launches_rocprof"):
                if k in es[0]:
                    r[k] = es[0][k]
        else:
            r["launches"] = [{"kernel": m, **{k: e.get(k) for k in ("hbm_bytes_per_launch", "avg_launch_ns_rocprof", "valu_issue_frac", "launches_rocprof")}} for m, e in zip(matches, es)]
            if all("valu_issue_frac" in e and "avg_launch_ns_rocprof" in e for e in es):      # SIMD-time-weighted
                r["valu_issue_frac"] = sum(e["valu_issue_frac"] * e["avg_launch_ns_rocprof"] for e in es) / sum(e["avg_launch_ns_rocprof"] for e in es)
    if note:
        r["note"] = note
    return r


def mfma_util(tr, dom, pipelined):
    """Matrix-core utilisation of the dominant kernel from the committed PMC pass (profiles/traffic.json, written by tools/summarize_profiles.py): busy cycles of the MFMA
    pipe over the kernel's busy cycles, and the fp64 MFMA flop rate against the 78.6 TFLOP/s dense fp64 matrix peak.  None when the counters were not collected."""
    if not tr:
        return None
    kern = {0: "k_nodes_linearize" if pipelined else "k_nodes", 1: "k_linearize", 2: "k_solve"}[dom]
    hit = [v for k, v in tr.get("kernels", {}).items() if kern in k and "mfma" in v]
    return hit[0]["mfma"] if hit else None


def detect_xgmi():
    """Link type between the GPUs of this node from `rocm-smi --showtopotype` (a child process: nothing here touches the GPU): True when every GPU pair is XGMI,
    False when some pair is not (PCIe), "unknown" when the tool is missing or there is one GPU."""
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "--showtopotype"], capture_output=True, text=True, timeout=30).stdout
    except Exception:
        return "unknown"
    kinds = [w for ln in out.splitlines() if ln.startswith("GPU") for w in ln.split()[1:] if w in ("XGMI", "PCIE")]
    if not kinds:
        return "unknown"
    return all(k == "XGMI" for k in kinds)


XGMI = "unknown"


def rccl_version(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        return None


def usable_cores():
    """Host threads this process may really use: the scheduler affinity, capped by the cgroup CPU quota (cpu.max) when one is set."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def hist(a):
    v, c = np.unique(np.asarray(a), return_counts=True)
    return {str(int(k)): int(n) for k, n in zip(v, c)}


def host_cpu_model():
    """Model name of the host CPU (the cores the `cpu_baseline` ran on)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def oracle_build_flags():
    """Compiler and flags the CPU baseline (oracle/liboracle.so) was built with: read from oracle/Makefile, the recipe build() runs."""
    try:
        txt = open(os.path.join(ROOT, "oracle", "Makefile")).read()
        cxx = [ln.split("?=", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("CXX ?=")]
        fl = [ln.split("?=", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("CXXFLAGS ?=")]
        return (cxx[0] if cxx else "g++") + " " + (fl[0] if fl else "")
    except Exception:
        return None


def pg_environment():
    """PG_* / PIGEON_* variables set in this process (diagnostic switches of a -DPG_DIAG build, library overrides): a run with any of them is not the shipped configuration."""
    return sorted(f"{k}={v}" for k, v in os.environ.items() if k.startswith("PG_") or k.startswith("PIGEON_"))


def _r(x, sig=5):
    """Round floats to `sig` significant digits (the compact line); everything else unchanged."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


COMPACT_LIMIT = 4096


def compact_line(full):
    """The LAST stdout line of bench.py: the contract keys only, strict JSON, <= 4 KB (the driver keeps an 8 KB tail of stdout: round 4's 20 KB line came back `parsed: null`).
    `full` is the whole record (written to bench_full.json).  Pure function of `full`: tests/test_abi_and_host.py runs it on canned records."""
    g = full.get
    rf = full.get("roofline") or {}
    roof = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "avg_launch_ms") if k in rf}
    if rf.get("kernels"):
        roof["kernels"] = rf["kernels"]
    if rf.get("hji_lookup"):
        roof["hji_lookup"] = rf["hji_lookup"]
    cb = full.get("cpu_baseline")
    cpu = None
    if cb:
        cpu = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "value_1thread", "cpu_model", "flags") if k in cb}
        cpu["sample"] = str(cb.get("sample", ""))[:160]
        acc = cb.get("accuracy") or {}
        if acc:
            cpu["gpu_vs_oracle_max_abs_applied_control_error"] = acc.get("max_abs_applied_control_error")
            cpu["identical_active_set_lists"] = acc.get("identical_active_set_lists")
    cfg = full.get("config") or {}
    line = {"metric": g("metric"), "value": g("value"), "unit": g("unit"), "n_gpus": g("n_gpus"), "steps": g("steps"), "warmup": g("warmup"), "ms_per_step": g("ms_per_step"),
            "higher_is_better": True, "scaling": g("scaling"), "vs_baseline": g("vs_baseline"), "dtype": g("dtype"), "data": g("data"),
            "config": {k: (str(cfg[k])[:200] if isinstance(cfg[k], str) else cfg[k]) for k in ("workload", "batch_per_gpu", "parallelism") if k in cfg},
            "solved": g("solved"), "roofline": roof}
    if cpu is not None:
        line["cpu_baseline"] = cpu
    if full.get("timing"):
        tm = full["timing"]
        line["timing"] = {k: tm.get(k) for k in ("repeats", "ms_per_step_min", "ms_per_step_median", "ms_per_step_max")}
    for k in ("phase_ms_short", "pipeline_fallbacks", "env", "kernel_source_sha16", "ranks", "collective", "per_rank_ms_per_step", "gather_ms", "rccl_version", "xgmi", "gather_ok", "roofline_consistency", "secondary"):
        if full.get(k) is not None:
            line[k] = full[k]
    line["full_record"] = "bench_full.json"
    line = _r(line)
    txt = json.dumps(line, allow_nan=False)
    # never past the limit: drop the optional keys in order of least importance (the contract keys always fit)
    for k in ("secondary", "phase_ms_short", "kernel_source_sha16", "env"):
        if len(txt) <= COMPACT_LIMIT:
            break
        line.pop(k, None)
        txt = json.dumps(line, allow_nan=False)
    if len(txt) > COMPACT_LIMIT and "kernels" in line["roofline"]:
        line["roofline"].pop("kernels"); txt = json.dumps(line, allow_nan=False)
    return txt


def _json_safe(x):
    """NaN / Inf -> null (strict JSON), numpy scalars -> Python."""
    if isinstance(x, dict):
        return {str(k): _json_safe(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_json_safe(v) for v in x]
    if isinstance(x, (np.floating, float)):
        x = float(x)
        return x if x == x and abs(x) != float("inf") else None
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.bool_,)):
        return bool(x)
    return x


def emit(full, path=None):
    """Write the whole record to bench_full.json (+ gpurun_out/ when that scratch directory exists, so that a gpurun call brings it home; `path`: there and nowhere else) and
    print the compact line LAST on stdout."""
    full = _json_safe(full)
    blob = json.dumps(full, allow_nan=False)
    targets = [path] if path else [os.path.join(d, "bench_full.json") for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)]
    for t in targets:
        try:
            with open(t, "w") as f:
                f.write(blob + "\n")
        except OSError:
            pass
    print(compact_line(full), flush=True)


def phases_of(m, n=3, cold=True, u_ptr=None):
    """Per-phase device time of a step of handle `m` (median over n steps, HIP events on the launch stream): the library's instrumentation (option "phase_timing": four event
    records per step, 13-25 us of stream time) is switched on for these steps only -- the timed loops of this file run without it, as the library does by default."""
    import torch
    m.set_option("phase_timing", 1)
    ph = []
    for _ in range(n):
        if cold:
            m.reset()
        m.step_dev(u_ptr); torch.cuda.synchronize(); ph.append(m.phase_ms())
    m.set_option("phase_timing", 0)
    return [float(v) for v in np.median(np.array(ph), axis=0)]


def cpu_baseline(pkg, traj, mpc, state, control, t0, toff, precision, local):
    """The reference ALGORITHM (OSQP-form ADMM with sparse LDL', default settings) on the host cores: oracle 'port'.
    Only this leg may touch oracle/: as the timed CPU baseline, and as the CHECKER of the GPU numbers (accuracy sample, config 1)."""
    from oracle import oracle as orc_mod
    cores = usable_cores()
    orc = orc_mod.Oracle(); orc.set_trajectory(traj.data)
    n1 = 128
    _, _, it1, st1, secs1 = orc.step_batch(state[:n1], control[:n1], t0[:n1], time_offsets=toff[:n1], solver=1, nthreads=1)
    orc2 = orc_mod.Oracle(); orc2.set_trajectory(traj.data)
    nall = min(len(t0), max(512, 128 * cores))
    _, _, it, st, secs = orc2.step_batch(state[:nall], control[:nall], t0[:nall], time_offsets=toff[:nall], solver=1, nthreads=cores)
    out = {"value": nall / secs, "unit": "solves/s", "cores": cores, "kind": "port",
           "sample": f"{nall} cold instances of the same workload on {cores} host threads (OSQP-port ADMM, eps 1e-3, mean {float(np.mean(it)):.0f} iterations); "
                     f"1 thread: {n1 / secs1:.1f} solves/s on {n1} instances",
           "value_1thread": n1 / secs1, "cpu_model": host_cpu_model(), "flags": oracle_build_flags(), "osqp_iters_hist": hist(it),
           "note": "Julia reference not run (no Julia toolchain; third-party sources absent): C++ restatement of the same algorithm"}

    # ---- checker: measured accuracy of the GPU batch against the exact optimum of the same QP data (sample of the headline batch) ----
    ns = 256
    mpc.reset(); mpc.step_dev(None); mpc.synchronize()
    qp = mpc.qp_data(0, ns); x, _ = mpc.solution(); _, _, act, _ = mpc.solve_info(); lam = mpc.multipliers()
    e2, ea, same = [], [], 0
    for b in range(ns):
        xe, ye, info = orc.solve_exact(qp[b]); X = orc.split_x(xe)
        e2.append(float(np.max(np.abs(x[b, 1, 6:] - X["u"][1])))); ea.append(float(np.max(np.abs(x[b, :, 6:] - X["u"]))))
        same += int(mpc.canonical_active_set(b, act[b], qp[b], lam=lam[b]) == orc_mod.active_set(orc.assemble_qp(qp[b]), xe, ye, tol=1e-6))
    out["accuracy"] = {"instances": ns, "max_abs_applied_control_error": max(e2), "median": float(np.median(e2)), "max_abs_any_control_error": max(ea),
                       "identical_active_set_lists": f"{same}/{ns}", "against": "exact optimum of the same QP data (oracle sparse IPM + polish), controls normalised; "
                       + ("tolerance 1e-6" if precision == "f64" else "fp32 arithmetic")}

    # ---- BASELINE config 5 on the CPU: decoupled lateral MPC, N = 50 (the reference's QP as it stands: no wall rows) ----
    try:
        od = orc_mod.OracleDecoupled(N_short=10, N_long=40); od.set_trajectory(traj.data)
        nd = min(len(t0), max(256, 32 * cores))
        _, itd, std, secsd = od.step_batch(state[:nd], control[:nd], t0[:nd], toff[:nd], nthreads=cores)
        out["decoupled_n50"] = {"value": nd / secsd, "unit": "solves/s", "cores": cores, "sample": f"{nd} cold instances", "osqp_iters_mean": float(np.mean(itd)),
                                "osqp_hit_max_iter": int(np.sum(std != 1))}
        # ... and the checker for the bench's decoupled_n50 line: the GPU's applied steering of a sample of the SAME batch, default solver configuration, with the wall
        # rows (the configs[4] line) and without, against a VERIFIED KKT point of the canonical QP built from the GPU's own QP data (OracleDecoupled.solve_exact_verified)
        unembed_qp, extend_with_walls = orc_mod.unembed_qp, orc_mod.extend_with_walls
        nsd = 128
        acc = {}
        for walls in (True, False):
            g = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, nsd, N_short=10, N_long=40, device=local, precision=precision, walls=walls)
            ud, stdg, _ = g.step_(state[:nsd], control[:nsd], t0[:nsd], time_offset=toff[:nsd])
            qpd = g.qp_data(); xd, _ = g.solution(); edges = g.wall_edges() if walls else None; pold = g.polish_info()
            errs, nver = [], 0
            for b in range(nsd):
                sd = unembed_qp(od, qpd[b])
                if walls:
                    qpw, _ = extend_with_walls(od, od.assemble_qp(sd), edges[b], od.unpack_sd(sd)["dt"], g.wall_weight)
                    xe, ye, info = od.solve_exact_verified(sd, qp=qpw, walls=edges[b], wall_weight=g.wall_weight)
                else:
                    xe, ye, info = od.solve_exact_verified(sd)
                if info["status"] == 1 and info["polished"] >= 1:
                    nver += 1; errs.append(abs(xd[b, 1, 6] - od.split_x(xe[:od.n])["delta"][1]))
            acc["with_walls" if walls else "without_walls"] = {"instances": nsd, "oracle_verified_kkt_points": nver, "max_abs_applied_steering_error_rad": float(np.max(errs)) if errs else None,
                                                               "median": float(np.median(errs)) if errs else None, "gpu_verified_by_polish": f"{int((pold >= 1).sum())}/{nsd}",
                                                               "gpu_solved": f"{int(pkg.is_solved(stdg).sum())}/{nsd}"}
            g.close()
        out["decoupled_n50"]["accuracy"] = dict(acc, against="exact optimum of the same QP data as a verified KKT point of the canonical lateral QP (fp64 oracle); every one of "
                                                "the 4096 instances <= 1e-6 is asserted by tests/test_gpu_decoupled.py::test_config5_as_shipped_every_instance_against_the_oracle")
    except Exception as e:          # never let a secondary object cost the headline
        out.setdefault("decoupled_n50", {})["error"] = repr(e)

    # ---- BASELINE config 1: ONE controller, cold step then 100 closed-loop steps (simulate semantics, 10 ms period), pg_step host->host
    #      latency on the GPU beside the CPU port's step time; x0 = path pose at s = 20 m, e = 0.3 m, dpsi = 0.05 rad, Ux = V_path ----
    c1 = {}
    for path in ["vail", "skidpadoval"]:
        tj = pkg.load_path_fixture(path)
        E, Nn, psi, kappa, V, t = pkg.synthetic.path_pose(tj, 20.0)
        q0 = np.array([E - 0.3 * np.cos(psi), Nn - 0.3 * np.sin(psi), psi + 0.05, V, 0.0, 0.0]); u0 = np.zeros(3)
        o1 = orc_mod.Oracle(); o1.set_trajectory(tj.data)
        g = pkg.BatchedTrajectoryTrackingMPC(tj, 1, device=local, precision=precision, phase_timing=False)
        res = {}
        for who in ("gpu", "cpu"):
            q, u, tt = q0.copy(), u0.copy(), float(t)
            lat = []
            for k in range(101):
                t_ = time.perf_counter()
                if who == "gpu":
                    un, st_, _ = g.step_(q[None], u[None], np.array([tt]), time_offset=np.array([0.0])); un = un[0]
                else:
                    un, _, _, st_, _ = o1.step_batch(q[None], u[None], np.array([tt]), time_offsets=np.array([0.0]), solver=1, nthreads=1); un = un[0]
                lat.append(time.perf_counter() - t_)
                q = o1.plant_step(q, u, 0.01); u = un; tt += 0.01
            s_, e_, _, _ = o1.path_coordinates(q[0], q[1])
            res[who] = {"cold_step_ms": 1e3 * lat[0], "closed_loop_ms_per_step_mean": 1e3 * float(np.mean(lat[1:])), "closed_loop_ms_per_step_max": 1e3 * float(np.max(lat[1:])),
                        "final_lateral_error_m": float(e_), "final_speed_mps": float(q[3])}
        g.close()
        c1[path] = res
    out["config1_single_instance"] = {"workload": "configs[0]: one coupled controller, N = 30, cold step + 100 closed-loop steps at 10 ms; gpu = pg_step host->host "
                                                  "(launch + PCIe included), cpu = OSQP port with warm start, 1 thread", **c1}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hji", action="store_true")
    ap.add_argument("--no-decoupled", action="store_true")
    ap.add_argument("--repeats", type=int, default=5, help="the timed region is repeated this many times (each: --steps steps between two barrier + synchronize brackets); "
                                                            "`value` is the MEDIAN block, min / max / every block go to bench_full.json (VERDICT r5: one 10 ms shot is thin as a measurement)")
    ap.add_argument("--no-f32", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the closed-loop rollout object")
    ap.add_argument("--no-warm", action="store_true", help="skip the warm-step loop (the profile campaign uses it so that every k_solve launch of the trace is a cold headline launch)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend for --gpus > 1: nccl = RCCL over xGMI (the product path); gloo = host-staged gather with the ranks sharing whatever GPUs exist "
                         "(test mode: exercises launch -> shard -> step -> gather end to end on a 1-GPU box)")
    ap.add_argument("--full-record", default=None, help="where the whole record goes (default: bench_full.json beside bench.py, and in gpurun_out/ when that exists); the last stdout line is the compact one either way")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64",
                    help="arithmetic type of the headline run: f64 = BASELINE configs[1] (default, the metric's config); f32 with --batch 8192 --gpus 8 = configs[3]")
    args = ap.parse_args()

    # `python bench.py --gpus N` (N > 1) started as a plain process: launch the N ranks ourselves, one process per GPU, through torch.distributed.run as a
    # CHILD process -- decided before anything here touches the GPU (no HIP call, no torch.cuda call has happened yet; a process that initialised the GPU
    # must never exec another program on this pool).  Rank 0 of the children prints the JSON line; this parent only forwards the exit code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk_:
            sk_.bind(("127.0.0.1", 0)); port = sk_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

    global XGMI
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) == 0:
        XGMI = detect_xgmi()          # (before this process initialises the GPU)
    import torch
    import torch.distributed as dist
    from __graft_entry__ import _load_pkg
    pkg = _load_pkg()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()          # test mode: ranks may share a GPU
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    B = args.batch
    try:
        TR = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        TR = None
    traj = pkg.load_path_fixture("skidpadoval")
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, device=local, precision=args.precision, phase_timing=False)      # (the library's default: no per-phase events on the stream)
    npdt = np.float64 if args.precision == "f64" else np.float32; tdt = torch.float64 if args.precision == "f64" else torch.float32
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345 + rank, traj_mode=True)
    dev = torch.device("cuda", local)
    d_state = torch.from_numpy(state.astype(npdt)).to(dev); d_control = torch.from_numpy(control.astype(npdt)).to(dev)
    d_t0 = torch.from_numpy(t0).to(dev); d_toff = torch.from_numpy(toff).to(dev)            # absolute time is double in both builds
    # controls of a step and their gather are DOUBLE-BUFFERED (N > 1): the all_gather of step i runs on a stream of its own under the kernels of step i + 1 (the
    # collective moves 96 KB per rank over xGMI: latency, which the next step hides); a buffer is reused only after the gather that read it has finished
    u_bufs = [torch.zeros(B, 3, dtype=tdt, device=dev) for _ in range(2 if world > 1 else 1)]
    g_bufs = [torch.zeros(world * B, 3, dtype=tdt, device=dev) for _ in range(2)] if world > 1 else None
    u_out = u_bufs[0]; gathered = g_bufs[0] if world > 1 else None
    main_stream = torch.cuda.current_stream()
    comm_stream = torch.cuda.Stream(device=dev) if (world > 1 and args.backend == "nccl") else None
    ev_done = [torch.cuda.Event() for _ in range(2)]; ev_free = [torch.cuda.Event() for _ in range(2)]; used = [False, False]
    step_no = [0]
    mpc.set_stream(main_stream.cuda_stream)
    mpc.set_inputs_dev(B, d_state.data_ptr(), d_control.data_ptr(), d_t0.data_ptr(), None, d_toff.data_ptr())

    def one_step(cold=True):
        nonlocal u_out, gathered
        k = step_no[0] & 1 if world > 1 else 0
        step_no[0] += 1
        u_out = u_bufs[k]
        if cold:
            mpc.reset()                                   # solved = false for every instance (hipMemsetAsync on the same stream)
        if comm_stream is not None and used[k]:
            main_stream.wait_event(ev_free[k])            # the gather of step i - 2 has read this buffer
        mpc.step_dev(u_out.data_ptr())
        if world > 1:
            gathered = g_bufs[k]
            if args.backend == "nccl":
                ev_done[k].record(main_stream)
                comm_stream.wait_event(ev_done[k])
                with torch.cuda.stream(comm_stream):
                    dist.all_gather_into_tensor(gathered, u_out)  # RCCL over xGMI: the only collective on the path
                    ev_free[k].record(comm_stream)
                used[k] = True
            else:
                g_host = torch.empty(world * B, 3, dtype=tdt)
                dist.all_gather_into_tensor(g_host, u_out.cpu())
                gathered.copy_(g_host)

    def sync():
        torch.cuda.synchronize()                          # (every stream of the device: the last gathers on the communication stream included)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    # the timed region: EXACTLY --steps steps between two (barrier + synchronize) brackets, taken --repeats times back to back; `value` is the median block
    # (N > 1: a block counts with the time of its slowest rank)
    blocks = []
    for _ in range(max(1, args.repeats)):
        sync()
        t_begin = time.perf_counter()
        for _ in range(args.steps):
            one_step()
            # no host sync inside the loop: events are read after the region
        torch.cuda.synchronize()
        sync()
        blocks.append(time.perf_counter() - t_begin)
    rank_ms = None; gather_ms = None
    if world > 1:
        # a scaling run is the timed loop, the gather check and ONE JSON line: every secondary object below is skipped when world > 1 (rank 0 would otherwise build
        # HJI tables, decoupled and fp32 handles while the other ranks wait in the barrier)
        mine = torch.tensor(blocks, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = np.array([[float(v) for v in t.tolist()] for t in every])            # [rank][block]
        blocks = [float(v) for v in per_rank.max(axis=0)]
        mid = int(np.argsort(blocks)[len(blocks) // 2])
        rank_ms = {"min": 1e3 * float(per_rank[:, mid].min()) / args.steps, "max": 1e3 * float(per_rank[:, mid].max()) / args.steps}
        # how long the collective itself takes (so that a weak-scaling efficiency below 0.9 can be attributed: launch overhead or the gather): a few more steps with an
        # event pair around each all_gather on the communication stream, outside the timed region
        if comm_stream is not None:
            g_ev = []
            for _ in range(6):
                one_step()
                k_ = (step_no[0] - 1) & 1
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(comm_stream):
                    e0.record(comm_stream); dist.all_gather_into_tensor(g_bufs[k_], u_bufs[k_]); e1.record(comm_stream)
                g_ev.append((e0, e1))
            sync()
            mine_g = torch.tensor([float(np.median([a.elapsed_time(b) for a, b in g_ev[1:]]))], dtype=torch.float64, device=dev)
            every_g = [torch.zeros_like(mine_g) for _ in range(world)]
            dist.all_gather(every_g, mine_g)
            gather_ms = {"per_rank_median": [float(t.item()) for t in every_g], "bytes_per_rank": int(u_out.numel() * u_out.element_size()),
                         "how": "HIP events on the communication stream around an all_gather_into_tensor of one step's controls, 5 launches outside the timed region"}
    elapsed = float(np.median(blocks))
    timing = {"repeats": len(blocks), "steps_per_block": args.steps, "block_s": blocks, "value_is": "median block", "ms_per_step_min": 1e3 * min(blocks) / args.steps,
              "ms_per_step_median": 1e3 * elapsed / args.steps, "ms_per_step_max": 1e3 * max(blocks) / args.steps}

    # per-phase device time (HIP events recorded by pg_step_dev on the launch stream): mean over a few extra steps outside the timed region
    mpc.set_option("phase_timing", 1)
    ph = []
    for _ in range(5):
        one_step(); torch.cuda.synchronize(); ph.append(mpc.phase_ms())
    mpc.set_option("phase_timing", 0)
    ph = np.mean(np.array(ph), axis=0)
    st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info()
    ok = int(pkg.is_solved(st).sum())
    # warm steps (second and later consecutive steps on the same inputs: warm nodes + warm start of the active set): reported as an extra, not as `value`
    sync(); tw = time.perf_counter()
    for _ in range(0 if args.no_warm else args.steps):
        one_step(cold=False)
    sync(); warm_elapsed = time.perf_counter() - tw
    # the same cold workload with the active-set guess OFF (every instance through the interior point; round-2-mid behaviour), as a reference for what the guess buys
    ipm_only = None; fused_line = None; per_phase_line = None
    if rank == 0 and world == 1 and not args.no_warm:
        m0 = pkg.BatchedTrajectoryTrackingMPC(traj, B, device=local, precision=args.precision, cold_guess=0, phase_timing=False)
        m0.set_stream(torch.cuda.current_stream().cuda_stream)
        m0.set_inputs_dev(B, d_state.data_ptr(), d_control.data_ptr(), d_t0.data_ptr(), None, d_toff.data_ptr())
        for _ in range(2):
            m0.reset(); m0.step_dev(u_out.data_ptr())
        torch.cuda.synchronize(); t_ = time.perf_counter()
        for _ in range(args.steps):
            m0.reset(); m0.step_dev(u_out.data_ptr())
        torch.cuda.synchronize(); t_ = time.perf_counter() - t_
        st0_, it0_, _, _ = m0.solve_info(); p0_ = m0.polish_info()
        ipm_only = {"value": B * args.steps / t_, "unit": "solves/s", "ms_per_step": 1e3 * t_ / args.steps, "phase_ms": phases_of(m0, u_ptr=u_out.data_ptr()),
                    "solved": f"{int(pkg.is_solved(st0_).sum())}/{B}", "ipm_iters_mean": float(np.mean(it0_)), "polish_rounds_hist": hist(p0_),
                    "config": "pg_config.cold_guess = 0: Mehrotra interior point to mu <= 3e-6 + active-set polish for every instance"}
        m0.close()
        # ... and with update_QP! fused into the solve kernel (pg_set_fusion(1), off by default: SURVEY 7.1 step 6; bit-identical results)
        mpc.set_fusion(True)
        for _ in range(2):
            mpc.reset(); mpc.step_dev(u_out.data_ptr())
        torch.cuda.synchronize(); t_ = time.perf_counter()
        for _ in range(args.steps):
            mpc.reset(); mpc.step_dev(u_out.data_ptr())
        torch.cuda.synchronize(); t_ = time.perf_counter() - t_
        fused_line = {"value": B * args.steps / t_, "unit": "solves/s", "ms_per_step": 1e3 * t_ / args.steps, "phase_ms": phases_of(mpc, u_ptr=u_out.data_ptr()),
                      "config": "pg_set_fusion(1): the wave that solves an instance linearises it first (one kernel for update_QP! + solve!); off by default"}
        mpc.set_fusion(0)
        # ... and with one launch per phase (pg_set_pipeline(0): nodes, update_QP!, solve as three kernels; the default pipelines the first two, bit-identical in fp64)
        mpc.set_pipeline(0)
        for _ in range(2):
            mpc.reset(); mpc.step_dev(u_out.data_ptr())
        torch.cuda.synchronize(); t_ = time.perf_counter()
        for _ in range(args.steps):
            mpc.reset(); mpc.step_dev(u_out.data_ptr())
        torch.cuda.synchronize(); t_ = time.perf_counter() - t_
        per_phase_line = {"value": B * args.steps / t_, "unit": "solves/s", "ms_per_step": 1e3 * t_ / args.steps, "phase_ms": phases_of(mpc, u_ptr=u_out.data_ptr()),
                          "config": "pg_set_pipeline(0): nodes (k_nodes), update_QP! (k_linearize_split) and solve as one launch each; the default runs the first two as one pipelined launch"}
        mpc.set_pipeline(1)
        mpc.reset(); mpc.step_dev(u_out.data_ptr()); torch.cuda.synchronize()         # (leave the headline controller's outputs as the last thing in u_out)
    # a user-level way to more throughput: the same batch as two halves on two handles / two streams, submitted alternately -- the latency-bound nodes kernel of one
    # half (64 wavefronts on a 1024-SIMD part) runs under the throughput-bound kernels of the other.  Reported beside the headline, never as `value`.
    two_streams = None
    if rank == 0 and world == 1 and not args.no_warm and B % 2 == 0 and B >= 2048:
        hs = []
        for k in range(2):
            sl = slice(k * B // 2, (k + 1) * B // 2)
            m2 = pkg.BatchedTrajectoryTrackingMPC(traj, B // 2, device=local, precision=args.precision, phase_timing=False)
            st2 = torch.cuda.Stream(device=dev)
            m2.set_stream(st2.cuda_stream)
            m2.set_inputs(state[sl], control[sl], t0[sl], time_offset=toff[sl])
            hs.append((m2, st2))
        for _ in range(2):
            for m2, _s in hs: m2.reset(); m2.step_dev()
        torch.cuda.synchronize(); t_ = time.perf_counter()
        for _ in range(args.steps):
            for m2, _s in hs: m2.reset(); m2.step_dev()
        torch.cuda.synchronize(); t_ = time.perf_counter() - t_
        ok2 = sum(int(pkg.is_solved(m2.solve_info()[0]).sum()) for m2, _s in hs)
        two_streams = {"value": B * args.steps / t_, "unit": "solves/s", "ms_per_step": 1e3 * t_ / args.steps, "solved": f"{ok2}/{B}",
                       "config": f"the same {B} cold instances as 2 x {B // 2} on two handles and two HIP streams, steps submitted alternately, one synchronisation at the end"}
        for m2, _s in hs: m2.close()
    # ... and two WHOLE batches in flight on one GPU (two handles, two streams, steps submitted alternately): the nodes + update_QP launch of one batch runs on the SIMDs
    # the solve launch of the other leaves idle (its tail) and vice versa.  Twice the instances per GPU -- not the benchmark's configuration: reported beside it.
    two_batches = None
    if rank == 0 and world == 1 and not args.no_warm and B >= 2048:
        hs = []
        for k in range(2):
            m2 = pkg.BatchedTrajectoryTrackingMPC(traj, B, device=local, precision=args.precision, phase_timing=False)
            st2 = torch.cuda.Stream(device=dev)
            m2.set_stream(st2.cuda_stream)
            s2, c2, t2, o2 = pkg.synthetic.config2_inputs(traj, B, seed=777 + k, traj_mode=True)
            m2.set_inputs(s2, c2, t2, time_offset=o2)
            hs.append((m2, st2))
        for _ in range(2):
            for m2, _s in hs: m2.reset(); m2.step_dev()
        torch.cuda.synchronize(); t_ = time.perf_counter()
        for _ in range(args.steps):
            for m2, _s in hs: m2.reset(); m2.step_dev()
        torch.cuda.synchronize(); t_ = time.perf_counter() - t_
        ok2 = sum(int(pkg.is_solved(m2.solve_info()[0]).sum()) for m2, _s in hs)
        two_batches = {"value": 2 * B * args.steps / t_, "unit": "solves/s", "ms_per_pair_of_steps": 1e3 * t_ / args.steps, "solved": f"{ok2}/{2 * B}",
                       "config": f"two independent batches of {B} cold instances on two handles and two HIP streams, steps submitted alternately, one synchronisation at the end "
                                 f"({2 * B} instances in flight per GPU: not the benchmark's configuration)"}
        for m2, _s in hs: m2.close()
    # SURVEY 8(d), config 2 "repeat for vail" and "a second run with time_offset = NaN (path mode; coupled_lat_long.jl:115)": the same cold step on the reference's
    # other long test path and on the benchmark path with every instance in path-tracking mode -- two short secondary numbers
    variants = None
    if rank == 0 and world == 1 and not args.no_warm:
        variants = {}
        for key, path, tmode in (("vail", "vail", True), ("path_mode", "skidpadoval", False)):
            tj = pkg.load_path_fixture(path)
            mv = pkg.BatchedTrajectoryTrackingMPC(tj, B, device=local, precision=args.precision, phase_timing=False)
            mv.set_stream(torch.cuda.current_stream().cuda_stream)
            sv, cv, tv, ov = pkg.synthetic.config2_inputs(tj, B, seed=12345, traj_mode=tmode)
            mv.set_inputs(sv, cv, tv, time_offset=ov)
            for _ in range(3):
                mv.reset(); mv.step_dev()
            torch.cuda.synchronize(); t_ = time.perf_counter()
            for _ in range(args.steps):
                mv.reset(); mv.step_dev()
            torch.cuda.synchronize(); t_ = time.perf_counter() - t_
            stv, itv, _, _ = mv.solve_info(); pv = mv.polish_info()
            variants[key] = {"value": B * args.steps / t_, "unit": "solves/s", "ms_per_step": 1e3 * t_ / args.steps, "phase_ms": phases_of(mv),
                             "solved": f"{int(pkg.is_solved(stv).sum())}/{B}", "interior_point_instances": int((itv > 0).sum()), "polish_rounds_hist": hist(pv),
                             "workload": f"configs[1] on {path}, " + ("trajectory mode (time_offset = 0)" if tmode else "PATH mode (time_offset = NaN: V, A from the speed profile at s; coupled_lat_long.jl:115)")}
            mv.close()
    # the gathered controls hold every rank's shard: this rank's block equals its own output
    gather_ok = True if world == 1 else bool(torch.equal(gathered[rank * B:(rank + 1) * B], u_out) and torch.isfinite(gathered).all().item())

    # closed-loop rollouts resident on the device (pg_simulate_dev = simulate of model_predictive_control.jl:80-100 for the whole batch): per step the
    # four compute phases + the plant's RK4 step; warm branch of the nodes and warm start of the active set (vs the same loop with that warm start off)
    roll = None
    if rank == 0 and world == 1 and not args.no_rollout:
        roll = {"workload": f"{B} controllers in closed loop on the device, 40 steps of 10 ms after 4 warm-up steps (skidpadoval, config-2 initial states), {args.precision}"}
        for warm in (True, False):
            mr = pkg.BatchedTrajectoryTrackingMPC(traj, B, device=local, precision=args.precision, warm_polish=warm, phase_timing=False)
            mr.set_stream(torch.cuda.current_stream().cuda_stream)
            mr.set_inputs_dev(B, d_state.data_ptr(), d_control.data_ptr(), d_t0.data_ptr(), None, d_toff.data_ptr())
            mr.simulate_(4); torch.cuda.synchronize(); tr_ = time.perf_counter()
            mr.simulate_(40); torch.cuda.synchronize(); tr_ = time.perf_counter() - tr_
            str_, itr_, _, _ = mr.solve_info()
            roll["warm_start_of_active_set" if warm else "without_it"] = {"value": B * 40 / tr_, "unit": "solves/s", "ms_per_step": 1e3 * tr_ / 40, "solved_last_step": f"{int(pkg.is_solved(str_).sum())}/{B}",
                                                                          "ipm_iters_mean_last_step": float(np.mean(itr_)), "served_by_warm_polish_alone": int((itr_ == 0).sum())}
            mr.close()

    # BASELINE config 5: decoupled (lateral) MPC, N = 50 (N_short = 10, N_long = 40), same batch, cold every step (pg_reset before each: the nodes of that formulation have no warm branch, its solver does)
    dec = None
    if rank == 0 and world == 1 and not args.no_decoupled:
        def run_dec(walls, polish=None):
            mpc_d = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, device=local, precision=args.precision, walls=walls, polish=polish, phase_timing=False)
            mpc_d.set_stream(torch.cuda.current_stream().cuda_stream)
            mpc_d.set_inputs_dev(B, d_state.data_ptr(), d_control.data_ptr(), d_t0.data_ptr(), None, d_toff.data_ptr())
            for _ in range(2):
                mpc_d.reset(); mpc_d.step_dev(u_out.data_ptr())
            torch.cuda.synchronize(); td = time.perf_counter()
            for _ in range(args.steps):
                mpc_d.reset(); mpc_d.step_dev(u_out.data_ptr())          # (reset: solved = false -- every timed step is a COLD step; k_solve_lat warm-starts otherwise)
            torch.cuda.synchronize(); td = time.perf_counter() - td
            std, itd, _, _ = mpc_d.solve_info(); pd_ = mpc_d.polish_info()
            phd = phases_of(mpc_d, u_ptr=u_out.data_ptr())
            r = {"value": B * args.steps / td, "unit": "solves/s", "ms_per_step": 1e3 * td / args.steps, "phase_ms": phd,
                 "solved": f"{int(pkg.is_solved(std).sum())}/{B}", "ipm_iters_mean": float(np.mean(itd)), "ipm_iters_max": int(np.max(itd)), "verified_by_polish": f"{int((pd_ >= 1).sum())}/{B}"}
            # roofline of the dominant kernel (k_solve_lat): the launch runs as long as its slowest instance -- passes through the loop = interior-point iterations + polish rounds
            passes = itd.astype(np.float64) + np.where(pd_ > 0, pd_, 3) + 1.0
            r["roofline"] = roofline_of(TR, ("k_solve_lat<1, true, true, 16, 1>", "k_solve_lat<1, true, false, 64, 2>") if walls else ("k_solve_lat<1, false, true, 16, 1>", "k_solve_lat<1, false, false, 64, 2>"), phd[2], B, BYTES_PER_SOLVE_COLD_FP64 if args.precision == "f64" else 68,
                                        float(np.sum(lat_flops(50, 13 if walls else 10, passes))), args.precision,
                                        note="k_solve_lat is bound by the dependent fp64 chains of its slowest instance (one wavefront per SIMD, every wavefront resident): the HBM fraction is ~1e-5 by "
                                             "construction; `traffic` is what the per-wavefront workspace and the packed stage records move through the memory side (L2 / Infinity Cache, not HBM-served)")
            mpc_d.close()
            return r
        dec = {"workload": f"configs[4]: Batch={B} decoupled MPC, N=50 + both_walls (soft rows edge_R - sw <= e <= edge_L + sw from the tube's edge channels: a build-defined "
                           f"extension, the reference snapshot has no wall constraint), {args.precision}"}
        dec.update(run_dec(True))
        dec["without_walls"] = run_dec(False)            # the reference's own lateral QP (decoupled_lat_long.jl as it stands)
        dec["without_walls_interior_point_only"] = run_dec(False, polish=False)      # (polish = 0, the round-2 default: 5 of these 4096 instances end 1e-6 .. 1e-5 from the optimum)
        dec["accuracy"] = "measured in this run: cpu_baseline.decoupled_n50.accuracy (sample of 128 against verified KKT points, with and without the wall rows)"
        dec["solver"] = "k_solve_lat: Mehrotra interior point on the 5-state stage form (sixteen lanes per instance) to mu <= 3e-6, active-set polish (verified KKT point), interior point resumed to 1e-12 where the polish does not verify"

        # ... and in CLOSED LOOP on the device (pg_simulate_dev; the reference runs the lateral QP with OSQP's WarmStart = true in the same loop as the coupled one,
        # decoupled_lat_long.jl:139, Pigeon.jl:34): k_solve_lat's warm start of the active set against the same loop without it, on the benchmark batch (random starts:
        # k_solve_lat ends with its slowest instance, and the instances a warm attempt does not serve set the time of the launch) and on a settled loop
        def run_loop(path, walls, burn, warm):
            tj = pkg.load_path_fixture(path)
            s_, c_, t_, o_ = pkg.synthetic.config2_inputs(tj, B, seed=12345)
            ml = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), tj, B, N_short=10, N_long=40, device=local, precision=args.precision, walls=walls, warm_polish=warm, phase_timing=False)
            ml.set_stream(torch.cuda.current_stream().cuda_stream)
            ml.set_inputs(s_, c_, t_, time_offset=o_)
            ml.simulate_(burn); torch.cuda.synchronize(); tl = time.perf_counter()
            ml.simulate_(40); torch.cuda.synchronize(); tl = time.perf_counter() - tl
            stl, itl, _, _ = ml.solve_info(); pl = ml.polish_info()
            r = {"value": B * 40 / tl, "unit": "solves/s", "ms_per_step": 1e3 * tl / 40, "solved_last_step": f"{int(pkg.is_solved(stl).sum())}/{B}",
                 "verified_last_step": f"{int((pl >= 1).sum())}/{B}", "served_by_warm_attempt_alone_last_step": int((itl == 0).sum()), "ipm_iters_mean_last_step": float(np.mean(itl))}
            ml.close()
            return r
        if not args.no_rollout:
          dec["closed_loop"] = {"workload": f"{B} lateral controllers (N = 50) in closed loop on the device, 40 steps of 10 ms behind a burn-in; per step: time grid + projection, k_nodes_dec, k_qp_dec, k_solve_lat, plant RK4",
                                "benchmark_batch_walls": {"burn_in_steps": 4, "warm_start_of_active_set": run_loop("skidpadoval", True, 4, True), "without_it": run_loop("skidpadoval", True, 4, False)},
                                "settled_loop_EastPaddock": {"burn_in_steps": 100, "warm_start_of_active_set": run_loop("EastPaddock", False, 100, True), "without_it": run_loop("EastPaddock", False, 100, False)},
                                "accuracy": "every instance of a warm step <= 1e-6 of the exact optimum of its own QP data: tests/test_gpu_decoupled_closed_loop.py (B = 4096, both loops)"}

    # HJI value/gradient lookup (the bandwidth-bound kernel of the path): 2^20 random in-grid relative states against the config-3 grid
    hji = None
    if rank == 0 and world == 1 and not args.no_hji:
        import ctypes as C

        def lookup_rate(knots, Vg, gg, label):
            torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]; t_set = time.perf_counter()
            mpc.set_hji_cache(knots, Vg, gg)
            mpc.synchronize(); torch.cuda.synchronize()
            setup_s = time.perf_counter() - t_set; table_bytes = free0 - torch.cuda.mem_get_info()[0]
            nq = 1 << 20
            xq = torch.from_numpy(pkg.synthetic.hji_queries(knots, nq).astype(npdt)).to(dev)
            out8 = torch.empty(nq, 8, dtype=tdt, device=dev)
            look = lambda: mpc._chk(mpc.lib.pg_hji_lookup8_dev(mpc.h, nq, C.c_void_p(xq.data_ptr()), C.c_void_p(out8.data_ptr())), "pg_hji_lookup8_dev")
            for _ in range(3):
                look()
            torch.cuda.synchronize()
            reps = 20
            # ONE event pair per launch (the handle launches on torch's current stream, set_stream above): the average LAUNCH DURATION, the quantity rocprofv3 --stats
            # reports for the kernel.  (Round 4 timed 20 back-to-back launches with one pair: the tail of a launch overlaps the ramp of the next and the figure came
            # out above every single launch of the committed trace.)  The back-to-back rate is kept beside it.
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for e0, e1 in evs:
                e0.record(); look(); e1.record()
            torch.cuda.synchronize()
            per = np.array([e0.elapsed_time(e1) for e0, e1 in evs])
            b2b = evs[0][0].elapsed_time(evs[-1][1]) / reps
            ms = float(per.mean())
            gbs = nq * 4096 / (ms * 1e-3) / 1e9
            r = {"lookups_per_s": nq / (ms * 1e-3), "avg_launch_ms": ms, "min_launch_ms": float(per.min()), "max_launch_ms": float(per.max()), "back_to_back_ms_per_launch": float(b2b),
                 "timing": f"HIP events around each of {reps} launches (mean); back_to_back = first start to last stop / {reps}",
                 "algorithmic_bytes_per_lookup": 4096, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": gbs / HBM_PEAK_GBS, "bound": "hbm", "grid": label, "lookups": nq, "finite": bool(torch.isfinite(out8).all().item()),
                 # pg_set_hji_grid: host re-pack + upload + cell-record build, once per grid (wall clock of the call), and the device memory the installed table holds
                 "setup_s": setup_s, "device_table_bytes": int(table_bytes)}
            mpc.clear_hji_cache()
            return r
        hji = lookup_rate(*pkg.synthetic.hji_grid_large(), "13x13x9x9x9x9x9 float32 (V, gradV), 10 M nodes; device layout (default since round 4): sixteen 256 B cell records per lookup, each read 256 contiguous bytes per instruction and lane group (1.9 GB table = 6 x the node table)")
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kernels"]["pg::k_hji_lookup<3>"]
            hji["traffic"] = tr["hbm_bytes_per_launch"]; hji["traffic_source"] = tr["source"]
            for k in ("avg_launch_ns_rocprof", "min_launch_ns_rocprof", "max_launch_ns_rocprof"):
                hji[k] = tr.get(k)
        except Exception:
            hji["traffic"] = None
        # the two larger device layouts (option "hji_cell_dims": 1 KiB records, 6 GB table; 4 KiB records, the default of rounds 1-3, 19 GB), same grid -- far beyond the 256 MiB
        # Infinity Cache in every layout
        fb = {}
        for cd, lbl in ((5, "1 KiB cell records (4 per lookup)"), (7, "4 KiB cell records (1 per lookup)")):
            mpc.set_option("hji_cell_dims", cd)
            try:
                r = lookup_rate(*pkg.synthetic.hji_grid_large(), lbl)
                fb[lbl] = {k: r[k] for k in ("lookups_per_s", "avg_launch_ms", "min_launch_ms", "max_launch_ms", "achieved", "frac", "setup_s", "device_table_bytes")}
            finally:
                mpc.set_option("hji_cell_dims", 3)
        hji["other_layouts"] = fb
        # SURVEY 8(d) secondary number: a FOUR-dimensional value grid (BASELINE.json says "4D": relative position, heading, other-car speed) run through the
        # same 7-D kernel with the three remaining dimensions collapsed to two knots each -- every lookup still gathers its 4096 B of corner data
        hji["grid_4d"] = lookup_rate(*pkg.synthetic.hji_grid_large(dims=(49, 49, 25, 2, 2, 25, 2)), "4-D grid 49x49x25x25 embedded as 49x49x25x2x2x25x2 (three collapsed dimensions), 12 M nodes")

    # BASELINE config 3: coupled MPC + HJI safety constraint on the precomputed 7-D grid, fp32 (libpigeon_hip_f32.so: same translation unit, real = float)
    f32 = None
    if rank == 0 and world == 1 and not args.no_f32 and args.precision == "f64":
        m32 = pkg.BatchedTrajectoryTrackingMPC(traj, B, device=local, precision="f32", phase_timing=False)
        m32.set_stream(torch.cuda.current_stream().cuda_stream)
        TR32 = {"kernels": (TR or {}).get("kernels_f32", {})}
        other = pkg.synthetic.other_cars(state, seed=777)
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        s32, c32, o32 = f(state), f(control), f(other)
        u32 = torch.zeros(B, 3, dtype=torch.float32, device=dev)

        def run32(with_hji):
            m32.set_inputs_dev(B, s32.data_ptr(), c32.data_ptr(), d_t0.data_ptr(), o32.data_ptr() if with_hji else None, d_toff.data_ptr())
            for _ in range(2):
                m32.reset(); m32.step_dev(u32.data_ptr())
            torch.cuda.synchronize(); t_ = time.perf_counter()
            for _ in range(args.steps):
                m32.reset(); m32.step_dev(u32.data_ptr())
            torch.cuda.synchronize(); t_ = time.perf_counter() - t_
            st_, it_, _, _ = m32.solve_info(); p_ = m32.polish_info(); ph32 = phases_of(m32, u_ptr=u32.data_ptr())
            rounds32 = np.where(p_ > 0, p_, np.where(p_ < 0, 6, 0))
            return {"value": B * args.steps / t_, "unit": "solves/s", "ms_per_step": 1e3 * t_ / args.steps, "phase_ms": ph32,
                    "solved": f"{int(pkg.is_solved(st_).sum())}/{B}", "ipm_iters_mean": float(np.mean(it_)), "ipm_iters_hist": hist(it_), "verified_by_polish": f"{int((p_ >= 1).sum())}/{B}",
                    "roofline": roofline_of(TR32, "k_solve<false, false, false, true" if with_hji else "k_solve<false, false, false, false", ph32[2], B, 68 + (16 + 4096 if with_hji else 0), float(np.sum(solve_flops(m32.N, it_.astype(np.float64), rounds32.astype(np.float64)))), "f32",
                                            note="solve phase of the fp32 library (k_solve, two waves per SIMD); bytes per solve: 68 (SURVEY 8d, fp32) + other car 16 B + 4096 B of corner records with the safety row")}

        plain = run32(False)
        f32 = {"workload": f"configs[2]: Batch={B} coupled MPC + HJI safety constraint (13x13x9x9x9x9x9 float32 grid), N=30, fp32, cold", "dtype": "f32",
               "accuracy": "controls within 1e-3 (normalised) of the exact optimum of the same QP data, median 4e-8 (tests/test_gpu_f32.py, all sampled instances)", "without_hji": plain}
        if not args.no_hji:
            knots, Vg, gg = pkg.synthetic.hji_grid_large()
            m32.set_hji_cache(knots, Vg, gg)
            f32.update(run32(True))
            _, _, Vh = m32.hji_constraint()
            f32["hji_rows_active"] = int(np.sum(Vh <= 0.05)); f32["hji_in_grid"] = int(np.sum(np.isfinite(Vh)))
        m32.close()
        # BASELINE config 5 in fp32 (SURVEY 8d lists it in both precisions): the same lateral batch through libpigeon_hip_f32.so, cold
        if dec is not None:
            def run_dec32(walls):
                md = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, device=local, precision="f32", walls=walls, phase_timing=False, allow_f32_long_lateral=True)
                md.set_stream(torch.cuda.current_stream().cuda_stream)
                md.set_inputs_dev(B, s32.data_ptr(), c32.data_ptr(), d_t0.data_ptr(), None, d_toff.data_ptr())
                for _ in range(2):
                    md.reset(); md.step_dev(u32.data_ptr())
                torch.cuda.synchronize(); t_ = time.perf_counter()
                for _ in range(args.steps):
                    md.reset(); md.step_dev(u32.data_ptr())
                torch.cuda.synchronize(); t_ = time.perf_counter() - t_
                st_, it_, _, _ = md.solve_info(); p_ = md.polish_info()
                r = {"value": B * args.steps / t_, "unit": "solves/s", "ms_per_step": 1e3 * t_ / args.steps, "phase_ms": phases_of(md, u_ptr=u32.data_ptr()),
                     "solved": f"{int(pkg.is_solved(st_).sum())}/{B}", "status_hist": hist(st_), "ipm_iters_mean": float(np.mean(it_)), "verified_by_polish": f"{int((p_ >= 1).sum())}/{B}"}
                md.close()
                return r
            dec["fp32"] = {"dtype": "f32", "with_walls": run_dec32(True), "without_walls": run_dec32(False),
                           "accuracy": "every instance against the oracle's exact optimum of the fp32 library's own QP data: tests/test_gpu_f32.py::test_f32_config5_full_size_every_instance_against_the_oracle "
                                       "(>= 99.5 % solve; applied steering max 6e-3 rad, 99th percentile 1.2e-3, median 9e-7: single precision on an open-loop unstable 8 s horizon)"}

    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        names = ["nodes(time_steps+project+nodes)", "update_qp(hji+linearize)", "solve(k_solve+extract)"]
        # pg_set_pipeline (default for 2048..8192 instances with cold ones): nodes and update_QP are ONE launch (k_nodes_linearize) and the event between
        # the two phases falls behind it -- the first phase then holds both
        pipelined = float(ph[1]) < 0.02 and float(ph[0]) > 0.1
        if pipelined:
            names = ["nodes+update_qp(time_steps+project, then ONE pipelined launch k_nodes_linearize)", "update_qp(nothing left: see the first phase)", "solve(k_solve+extract)"]
        lin_ms = float(ph[0]) if pipelined else float(ph[1])
        dom = int(np.argmax(ph))
        dom_ms = float(ph[dom])
        bytes_per_solve = BYTES_PER_SOLVE_COLD_FP64 if args.precision == "f64" else 68        # 112 B fp64; fp32 = 9 floats + t0 (double) in, 3 floats + status + iters out
        achieved = B * bytes_per_solve / (dom_ms * 1e-3) / 1e9
        traffic = None; traffic_src = None; traffic_stale = None; tr = None; dom_extra = {}
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            # the counters were taken with a particular build: traffic.json records the hash of the kernel sources it was taken at (tools/summarize_profiles.py), and the
            # line says so when the sources have changed since (the GPU box has no .git: a content hash, not a commit)
            traffic_stale = tr.get("kernel_source_sha16") != kernel_source_sha16()
            kern = {0: "k_nodes_linearize" if pipelined else "k_nodes", 1: "k_linearize", 2: "k_solve"}[dom]                  # the kernel that makes up the dominant phase (no HJI row in the headline run)
            hit = [v for k, v in tr.get("kernels", {}).items() if kern in k]
            traffic = hit[0]["hbm_bytes_per_launch"] if hit else tr.get("hbm_bytes_per_launch"); traffic_src = (hit[0] if hit else tr).get("source")
            dom_extra = {k: hit[0][k] for k in ("valu_issue_frac", "valu_insts_per_launch", "avg_launch_ns_rocprof", "min_launch_ns_rocprof", "max_launch_ns_rocprof", "stddev_launch_ns_rocprof") if hit and k in hit[0]}
        except Exception:
            pass
        # The linearisation's flops are COUNTED, not modelled (round 6; rounds 1-5 priced the launch with the forward-mode count of an algorithm it no longer runs: 130 + 260 per
        # direction and evaluation, 1.0e10 per launch, i.e. "0.42 of the vector peak" -- above anything the 80.6 M counted VALU instructions could deliver).  What runs: the
        # right-hand side with its local Jacobian once per RK4 stage (tracking_jac: 356 flops by hand with every reciprocal / square root / sincos counted as one), then
        # ~51 flops per tangent direction (+ 6 for a control direction) -- 832 per ramp stage, 699 per zero-order-hold stage by hand; the ISA of the stage loops (software
        # sincos, refined reciprocals and square roots included) holds 1204 / 912: tools/isa_mix.py -> profiles/isa_mix.json, per interval x (N_short, N_long) x B.
        lin_fl, isa_lin = linearize_flops_from_isa(B, mpc.N_short, mpc.N - mpc.N_short, args.precision)
        peak_lin = FP64_VALU_PEAK_TF if args.precision == "f64" else 2 * FP64_VALU_PEAK_TF
        valu_lin = {"bound": "valu-" + args.precision, "kernel": "k_nodes_linearize" if pipelined else "k_linearize", "algorithmic_flops_per_launch": lin_fl, "achieved": lin_fl / (lin_ms * 1e-3) / 1e12, "peak": peak_lin,
                    "unit": "TFLOP/s", "frac": lin_fl / (lin_ms * 1e-3) / 1e12 / peak_lin, "avg_launch_ms": lin_ms, "isa": isa_lin,
                    # an UPPER bound on executed arithmetic: every fp64 arithmetic instruction of the launch (ISA count) taken as a 64-lane multiply-add
                    "executed_fp64_frac": None if not isa_lin else isa_lin["arith_wave_insts"] * 128.0 / (lin_ms * 1e-3) / 1e12 / peak_lin,
                    "source": "flops counted from the ISA of the kernel's loop nests x trip counts (tools/isa_mix.py, profiles/isa_mix.json; the nodes recurrence -- 2 % of the instructions -- is not in it); time live (HIP events: " + ("the nodes + update_qp phase -- projection, the nodes recurrence and the linearisation running under it" if pipelined else "the update_qp phase") + ", no HJI row in the headline run)"}
        # secondary roofline of the dominant kernel: ALGORITHMIC flops (model above x the iteration counts of THIS run) against the fp64 / fp32 vector peak --
        # the resource class that binds (the step moves 112 B per solve through HBM, so its HBM fraction is ~1e-5 by construction)
        rounds = np.where(pol > 0, pol, np.where(pol < 0, 6, 0))
        fl = float(np.sum(solve_flops(mpc.N, it.astype(np.float64), rounds.astype(np.float64))))
        peak_tf = FP64_VALU_PEAK_TF if args.precision == "f64" else 2 * FP64_VALU_PEAK_TF
        valu = {"bound": "valu-" + args.precision, "kernel": "k_solve", "algorithmic_flops_per_launch": fl, "flops_per_solve_mean": fl / B, "achieved": fl / (float(ph[2]) * 1e-3) / 1e12,
                "peak": peak_tf, "unit": "TFLOP/s", "frac": fl / (float(ph[2]) * 1e-3) / 1e12 / peak_tf, "avg_launch_ms": float(ph[2]),
                "source": "flop model of the stage-structured interior point (bench.py solve_flops, EXPERIMENTS.md 6) x live iteration / polish-round counts; time live (HIP events)"}
        # the two kernels that make up the step (nodes + update_QP! launch, solve launch), each with what the committed counter passes say about it: the headline `roofline`
        # names the longer one and lists BOTH (their phases are within a few per cent of each other and swap places from run to run)
        def kernel_entry(name, match, ms, flop_obj):
            e = {"kernel": name, "avg_launch_ms": ms, "hbm_frac": B * bytes_per_solve / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "valu_flop_frac": flop_obj["frac"]}
            if flop_obj.get("executed_fp64_frac") is not None:
                e["executed_fp64_frac"] = flop_obj["executed_fp64_frac"]
            hit_ = [v for k, v in (tr or {}).get("kernels" if args.precision == "f64" else "kernels_f32", {}).items() if match in k]
            if hit_:
                h_ = hit_[0]
                e["traffic"] = h_.get("hbm_bytes_per_launch"); e["valu_issue_frac"] = h_.get("valu_issue_frac"); e["avg_launch_ms_rocprof"] = (h_.get("avg_launch_ns_rocprof") or 0) * 1e-6 or None
                hw_ = h_.get("hw_flops")
                if isinstance(hw_, dict) and hw_.get("valu_flops_per_launch"):
                    # the hardware's own count of executed arithmetic (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 of the committed PMC pass) over the LIVE duration
                    e["hw_valu_flop_frac"] = hw_["valu_flops_per_launch"] / (ms * 1e-3) / 1e12 / hw_["peak_tflops"]; e["hw_arith_share_of_valu_insts"] = hw_.get("arith_share_of_valu_insts")
                    # ... which IS the flop fraction of this kernel: the model / ISA count stays beside it as `model_flop_frac`
                    e["model_flop_frac"] = e["valu_flop_frac"]; e["valu_flop_frac"] = e["hw_valu_flop_frac"]; e["valu_flop_frac_source"] = "counters: 64 x (ADD + MUL + TRANS + 2 FMA) wave instructions of the committed PMC pass / live duration"
                m_ = h_.get("mfma")
                if isinstance(m_, dict):
                    e["mfma_busy_frac"] = m_.get("busy_frac_of_simd_time"); e["mfma_flop_frac"] = m_.get("frac_of_peak")
            return e
        lin_match = ("k_nodes_linearize" if pipelined else "k_linearize")
        kernels = [kernel_entry("k_solve", "k_solve<false, false, false, false", float(ph[2]), valu), kernel_entry(lin_match, lin_match, lin_ms, valu_lin)]
        kernels.sort(key=lambda e: -e["avg_launch_ms"])
        try:
            fallbacks = int(mpc.pipeline_fallbacks())
        except Exception:
            fallbacks = None
        line = {
            "metric": "MPC QP solves/sec (N=30 coupled, X1 model)", "kernel_source_sha16": kernel_source_sha16(), "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "timing": timing, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (f"configs[1]: Batch={B}/GPU coupled MPC, N=30 (N_short=10, N_long=20), X1, randomised x0 along skidpadoval, cold start, HJI inactive, fp64" if args.precision == "f64"
                                    else f"configs[3]: Batch={world * B} coupled MPC, N=30, fp32, sharded {B}/GPU x{world}, RCCL all_gather of controls, cold start"),
                       "batch_per_gpu": B, "parallelism": f"batch-sharded x{world}, all_gather of controls ({args.backend})" if world > 1 else "single GPU",
                       "solver": "active-set rounds from the empty set on the stage-structured QP (Riccati; pg_config.cold_guess = 8 + structural rules for the steering rows), verified KKT point; instances the rounds do not serve: "
                                 "Mehrotra interior point to mu <= " + ("3e-6" if args.precision == "f64" else "1e-4") + ", then active-set polish (verified KKT point)",
                       "accuracy": "measured in this run: cpu_baseline.accuracy (sample of 256); every one of the 4096 instances <= 1e-6 is asserted by tests/test_gpu_full_size.py (measured max 5e-11)",
                       "instrumentation": "the library's default (option phase_timing = 0: no per-phase HIP events on the stream) in every timed loop; phase_ms / roofline launch times come from "
                                          "separate steps with the events on (they cost 13-25 us of stream time per step)"},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale, "avg_launch_ms": dom_ms, **dom_extra, "mfma": mfma_util(tr, 2, pipelined), "valu": valu, "valu_linearize": valu_lin,
                         "kernels": kernels,
                         "note": f"algorithmic HBM bytes are {bytes_per_solve} B/solve (SURVEY 8d): the step is VALU/LDS/latency bound by construction, not HBM bound; see roofline.kernels (valu_flop_frac, valu_issue_frac, mfma_busy_frac) and roofline.hji_lookup (the bandwidth-bound kernel of the path)"},
            "phase_ms": {n: float(v) for n, v in zip(names, ph)},
            "phase_ms_short": {"nodes+update_qp" if pipelined else "nodes": float(ph[0]), "update_qp": float(ph[1]), "solve": float(ph[2])},
            "pipeline_fallbacks": fallbacks, "env": pg_environment(),
            "warm_value": None if args.no_warm else world * B * args.steps / warm_elapsed,
            "solved": f"{ok}/{B}", "gather_ok": gather_ok, "ranks": world if world == 1 else dist.get_world_size(), "collective": None if world == 1 else ("rccl" if args.backend == "nccl" else "gloo"),
            "per_rank_ms_per_step": rank_ms, "gather_ms": gather_ms, "rccl_version": rccl_version(torch) if world > 1 else None, "xgmi": XGMI if world > 1 else None, "ipm_iters_mean": float(np.mean(it)), "ipm_iters_max": int(np.max(it)), "ipm_iters_hist": hist(it),
            "polish_rounds_hist": hist(pol), "polish_note": "k >= 1: verified in round k; 0: not run; -1: not verified (interior-point iterate at 1e-12 kept)",
            "served_by_active_set_guess_alone": int((it == 0).sum()),
        }
        if ipm_only is not None:
            line["interior_point_only"] = ipm_only
        if two_streams is not None:
            line["two_half_batches_on_two_streams"] = two_streams
        if two_batches is not None:
            line["two_batches_on_two_streams"] = two_batches
        if fused_line is not None:
            line["fused_step"] = fused_line
        if per_phase_line is not None:
            line["launch_per_phase"] = per_phase_line
        if roll is not None:
            line["closed_loop_rollout"] = roll
        if variants is not None:
            line["config2_variants"] = variants
        if hji is not None:
            line["hji_lookup"] = hji
            line["roofline"]["hji_lookup"] = {k: hji.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "min_launch_ms", "max_launch_ms", "traffic")}
            line["roofline"]["hji_lookup"]["algorithmic_bytes_per_launch"] = hji["lookups"] * 4096
        if dec is not None:
            line["decoupled_n50"] = dec
        if f32 is not None:
            line["fp32"] = f32
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pkg, traj, mpc, state, control, t0, toff, args.precision, local)
        # headline numbers of the other BASELINE configs, for the compact line (everything else about them: bench_full.json)
        sec = {"warm_solves_per_s": line["warm_value"]}
        if f32 is not None:
            sec["config3_f32_hji_solves_per_s"] = f32.get("value"); sec["f32_cold_solves_per_s"] = f32["without_hji"]["value"]
        if dec is not None:
            sec["config5_n50_walls_solves_per_s"] = dec.get("value"); sec["config5_n50_solves_per_s"] = dec["without_walls"]["value"]; sec["config5_verified"] = dec.get("verified_by_polish")
        if roll is not None:
            sec["closed_loop_solves_per_s"] = roll["warm_start_of_active_set"]["value"]
        if variants is not None:
            sec["config2_vail_solves_per_s"] = variants["vail"]["value"]; sec["config2_path_mode_solves_per_s"] = variants["path_mode"]["value"]
        line["secondary"] = {k: v for k, v in sec.items() if v is not None}
        line["roofline_consistency"] = roofline_consistency(line)
        emit(line, args.full_record)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
