#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W : MPC QP solves/sec on N MI355X (one process per GPU, RCCL gather of controls).

A "step" = the reference's whole per-timestep callback body (compute_time_steps! -> compute_linearization_nodes! -> update_QP! ->
solve! -> get_next_control; /root/reference/src/ros_integration.jl:96-99,124) for a batch of B = 4096 independent instances per GPU
(BASELINE.json configs[1]: coupled MPC, N = 30, X1, randomised x0 along the skidpadoval test path, fp64), COLD (solved = false),
inputs resident in HBM before the timed region.  Weak scaling: every rank owns its own 4096 instances (seed 12345 + rank).
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
BYTES_PER_SOLVE_COLD_FP64 = 112           # state 6 + control 3 + t0 in, control 3 + status + iters out


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


def cpu_baseline(pkg, traj, state, control, t0, toff):
    """The reference ALGORITHM (OSQP-form ADMM with sparse LDL', default settings) on the host cores: oracle 'port'.
    Only this leg and the checker may touch oracle/."""
    from oracle import oracle as orc_mod
    cores = usable_cores()
    orc = orc_mod.Oracle(); orc.set_trajectory(traj.data)
    n1 = 128
    _, _, it1, st1, secs1 = orc.step_batch(state[:n1], control[:n1], t0[:n1], time_offsets=toff[:n1], solver=1, nthreads=1)
    orc2 = orc_mod.Oracle(); orc2.set_trajectory(traj.data)
    nall = min(len(t0), max(512, 128 * cores))
    _, _, it, st, secs = orc2.step_batch(state[:nall], control[:nall], t0[:nall], time_offsets=toff[:nall], solver=1, nthreads=cores)
    return {"value": nall / secs, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"{nall} cold instances of the same workload on {cores} host threads (OSQP-port ADMM, eps 1e-3, mean {float(np.mean(it)):.0f} iterations); "
                      f"1 thread: {n1 / secs1:.1f} solves/s on {n1} instances",
            "value_1thread": n1 / secs1, "note": "Julia reference not run (no Julia toolchain; third-party sources absent): C++ restatement of the same algorithm"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hji", action="store_true")
    ap.add_argument("--no-decoupled", action="store_true")
    ap.add_argument("--no-f32", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend for --gpus > 1: nccl = RCCL over xGMI (the product path); gloo = host-staged gather with the ranks sharing whatever GPUs exist "
                         "(test mode: exercises launch -> shard -> step -> gather end to end on a 1-GPU box)")
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
    traj = pkg.load_path_fixture("skidpadoval")
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, device=local, precision=args.precision)
    npdt = np.float64 if args.precision == "f64" else np.float32; tdt = torch.float64 if args.precision == "f64" else torch.float32
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345 + rank, traj_mode=True)
    dev = torch.device("cuda", local)
    d_state = torch.from_numpy(state.astype(npdt)).to(dev); d_control = torch.from_numpy(control.astype(npdt)).to(dev)
    d_t0 = torch.from_numpy(t0).to(dev); d_toff = torch.from_numpy(toff).to(dev)            # absolute time is double in both builds
    u_out = torch.zeros(B, 3, dtype=tdt, device=dev)
    gathered = torch.zeros(world * B, 3, dtype=tdt, device=dev) if world > 1 else None
    mpc.set_stream(torch.cuda.current_stream().cuda_stream)
    mpc.set_inputs_dev(B, d_state.data_ptr(), d_control.data_ptr(), d_t0.data_ptr(), None, d_toff.data_ptr())

    def one_step(cold=True):
        if cold:
            mpc.reset()                                   # solved = false for every instance (hipMemsetAsync on the same stream)
        mpc.step_dev(u_out.data_ptr())
        if world > 1:
            if args.backend == "nccl":
                dist.all_gather_into_tensor(gathered, u_out)  # RCCL over xGMI: the only collective on the path
            else:
                g_host = torch.empty(world * B, 3, dtype=tdt)
                dist.all_gather_into_tensor(g_host, u_out.cpu())
                gathered.copy_(g_host)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    phase = np.zeros(3)
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        one_step()
        # no host sync inside the loop: events are read after the region
    torch.cuda.synchronize()
    sync()
    elapsed = time.perf_counter() - t_begin
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # per-phase device time of the LAST step (HIP events recorded by pg_step_dev on the launch stream)
    phase = np.array(mpc.phase_ms())
    # average the dominant kernel over a few extra steps (outside the timed region) for the roofline line
    ph = []
    for _ in range(5):
        one_step(); torch.cuda.synchronize(); ph.append(mpc.phase_ms())
    ph = np.mean(np.array(ph), axis=0)
    # warm steps (second and later consecutive steps): reported as an extra, not as `value`
    sync(); tw = time.perf_counter()
    for _ in range(args.steps):
        one_step(cold=False)
    sync(); warm_elapsed = time.perf_counter() - tw

    st, it, act, mu = mpc.solve_info()
    ok = int((st == pkg.SOLVED).sum())
    # the gathered controls hold every rank's shard: this rank's block equals its own output
    gather_ok = True if world == 1 else bool(torch.equal(gathered[rank * B:(rank + 1) * B], u_out) and torch.isfinite(gathered).all().item())

    # BASELINE config 5: decoupled (lateral) MPC, N = 50 (N_short = 10, N_long = 40), same batch, cold every step (that formulation has no warm branch)
    dec = None
    if rank == 0 and not args.no_decoupled:
        def run_dec(walls):
            mpc_d = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, device=local, precision=args.precision, walls=walls)
            mpc_d.set_stream(torch.cuda.current_stream().cuda_stream)
            mpc_d.set_inputs_dev(B, d_state.data_ptr(), d_control.data_ptr(), d_t0.data_ptr(), None, d_toff.data_ptr())
            for _ in range(2):
                mpc_d.step_dev(u_out.data_ptr())
            torch.cuda.synchronize(); td = time.perf_counter()
            for _ in range(args.steps):
                mpc_d.step_dev(u_out.data_ptr())
            torch.cuda.synchronize(); td = time.perf_counter() - td
            std, itd, _, _ = mpc_d.solve_info()
            r = {"value": B * args.steps / td, "unit": "solves/s", "ms_per_step": 1e3 * td / args.steps, "phase_ms": [float(v) for v in mpc_d.phase_ms()],
                 "solved": f"{int((std == pkg.SOLVED).sum())}/{B}", "ipm_iters_mean": float(np.mean(itd))}
            mpc_d.close()
            return r
        dec = {"workload": f"configs[4]: Batch={B} decoupled MPC, N=50 + both_walls (soft rows edge_R - sw <= e <= edge_L + sw from the tube's edge channels: a build-defined "
                           f"extension, the reference snapshot has no wall constraint), {args.precision}"}
        dec.update(run_dec(True))
        dec["without_walls"] = run_dec(False)            # the reference's own lateral QP (decoupled_lat_long.jl as it stands)

    # HJI value/gradient lookup (the bandwidth-bound kernel of the path): 2^20 random in-grid relative states against the config-3 grid
    hji = None
    if rank == 0 and not args.no_hji:
        import ctypes as C
        knots, Vg, gg = pkg.synthetic.hji_grid_large()
        mpc.set_hji_cache(knots, Vg, gg)
        nq = 1 << 20
        xq = torch.from_numpy(pkg.synthetic.hji_queries(knots, nq).astype(npdt)).to(dev)
        out8 = torch.empty(nq, 8, dtype=tdt, device=dev)
        look = lambda: mpc._chk(mpc.lib.pg_hji_lookup8_dev(mpc.h, nq, C.c_void_p(xq.data_ptr()), C.c_void_p(out8.data_ptr())), "pg_hji_lookup8_dev")
        for _ in range(3):
            look()
        torch.cuda.synchronize()
        reps = 20
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)     # the handle launches on torch's current stream (set_stream above)
        ev0.record()
        for _ in range(reps):
            look()
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        gbs = nq * 4096 / (ms * 1e-3) / 1e9
        hji = {"lookups_per_s": nq / (ms * 1e-3), "avg_launch_ms": ms, "algorithmic_bytes_per_lookup": 4096, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": gbs / HBM_PEAK_GBS, "bound": "hbm", "grid": "13x13x9x9x9x9x9 float32 (V, gradV), 10 M nodes; device layout: one contiguous 4 KiB cell record per lookup (41 GB table, capacity traded for line efficiency)", "lookups": nq,
               "finite": bool(torch.isfinite(out8).all().item())}
        mpc.clear_hji_cache()

    # BASELINE config 3: coupled MPC + HJI safety constraint on the precomputed 7-D grid, fp32 (libpigeon_hip_f32.so: same sources, arithmetic type swapped)
    f32 = None
    if rank == 0 and not args.no_f32 and args.precision == "f64":
        m32 = pkg.BatchedTrajectoryTrackingMPC(traj, B, device=local, precision="f32")
        m32.set_stream(torch.cuda.current_stream().cuda_stream)
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
            st_, it_, _, _ = m32.solve_info()
            return {"value": B * args.steps / t_, "unit": "solves/s", "ms_per_step": 1e3 * t_ / args.steps, "phase_ms": [float(v) for v in m32.phase_ms()],
                    "solved": f"{int((st_ == pkg.SOLVED).sum())}/{B}", "ipm_iters_mean": float(np.mean(it_))}

        plain = run32(False)
        f32 = {"workload": f"configs[2]: Batch={B} coupled MPC + HJI safety constraint (13x13x9x9x9x9x9 float32 grid), N=30, fp32, cold", "dtype": "f32",
               "accuracy": "controls within 5e-3 (normalised) of the exact optimum, median 5e-5 (tests/test_gpu_f32.py)", "without_hji": plain}
        if not args.no_hji:
            knots, Vg, gg = pkg.synthetic.hji_grid_large()
            m32.set_hji_cache(knots, Vg, gg)
            f32.update(run32(True))
            _, _, Vh = m32.hji_constraint()
            f32["hji_rows_active"] = int(np.sum(Vh <= 0.05)); f32["hji_in_grid"] = int(np.sum(np.isfinite(Vh)))
        m32.close()

    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        names = ["nodes(time_steps+project+nodes)", "update_qp(hji+linearize+limits)", "solve(k_solve+extract)"]
        dom = int(np.argmax(ph))
        dom_ms = float(ph[dom])
        bytes_per_solve = BYTES_PER_SOLVE_COLD_FP64 if args.precision == "f64" else 68        # 112 B fp64; fp32 = 9 floats + t0 (double) in, 3 floats + status + iters out
        achieved = B * bytes_per_solve / (dom_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # secondary roofline of the dominant kernel: fp64 VALU issue slots (the resource that actually binds).  Instruction count per launch
        # from the committed PMC pass (SQ_INSTS_VALU, a property of the code and the inputs), time live; peak = 1024 SIMDs x 2.4 GHz / 4 clocks per wave-instruction
        valu = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_final", "pmc_summary.json")))
            ins = pm["pg::k_solve<false, false>"]["SQ_INSTS_VALU"]["mean_per_launch"] * (B / 4096.0)
            if args.precision == "f64" and dom == 2:
                valu = {"wave_instructions_per_launch": ins, "achieved": ins / (dom_ms * 1e-3) / 1e9, "peak": 1024 * 2.4e9 / 4 / 1e9, "unit": "G wave-instr/s",
                        "frac": ins / (dom_ms * 1e-3) / (1024 * 2.4e9 / 4), "source": "profiles/r01_final/pmc_summary.json (SQ_INSTS_VALU)"}
        except Exception:
            valu = None
        line = {
            "metric": "MPC QP solves/sec (N=30 coupled, X1 model)", "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (f"configs[1]: Batch={B}/GPU coupled MPC, N=30 (N_short=10, N_long=20), X1, randomised x0 along skidpadoval, cold start, HJI inactive, fp64" if args.precision == "f64"
                                    else f"configs[3]: Batch={world * B} coupled MPC, N=30, fp32, sharded {B}/GPU x{world}, RCCL all_gather of controls, cold start"),
                       "batch_per_gpu": B, "parallelism": f"batch-sharded x{world}, all_gather of controls" if world > 1 else "single GPU",
                       "solver": "Mehrotra interior point on the stage-structured QP (Riccati), tol " + ("1e-12" if args.precision == "f64" else "1e-5"),
                       "accuracy": "|u-u*| (normalised) vs exact optimum of the same QP data over the whole batch: median 2e-12, 99.9% <= 2.5e-7, max 3e-6 (3 of 4096 instances above 1e-6; tools/gpu_accuracy_full.py)" if args.precision == "f64" else "max|u-u*| <= 5e-3, median 5e-5 (normalised) vs exact optimum"},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "avg_launch_ms": dom_ms, "valu_issue": valu,
                         "note": f"algorithmic HBM bytes are {bytes_per_solve} B/solve (SURVEY 8d): the step is VALU/LDS/latency bound by construction, not HBM bound"},
            "phase_ms": {n: float(v) for n, v in zip(names, ph)},
            "warm_value": world * B * args.steps / warm_elapsed,
            "solved": f"{ok}/{B}", "gather_ok": gather_ok, "ipm_iters_mean": float(np.mean(it)), "ipm_iters_max": int(np.max(it)),
        }
        if hji is not None:
            line["hji_lookup"] = hji
        if dec is not None:
            line["decoupled_n50"] = dec
        if f32 is not None:
            line["fp32"] = f32
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pkg, traj, state, control, t0, toff)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
