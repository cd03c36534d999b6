"""Definitions of the on-vehicle / knob cases shared by tools/make_onvehicle_golden.py, tests/test_onvehicle_golden.py (CPU) and tests/test_gpu_onvehicle.py (GPU).

The reference's ONLY executed configuration is the module's own dry run (/root/reference/src/Pigeon.jl:34-58): CoupledTrajectoryTrackingMPC(X1(),
straight_trajectory(30., 5.), N_short=5, N_long=10) with the HJI cache installed, and DecoupledTrajectoryTrackingMPC(X1(), straight_trajectory(30., 5.)), both from
state (0, 0, 0, 5, 0, 0) with zero control at t = 0 (time_offset = NaN: path mode) -- a TWO-node tube.  The other cases turn the construction knobs that no other test
turns: use_correction_step = false (model_predictive_control.jl:22-24), R_delta, R_Fx > 0 (coupled_lat_long.jl:36-37), N_HJI = 10, rk4_substeps = 4, dt_long = 0.1."""
import numpy as np

B = 8
HJI_DIMS = (5, 5, 4, 4, 4, 4, 4)

CASES = {
    # name: (formulation, trajectory, kwargs of the constructor, control-parameter overrides, with HJI grid)
    "singleton_coupled": ("coupled", "straight", dict(N_short=5, N_long=10), {}, False),
    "singleton_coupled_hji": ("coupled", "straight", dict(N_short=5, N_long=10), {}, True),
    "singleton_decoupled": ("decoupled", "straight", dict(N_short=10, N_long=20), {}, False),
    "singleton_decoupled_short": ("decoupled", "straight", dict(N_short=5, N_long=10), {}, False),
    "no_correction_step": ("coupled", "skidpadoval", dict(N_short=10, N_long=20, use_correction_step=False), {}, False),
    "input_weights": ("coupled", "skidpadoval", dict(N_short=10, N_long=20), dict(R_delta=0.3, R_Fx=0.2), False),
    "n_hji_10": ("coupled", "skidpadoval", dict(N_short=10, N_long=20), dict(N_HJI=10), True),
    "rk4_substeps_4": ("coupled", "skidpadoval", dict(N_short=10, N_long=20, rk4_substeps=4), {}, False),
    "dt_long_0p1": ("coupled", "skidpadoval", dict(N_short=10, N_long=20, dt_long=0.1), {}, False),
}


def trajectory(pkg, name):
    return pkg.straight_trajectory(30.0, 5.0) if name == "straight" else pkg.load_path_fixture(name)


def inputs(pkg, traj, traj_name, seed):
    """(state [B,6], control [B,3], t0 [B], time_offset [B]).  Straight tube: instance 0 is EXACTLY the reference's dry run; the others perturb it (the tube is 30 m long and
    the horizon of the singleton covers ~10 m of it; two instances run past its end on purpose: linear extrapolation, trajectories.jl:32-35)."""
    rng = np.random.default_rng(seed)
    if traj_name != "straight":
        return pkg.synthetic.config2_inputs(traj, B, seed=seed)
    state = np.zeros((B, 6)); control = np.zeros((B, 3)); t0 = np.zeros(B); toff = np.full(B, np.nan)
    state[:, 3] = 5.0
    for b in range(1, B):
        n = rng.uniform(1.0, 24.0)
        state[b] = [rng.uniform(-0.4, 0.4), n, rng.uniform(-0.06, 0.06), rng.uniform(4.3, 5.7), rng.uniform(-0.15, 0.15), rng.uniform(-0.04, 0.04)]
        d0, fx = rng.uniform(-0.03, 0.03), rng.uniform(-400.0, 400.0)
        control[b] = [d0, (0.0 if fx > 0 else 0.6) * fx, (1.0 if fx > 0 else 0.4) * fx]
        t0[b] = n / 5.0 + rng.uniform(-0.1, 0.1)
        toff[b] = 0.0 if b % 2 == 0 else np.nan
    return state, control, t0, toff
