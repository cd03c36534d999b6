"""The diagnostic tools of rounds 1-4 steered the library through PG_* environment variables; since round 5 the library reads nothing from the environment (pg_set_option).
The tools keep their command lines: this helper turns the variables they document into the `options=` dictionary of the Python mirror (tools only: the product never reads
the environment).  Options named diag_* exist in the diagnostic library only: `precision_for(opts, "f64")` picks it."""
import os

_MAP = {
    "PG_CLIP_GUESS": "clip_guess", "PG_CK_RICCATI": "ck_riccati", "PG_WARM_TRIVIAL_COLD": "warm_trivial_cold", "PG_SOLVE_SPLIT": "solve_split", "PG_HJI_SEED": "hji_seed",
    "PG_HJI_ROUNDS": "hji_rounds", "PG_PIPE_MIN": "pipe_min", "PG_PIPE_MAX": "pipe_max", "PG_LIN_LPI": "lin_lanes", "PG_GRAPH": "graph", "PG_HJI_CELL_DIMS": "hji_cell_dims",
    "PG_LAT_MEM": "lat_workspace", "PG_LAT_MU0_COST": "lat_mu0_cost", "PG_LAT_FAR_COST": "lat_far_cost", "PG_LAT_POLISH2": "lat_polish2", "PG_LAT_WARM_ROUNDS": "lat_warm_rounds",
    "PG_LAT_SPLIT": "lat_split", "PG_LAT_RHO_SCALE": "lat_rho_scale", "PG_LAT_POLISH_ROUNDS": "lat_polish_rounds", "PG_LAT_SETTLE": "lat_settle", "PG_LAT_WIPM": "lat_wipm",
    "PG_LAT_PIN": "lat_pin", "PG_DEBUG_INSTANCE": "diag_instance", "PG_LIN_G": "diag_lin_groups", "PG_PIPE_FAULT": "diag_pipe_fault",
}


def options_from_env(extra=None):
    opts = {}
    for env, name in _MAP.items():
        if env in os.environ:
            opts[name] = float(os.environ[env])
    if "PG_SOLVE_LAT" in os.environ:                       # 1 = k_solve_lat, 0 = the embedding in k_solve
        opts["lateral_solver"] = 1.0 if os.environ["PG_SOLVE_LAT"] == "1" else 2.0
    opts.update(extra or {})
    return opts


def precision_for(opts, precision="f64"):
    return "f64-diag" if precision == "f64" and any(k.startswith("diag_") for k in opts) else precision
