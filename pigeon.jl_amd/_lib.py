"""ctypes binding of libpigeon_hip.so (C ABI: include/pigeon_mpc.h).  No CPU fallback: a missing library or GPU raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpigeon_hip.so")             # fp64 (the reference's arithmetic type)
LIB_PATH_F32 = os.path.join(_HERE, "csrc", "libpigeon_hip_f32.so")     # fp32 build of the same sources (BASELINE configs 3/4)
LIB_PATH_DIAG = os.path.join(_HERE, "csrc", "libpigeon_hip_diag.so")   # fp64, -DPG_DIAG: fault injection and traces for tests / tools (precision "f64-diag"); never the product
_libs = {}


class PigeonError(RuntimeError):
    pass


class pg_vehicle(C.Structure):
    _fields_ = [(n, C.c_double) for n in ["G", "m", "Izz", "L", "a", "b", "h", "mu", "Caf", "Car", "Cd0", "Cd1", "Cd2", "fwd_frac", "rwd_frac",
                                          "fwb_frac", "rwb_frac", "Fx_max", "Fx_min", "Px_max", "delta_max", "kappa_max"]]


class pg_control_params(C.Structure):
    _fields_ = [(n, C.c_double) for n in ["V_min", "V_max", "k_V", "k_s", "deltadot_max", "Q_ds", "Q_dpsi", "Q_e", "W_beta", "W_r", "W_HJI",
                                          "R_delta", "R_ddelta", "R_Fx", "R_dFx"]] + [("N_HJI", C.c_int32), ("_pad", C.c_int32)]


class pg_config(C.Structure):
    _fields_ = [("vehicle", pg_vehicle), ("control", pg_control_params), ("N_short", C.c_int32), ("N_long", C.c_int32), ("dt_short", C.c_double),
                ("dt_long", C.c_double), ("use_correction_step", C.c_int32), ("rk4_substeps", C.c_int32), ("hji_eps", C.c_double),
                ("batch_capacity", C.c_int32), ("device", C.c_int32), ("ipm_max_iter", C.c_int32), ("formulation", C.c_int32), ("ipm_tol", C.c_double),
                ("ipm_mu0", C.c_double), ("walls", C.c_int32), ("allow_f32_long_lateral", C.c_int32), ("wall_weight", C.c_double),
                ("polish", C.c_int32), ("_pad3", C.c_int32), ("polish_rho", C.c_double), ("polish_tol", C.c_double), ("polish_ipm_tol", C.c_double),
                ("warm_polish", C.c_int32), ("cold_guess", C.c_int32)]


# every symbol include/pigeon_mpc.h declares (tests check that the built library exports each one)
SYMBOLS = ["pg_precision_bits", "pg_abi_layout", "pg_default_config", "pg_default_config_decoupled", "pg_create", "pg_destroy", "pg_last_error", "pg_get_config", "pg_get_u_normalization", "pg_set_trajectory", "pg_set_trajectories", "pg_set_trajectory_index",
           "pg_set_hji_grid", "pg_clear_hji_grid", "pg_reset", "pg_set_inputs", "pg_set_inputs_dev", "pg_compute_time_steps",
           "pg_compute_linearization_nodes", "pg_update_qp", "pg_solve", "pg_get_next_control", "pg_get_next_control_dev", "pg_get_next_control_hji", "pg_get_next_control_hji_dev", "pg_step", "pg_step_dev", "pg_simulate_dev", "pg_simulate_clock", "pg_get_state",
           "pg_set_stream", "pg_set_fusion", "pg_set_pipeline", "pg_set_option", "pg_get_option", "pg_get_pipeline_fallbacks", "pg_synchronize", "pg_get_time_steps", "pg_get_nodes", "pg_get_path_coordinates", "pg_qp_len", "pg_get_qp", "pg_set_qp", "pg_get_solution",
           "pg_get_solve_info", "pg_get_polish_info", "pg_get_multipliers", "pg_get_phase_ms", "pg_hji_lookup", "pg_hji_lookup_dev", "pg_hji_lookup8_dev", "pg_hji_grid_dims", "pg_hji_slice", "pg_get_hji_constraint", "pg_get_walls"]


def load_library(precision="f64"):
    """Loads the HIP library of the requested arithmetic type.  PyTorch-ROCm is imported first so that both share ONE HIP runtime in this process."""
    assert precision in ("f64", "f32", "f64-diag")
    if precision in _libs:
        return _libs[precision]
    path = {"f64": LIB_PATH, "f32": LIB_PATH_F32, "f64-diag": LIB_PATH_DIAG}[precision]
    # A/B runs of an experimental build select it here (PIGEON_HIP_LIB / PIGEON_HIP_LIB_F32 = path of the .so) instead of copying it over the shipped library
    # (read by this Python mirror only: the C library itself reads nothing from the environment)
    path = os.environ.get({"f64": "PIGEON_HIP_LIB", "f32": "PIGEON_HIP_LIB_F32", "f64-diag": "PIGEON_HIP_LIB_DIAG"}[precision], path)
    if not os.path.exists(path):
        raise PigeonError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)")
    try:
        import torch  # noqa: F401  (plumbing: device memory / streams / torch.distributed live in the same HIP runtime)
    except Exception:
        pass
    lib = C.CDLL(path)
    lib.pg_last_error.restype = C.c_char_p
    lib.pg_last_error.argtypes = [C.c_void_p]
    for s in SYMBOLS:
        getattr(lib, s)
    assert lib.pg_precision_bits() == (32 if precision == "f32" else 64)
    check_layout(lib)
    _libs[precision] = lib
    return lib


LAYOUT_FIELDS = ["control", "N_short", "dt_short", "use_correction_step", "hji_eps", "batch_capacity", "ipm_max_iter", "formulation", "ipm_tol", "ipm_mu0", "walls", "wall_weight",
                 "polish", "polish_rho", "polish_tol", "polish_ipm_tol", "warm_polish", "cold_guess"]


def mirror_layout():
    """The numbers pg_abi_layout reports, computed from the ctypes mirrors above."""
    return [C.sizeof(pg_config), C.sizeof(pg_vehicle), C.sizeof(pg_control_params)] + [getattr(pg_config, f).offset for f in LAYOUT_FIELDS] + \
           [pg_control_params.N_HJI.offset, pg_vehicle.kappa_max.offset]


def check_layout(lib):
    """The ctypes structs are a hand copy of include/pigeon_mpc.h: refuse to run against a library whose struct layout differs."""
    n = lib.pg_abi_layout(None, 0)
    out = (C.c_int32 * n)()
    lib.pg_abi_layout(out, n)
    if list(out) != mirror_layout():
        raise PigeonError(f"struct layout of the ctypes mirror {mirror_layout()} differs from the library's {list(out)}: update pigeon.jl_amd/_lib.py to include/pigeon_mpc.h")


def check(lib, h, rc, what):
    if rc != 0:
        msg = lib.pg_last_error(h)
        raise PigeonError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")
