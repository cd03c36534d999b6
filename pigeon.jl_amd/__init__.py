"""MI355X-native batched MPC hot path of StanfordASL/Pigeon.jl (coupled lat/long tracking QP + HJI lookup).

Host mirror of the reference's call convention over the C ABI in include/pigeon_mpc.h; all compute is HIP on gfx950.
"""
from ._lib import PigeonError, load_library, LIB_PATH, LIB_PATH_F32, SYMBOLS  # noqa: F401
from .mpc import BatchedTrajectoryTrackingMPC, CoupledTrajectoryTrackingMPC, DecoupledTrajectoryTrackingMPC, decoupled_canonical_active_set, simulate, SOLVED, MAX_ITER, NUMERICAL, INFEASIBLE_X0, SOLVED_UNVERIFIED, is_solved  # noqa: F401
from .trajectories import TrajectoryTube, straight_trajectory, load_path_fixture, invcumtrapz  # noqa: F401
from .vehicles import X1, CoupledControlParams, DecoupledControlParams  # noqa: F401
from .hji_io import load_hji_grid, save_hji_grid, trace_zero_contour  # noqa: F401
from . import synthetic, sharding, trajectories, hji_io  # noqa: F401
