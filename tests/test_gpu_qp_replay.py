"""pg_set_qp: QP data installed by hand are solved exactly like the ones update_QP! writes; and a recorded QP of the wide-random stress regime on which the
absolute solve of the active-set polish is limited by the conditioning of its penalty (EXPERIMENTS.md 4.1: why the corrector is a refinement step)."""
import os
import numpy as np
import pytest

from conftest import make_oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_installed_qp_data_are_solved_like_computed_ones(pkg, skidpad):
    B = 64
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=5)
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B)
    u, st, it = mpc.step_(state, control, t0, time_offset=toff)
    qp = mpc.qp_data(); x, _ = mpc.solution()
    mpc.reset()
    mpc.set_qp_data(qp[::-1].copy())                      # the same problems in reverse order
    mpc.solve_()
    x2, _ = mpc.solution(); st2, it2, _, _ = mpc.solve_info()
    assert np.all(st2 == pkg.SOLVED)
    assert np.array_equal(x2[::-1], x)                    # cold solve of identical data: identical bits
    assert np.array_equal(mpc.qp_data(), qp[::-1])
    with pytest.raises(pkg.PigeonError):
        mpc.set_qp_data(qp, b0=1)                         # range outside the batch
    mpc.close()


def test_recorded_ill_conditioned_qp(pkg, oracle_mod, skidpad):
    """A QP recorded from the fuzz regime (EastPaddock, warm step): at one long stage the rear tyre is saturated by braking, the stability envelope collapses to the line
    Uy = b r (all four envelope rows parallel, G = 0) and both slack rows are active with them.  The interior point does not converge on it at all (160 iterations with
    the polish off).  The polish finds the oracle's active set -- but the Riccati recursion with the penalty rho = 1e7 on those rows loses the weakly curved directions:
    solved as an absolute problem the controls came out 4e-3 off (1.6e-4 at rho = 1e6, 8e-11 at rho = 1e3: the error goes with rho^2).  The corrector of a polish round is
    therefore solved as a CORRECTION to the predictor's point (iterative refinement: residual evaluated at the actual point) and repeated on the same matrices until
    the correction is small: exact at every penalty."""
    qp = np.load(os.path.join(HERE, "golden", "qp_cases", "collapsed_envelope_eastpaddock.npz"))["qp"]
    orc = make_oracle(oracle_mod, skidpad)
    xe, ye, info = orc.solve_exact(qp)
    assert info["status"] == 1
    S = orc.split_x(xe)
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, 2, seed=1)
    for rho in (None, 1e6, 1e3):
        mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, 2, **({} if rho is None else dict(polish_rho=rho)))
        mpc.step_(state, control, t0, time_offset=toff); mpc.reset()       # (any valid step: sizes the batch)
        mpc.set_qp_data(np.stack([qp, qp])); mpc.solve_()
        x, _ = mpc.solution(); st, it, _, _ = mpc.solve_info(); pol = mpc.polish_info()
        assert np.all(st == pkg.SOLVED) and np.all(pol >= 1) and np.array_equal(x[0], x[1])
        assert np.max(np.abs(x[0, :, 6:] - S["u"])) < 1e-7, rho          # measured 2.5e-9 / 1e-11 / 8e-11
        assert np.max(np.abs(x[0, :, :6] - S["q"])) < 1e-6, rho
        mpc.close()
