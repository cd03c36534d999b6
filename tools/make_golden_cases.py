#!/usr/bin/env python3
"""Generate tests/golden/coupled_cases.npz: inputs + oracle outputs for a few seeded cases (cold and warm second step).

The reference holds NO golden vectors for this path and cannot run here (no Julia), so these vectors come from the
build's own CPU oracle: they pin the oracle against silent drift and give the GPU path a fixed target.  PARITY UNPINNED
with respect to the Julia original."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402
from oracle import oracle as orc_mod  # noqa: E402


def main():
    pkg = load_pkg()
    out = {}
    for path in ["skidpadoval", "vail"]:
        traj = pkg.load_path_fixture(path)
        orc = orc_mod.Oracle(); orc.set_trajectory(traj.data)
        B = 6
        state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=2024, traj_mode=(path == "skidpadoval"))
        u1, sol1, it1, st1, _ = orc.step_batch(state, control, t0, time_offsets=toff, solver=0, want_sol=True)
        state2 = np.stack([orc.plant_step(state[b], control[b], 0.01) for b in range(B)])
        u2, sol2, it2, st2, _ = orc.step_batch(state2, u1, t0 + 0.01, time_offsets=toff, solver=0, want_sol=True)
        sds = []
        for b in range(B):
            ts, dt = orc.time_steps(t0[b])
            qs, us, ps = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
            sds.append(orc.update_qp(qs, us, ps, dt, state[b], control[b]))
        out.update({f"{path}_state": state, f"{path}_control": control, f"{path}_t0": t0, f"{path}_toff": toff, f"{path}_u1": u1, f"{path}_sol1": sol1,
                    f"{path}_state2": state2, f"{path}_u2": u2, f"{path}_sol2": sol2, f"{path}_qp1": np.array(sds)})
        assert (st1 == 1).all() and (st2 == 1).all()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "coupled_cases.npz"), **out)
    print("wrote coupled_cases.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
