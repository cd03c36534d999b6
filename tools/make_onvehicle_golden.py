"""Writes tests/golden/onvehicle_cases.npz: the oracle's outputs (time grid, path coordinates, nodes, refreshed QP data, exact controls, active-set lists) for the cases
of tests/onvehicle_cases.py -- the reference's own dry-run configuration (Pigeon.jl:34-58) and the construction knobs no other test turns.  The oracle is the CPU
restatement (parity unpinned: EXPERIMENTS.md 5); the file pins ITS outputs against drift and gives the GPU tests fixed vectors.   python tools/make_onvehicle_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
import onvehicle_cases as oc
from oracle import oracle as om


def oracle_for(pkg, name):
    form, tname, kw, cp, hji = oc.CASES[name]
    traj = oc.trajectory(pkg, tname)
    if form == "coupled":
        o = om.Oracle(**kw); o.set_control_params(**cp)
        if hji:
            knots, V, g = pkg.synthetic.hji_grid(dims=oc.HJI_DIMS); o.set_hji_grid(knots, V, g)
    else:
        o = om.OracleDecoupled(**kw)
    o.set_trajectory(traj.data)
    return o, traj


def run_case(pkg, name, seed):
    form, tname, kw, cp, hji = oc.CASES[name]
    o, traj = oracle_for(pkg, name)
    state, control, t0, toff = oc.inputs(pkg, traj, tname, seed)
    other = pkg.synthetic.other_cars(state, seed=seed + 100) if hji else np.zeros((oc.B, 4))
    out = dict(state=state, control=control, t0=t0, toff=toff, other=other)
    TS, SEP, QS, US, PS, SD, U, ACT = [], [], [], [], [], [], [], []
    for b in range(oc.B):
        ts, dt = o.time_steps(t0[b])
        if form == "coupled":
            qs, us, ps = o.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
            sd = o.update_qp(qs, us, ps, dt, state[b], control[b], other[b])
            xe, ye, info = o.solve_exact(sd); assert info["status"] == 1 and info["polished"] >= 1, (name, b, info)
            X = o.split_x(xe); un = o.u_norm
            u = o.next_control(X["u"][1])
            act = om.active_set(o.assemble_qp(sd), xe, ye, tol=1e-6)
            SEP.append(o.path_coordinates(state[b, 0], state[b, 1])[:3])
        else:
            qs, us, ps = o.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
            sd = o.update_qp(qs, us, ps, dt)
            xe, ye, info = o.solve_exact_verified(sd); assert info["status"] == 1 and info["polished"] >= 1, (name, b, info)
            X = o.split_x(xe)
            u = o.next_control(X["delta"][1], us[1, 1])
            act = om.active_set(o.assemble_qp(sd), xe, ye, tol=1e-6)
            SEP.append((0.0, 0.0, 0.0))
        TS.append(ts); QS.append(qs); US.append(us); PS.append(ps); SD.append(sd); U.append(u)
        assert len(act) <= 160; a = np.zeros(160, dtype=np.int32); a[:len(act)] = act; ACT.append(np.concatenate([[len(act)], a]))
    out.update(ts=np.array(TS), sep=np.array(SEP), qs=np.array(QS), us=np.array(US), ps=np.array(PS), sd=np.array(SD), u=np.array(U), act=np.array(ACT))
    return out


if __name__ == "__main__":
    pkg = load_pkg(); om.build()
    G = {}
    for i, name in enumerate(oc.CASES):
        r = run_case(pkg, name, seed=500 + i)
        for k, v in r.items():
            G[f"{name}__{k}"] = v
        print(name, "controls", np.round(r["u"][0], 6), "active rows", r["act"][:, 0].tolist())
    path = os.path.join(ROOT, "tests", "golden", "onvehicle_cases.npz")
    np.savez_compressed(path, **G)
    print("wrote", path, os.path.getsize(path), "bytes")
