"""X1 parameter dictionary — mirrors /root/reference/src/vehicles.jl:1-59 (same keys, same formulas)."""
import math


def X1():
    X = {}
    X["G"] = 9.80665
    X["mfl"], X["mfr"], X["mrl"], X["mrr"] = 484.0, 455.0, 521.0, 504.0
    X["m"] = X["mfl"] + X["mfr"] + X["mrl"] + X["mrr"]
    X["Ixx"], X["Iyy"], X["Izz"] = 175.0, 1000.0, 2900.0
    X["L"] = 2.87
    X["d"] = 1.63
    X["a"] = (X["mrl"] + X["mrr"]) / X["m"] * X["L"]
    X["b"] = (X["mfl"] + X["mfr"]) / X["m"] * X["L"]
    X["hf"], X["hr"], X["h1"] = 0.1, 0.1, 0.37
    X["h"] = X["hf"] * X["b"] / X["L"] + X["hr"] * X["a"] / X["L"] + X["h1"]
    X["mu"] = 0.92
    X["Caf"], X["Car"] = 150e3, 220e3
    X["Fx_max"], X["Px_max"] = 5600.0, 75e3
    X["Cd0"], X["Cd1"], X["Cd2"] = 241.0, 25.1, 0.0
    X["fwd_frac"] = 0.0
    X["rwd_frac"] = 1 - X["fwd_frac"]
    X["fwb_frac"] = 0.6
    X["rwb_frac"] = 1 - X["fwb_frac"]
    X["Fx_min"] = max(-X["m"] * X["G"] * X["a"] * X["mu"] / (X["L"] * X["rwb_frac"] + X["mu"] * X["h"]),
                      -X["m"] * X["G"] * X["b"] * X["mu"] / (X["L"] * X["fwb_frac"] - X["mu"] * X["h"]))
    X["delta_max"] = 18 * math.pi / 180
    X["kappa_max"] = math.tan(X["delta_max"]) / X["L"]
    return X


def CoupledControlParams(**kw):
    """Keyword constructor of /root/reference/src/coupled_lat_long.jl:23-40."""
    p = dict(V_min=1.0, V_max=15.0, k_V=10 / 4 / 100, k_s=10 / 4 / 10000, deltadot_max=0.344, Q_ds=1.0, Q_dpsi=1.0, Q_e=1.0,
             W_beta=50 / (10 * math.pi / 180), W_r=50.0, W_HJI=500.0, N_HJI=3, R_delta=0.0, R_ddelta=0.1, R_Fx=0.0, R_dFx=0.5)
    for k, v in kw.items():
        if k not in p:
            raise KeyError(k)
        p[k] = v
    return p


def DecoupledControlParams(**kw):
    """Keyword constructor of /root/reference/src/decoupled_lat_long.jl:18-30."""
    d10 = 10 * math.pi / 180
    p = dict(V_min=1.0, V_max=15.0, k_V=10 / 4 / 100, k_s=10 / 4 / 10000, deltadot_max=0.344, Q_dpsi=1 / d10 ** 2, Q_e=1.0, W_beta=50 / d10, W_r=50.0,
             R_delta=0.0, R_ddelta=0.01 / d10 ** 2)
    for k, v in kw.items():
        if k not in p:
            raise KeyError(k)
        p[k] = v
    return p
