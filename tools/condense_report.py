#!/usr/bin/env python3
"""Numbers behind EXPERIMENTS.md 11.x (partial condensing, numpy twin oracle/condense_numpy.py; CPU only): agreement with the stage-wise recursion and the condition numbers
of the pivots, per block size, on QP data of configs 2 (coupled N = 30) and 5 (lateral N = 50)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
from oracle import oracle as oracle_mod, condense_numpy as cn, lat_ipm_numpy as lp
import test_condense_numpy as T
skid = pkg.load_path_fixture("skidpadoval")
idx = list(range(0, 4096, 64))
rel = T._rel
print("== coupled, N = 30 (8 states, 2 inputs per stage)")
o, sds = T._coupled_sds(pkg, oracle_mod, skid, idx)
rng = np.random.default_rng(0)
for rho in (0.0, 1e7):
    for m in (1, 2, 3, 5):
        errs, cmax, anorm = [], [], []
        for sd in sds:
            S = o.unpack_sd(sd); hm = None
            if rho > 0:
                hm = np.zeros((30, 16), bool); n = int(rng.integers(3, 15)); hm[:n, 12] = True; hm[n + 1:n + 5, 3] = True
            st, QN, qN, x0 = cn.coupled_stages(S, o.control_params(), rho=rho, held=hm)
            x, v, c1 = cn.riccati(st, QN, qN, x0)
            xc, vc, cm = cn.riccati_condensed(st, QN, qN, x0, m)
            cs, _ = cn.condense(st, m)
            errs.append(max(rel(xc, x), rel(np.array(vc), np.array(v)))); cmax.append(cm.max()); anorm.append(max(np.linalg.norm(s["A"], 2) for s in cs))
        print(f"rho {rho:7.0e} block {m}: stages {-(-30 // m):2d}, pivot {2 * m} x {2 * m}; max rel. deviation from the stage-wise solution {max(errs):.1e}; cond(pivot) median {np.median(cmax):.1e} max {max(cmax):.1e}; |A~|_2 max {max(anorm):.2f}")
print("== lateral, N = 50 (5 states, 1 input per stage; open-loop unstable horizon)")
o, sds = T._lateral_sds(pkg, oracle_mod, skid, idx)
rng = np.random.default_rng(1)
for rho in (0.0, 1e7, 1e10):
    for m in (1, 2, 3, 5, 10, 25, 50):
        errs, cmax, anorm = [], [], []
        for sd in sds:
            D = lp.stage_data(o.unpack_sd(sd), o.cp); hm = None
            if rho > 0:
                hm = np.zeros((50, 10), bool); n = int(rng.integers(3, 12)); hm[:n, 8] = True; hm[n + 1:n + 6, 0] = True; hm[30:34, 9] = True
            st, QN, qN, x0 = cn.lateral_stages(D, rho=rho, held=hm)
            x, v, c1 = cn.riccati(st, QN, qN, x0)
            try:
                xc, vc, cm = cn.riccati_condensed(st, QN, qN, x0, m)
            except np.linalg.LinAlgError:
                errs.append(np.inf); continue
            cs, _ = cn.condense(st, m)
            errs.append(max(rel(xc, x), rel(np.array(vc), np.array(v)))); cmax.append(cm.max()); anorm.append(max(np.linalg.norm(s["A"], 2) for s in cs))
        print(f"rho {rho:7.0e} block {m:2d}: stages {-(-50 // m):2d}, pivot {m} x {m}; max rel. deviation {max(errs):.1e}; cond(pivot) median {np.median(cmax):.1e} max {max(cmax):.1e}; |A~|_2 max {max(anorm):.2e}")
