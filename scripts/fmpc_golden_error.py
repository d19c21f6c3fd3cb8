"""Largest relative deviation of the HIP FMPC path from each golden case (tests/golden/fmpc_golden.npz), per output: which cases are
ill-conditioned enough to show a kernel's rounding.   NMPC_HIP_FMPC_RICCATI=quad|fused python scripts/fmpc_golden_error.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_fmpc_golden as G
from nmpc_amd import fmpc as F

groups = {}
for name in G.NAMES:
    g, kw = G.case(name)
    groups.setdefault((str(g["model"]), tuple(sorted(kw.items())), tuple(g["params"])), []).append((name, g))
classes = {"fmpc_oscillator": F.FmpcProblemOscillator, "fmpc_cartpole": F.FmpcProblemCartPole, "fmpc_pointmass": F.FmpcProblemPointMass}
worst = []
for (model, kw_items, params), cases in groups.items():
    kw = dict(kw_items)
    prob = classes[model]()
    prob.p[:] = params
    B, T = len(cases), kw["horizon_steps"]
    s = F.FmpcSolverBatch(prob, B, T)
    for k, v in kw.items():
        if k != "horizon_steps":
            setattr(s.config(), k, bool(v) if k in F.Configuration._BOOL else v)
    var = F.Variable(*(np.stack([g["in_" + k] for _, g in cases]) for k in ("x", "u", "lam", "s", "nu")))
    s.setVariable(var, barrier_eps=np.array([float(g["barrier_eps_in"]) for _, g in cases]))
    st = s.solve(np.array([float(g["t0"]) for _, g in cases]), np.stack([g["x0"] for _, g in cases]))
    out = s.variable()
    for b, (name, g) in enumerate(cases):
        err = 0.0
        for k, a in zip(("x", "u", "lam", "s", "nu"), out.arrays()):
            want = np.asarray(g["out_" + k]); d = np.abs(a[b] - want) / (1e-10 / 1e-8 + np.abs(want))
            err = max(err, float(np.nanmax(d)))
        worst.append((err, name, int(st[b]) == int(g["status"])))
worst.sort(reverse=True)
print(os.environ.get("NMPC_HIP_FMPC_RICCATI", "auto"), s.kernelNames()[2], " worst scaled deviations (|d| / (1e-2 + |want|)):")
for e, n, ok in worst[:6]:
    print(f"   {n:32s} {e:.3e}  status equal {ok}")
