"""FMPC golden vectors (tests/golden/fmpc_golden.npz, made by tests/golden/make_fmpc_golden.py).

  * not gpu: the CPU oracle built on THIS host reproduces the committed vectors bit for bit (the checker that runs on the GPU box
    is the checker that was pinned in the build container);
  * gpu: the HIP path, through the C-ABI, reproduces them: Status and iteration count exact, values within rtol 1e-8 / atol 1e-10.
"""
import os

import numpy as np
import pytest

from oracle import fmpc as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fmpc_golden.npz")
D = np.load(GOLDEN)
NAMES = [str(n) for n in D["__names__"]]
CFG_KEYS = [str(k) for k in D["__cfg_keys__"]]
INT_KEYS = set(CFG_KEYS) - {"kkt_error_thre"}


def case(name):
    g = {k.split("/", 1)[1]: D[k] for k in D.files if k.startswith(name + "/")}
    kw = {k: (int(v) if k in INT_KEYS else float(v)) for k, v in zip(CFG_KEYS, g["cfg"])}
    return g, kw


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_fmpc_golden(name):
    g, kw = case(name)
    var = O.Variable(g["in_x"], g["in_u"], g["in_lam"], g["in_s"], g["in_nu"])
    r = O.solve(str(g["model"]), O.default_config(**kw), g["params"], float(g["t0"]), g["x0"], var, float(g["barrier_eps_in"]))
    assert r.status == int(g["status"]) and r.iters == int(g["iters"])
    for k, a in zip(("x", "u", "lam", "s", "nu"), r.variable.arrays()):
        assert np.array_equal(a, g["out_" + k]), k
    assert np.array_equal(r.trace, g["trace"]) and r.barrier_eps == float(g["barrier_eps_out"])
    assert np.array_equal(r.K, g["K"]) and np.array_equal(r.P, g["P"])


@pytest.mark.gpu
def test_hip_reproduces_fmpc_golden():
    """Every case of one problem type and shape goes through one handle (cases of a type with equal horizons share a batch)."""
    from nmpc_amd import fmpc as F

    groups = {}
    for name in NAMES:
        g, kw = case(name)
        key = (str(g["model"]), tuple(sorted(kw.items())), tuple(g["params"]))
        groups.setdefault(key, []).append((name, g))
    classes = {"fmpc_oscillator": F.FmpcProblemOscillator, "fmpc_cartpole": F.FmpcProblemCartPole, "fmpc_pointmass": F.FmpcProblemPointMass}
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
        it, out, tr, cl, be = s.iters(), s.variable(), s.traceDataList(), s.coeffList(), s.barrierEps()
        for b, (name, g) in enumerate(cases):
            assert st[b] == int(g["status"]) and it[b] == int(g["iters"]), name
            for k, a in zip(("x", "u", "lam", "s", "nu"), out.arrays()):
                assert np.allclose(a[b], g["out_" + k], rtol=1e-8, atol=1e-10), (name, k)
            assert np.allclose(tr[b], g["trace"], rtol=1e-8, atol=1e-10), name
            assert np.isclose(be[b], float(g["barrier_eps_out"]), rtol=1e-8)
            if int(g["status"]) == 5:  # the gains of the last iteration's backward pass
                assert np.allclose(cl["K"][b], g["K"], rtol=1e-7, atol=1e-9), name
                assert np.allclose(cl["P"][b], g["P"], rtol=1e-7, atol=1e-9), name
