"""A/B of fmpc_tail_kernel (step length + update + the head of the next iteration in one launch) against the separate kernels
(NMPC_HIP_FMPC_TAIL=0), same library, same inputs: every output of the solve must have the same bits; prints both solve times.
usage: fmpc_tail_ab.py [quick]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmpc_amd import fmpc as F  # noqa: E402

CASES = [  # model, B, T, max_iter, swing-up, reps
    ("fmpc_cartpole", 4096, 200, 5, True, 12),
    ("fmpc_cartpole", 4096, 200, 5, False, 6),
    ("fmpc_cartpole", 4096, 200, 2, False, 3),
    ("fmpc_cartpole", 4096, 200, 30, False, 3),
    ("fmpc_cartpole", 1000, 37, 8, True, 3),
    ("fmpc_cartpole", 17, 5, 3, True, 3),
    ("fmpc_cartpole", 100, 1, 2, True, 3),
    ("fmpc_cartpole", 256, 130, 1, True, 3),
    ("fmpc_oscillator", 2048, 100, 10, False, 3),
    ("fmpc_oscillator", 333, 64, 4, False, 3),
]
PROBLEMS = {"fmpc_cartpole": F.FmpcProblemCartPole, "fmpc_oscillator": F.FmpcProblemOscillator}


def run(tail, model, B, T, max_iter, swing, reps, poison=False):
    os.environ["NMPC_HIP_FMPC_TAIL"] = "1" if tail else "0"  # read when the handle is created
    prob = PROBLEMS[model](0.01)
    rng = np.random.default_rng(B * 1000 + T)
    n = prob.state_dim
    x0 = np.zeros((B, n))
    x0[:, 0] = rng.uniform(-1, 1, B)
    x0[:, 1] = rng.uniform(-0.3, 0.3, B) + (np.pi if swing and n == 4 else 0.0)
    if poison:  # a few instances start from NaN / a huge state: the error paths
        x0[::7, 0] = np.nan
        x0[3::11, 1] = 1e200
    s = F.FmpcSolverBatch(prob, B, T)
    s.config().max_iter = max_iter
    var = F.Variable.make(prob, T, B)
    ms = []
    for _ in range(reps):
        var.reset(0.0, 0.0, 0.0, 1.0, 1.0)
        try:
            s.solve(0.0, x0, var)
        except RuntimeError:
            pass
        ms.append(s.computationDuration().solve)
    out = s.variable()
    delta = s.deltaVariable()
    arrays = dict(zip(("x", "u", "lambda", "s", "nu"), out.arrays()))
    arrays.update(zip(("dx", "du", "dlambda", "ds", "dnu"), delta.arrays()))
    arrays.update(partials=s.partials())
    arrays.update(status=s.status(), iters=s.iters(), barrier_eps=s.barrierEps(), trace=s.traceDataList())
    arrays.update(("gain_" + k, v) for k, v in s.coeffList().items())
    h = hashlib.sha256()
    for k in sorted(arrays):
        h.update(np.ascontiguousarray(arrays[k]).tobytes())
    names = s.kernelNames()
    return h.hexdigest()[:16], float(np.median(ms[1:])) if len(ms) > 1 else ms[0], np.bincount(s.status(), minlength=7), names, arrays


def explain(a0, a1):
    for k in sorted(a0):
        x, y = np.ascontiguousarray(a0[k]), np.ascontiguousarray(a1[k])
        ne = x.view(np.uint8).reshape(x.shape + (-1,)) != y.view(np.uint8).reshape(y.shape + (-1,))
        ne = ne.any(axis=-1)
        if ne.any():
            idx = np.argwhere(ne)
            with np.errstate(invalid="ignore"):
                print(f"    {k}: {ne.sum()} of {ne.size} entries differ, first at {idx[0].tolist()}: {x[tuple(idx[0])]!r} vs {y[tuple(idx[0])]!r}; "
                      f"instances {np.unique(idx[:, 0])[:8].tolist()}")


bad = 0
for case in CASES[: (4 if len(sys.argv) > 1 else None)]:
    for poison in (False, True):
        d0, ms0, st0, n0, a0 = run(False, *case, poison=poison)
        d1, ms1, st1, n1, a1 = run(True, *case, poison=poison)
        same = d0 == d1
        bad += 0 if same else 1
        print(f"{case[0]:16s} B={case[1]:5d} T={case[2]:3d} max_iter={case[3]:2d} swing={int(case[4])} poison={int(poison)}: "
              f"separate {ms0:7.3f} ms  tail {ms1:7.3f} ms  status {st1.tolist()}  {'SAME bits' if same else 'DIFFERENT ' + d0 + ' ' + d1}"
              f"  [{'tail' if 'fmpc_tail_kernel' in n1 else 'no tail'}]")
        if not same:
            explain(a0, a1)
print("all cases agree" if bad == 0 else f"{bad} cases differ")
sys.exit(1 if bad else 0)
