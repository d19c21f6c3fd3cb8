"""Repeated FMPC solves of the same batch (the three-launch sequence: fused Riccati kernel with producer waves, delta, tail) must
return the same bits every time — under the product build and, with NMPC_HIP_DDP_LIB pointing at one, under the wave-timing fuzz
builds (every barrier of these kernels is a syncThreadsFuzzed).    usage: fmpc_determinism.py [repetitions]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmpc_amd import fmpc as F  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
CASES = [("fmpc_cartpole", 4096, 200, 5), ("fmpc_cartpole", 1000, 37, 8), ("fmpc_cartpole", 512, 200, 40), ("fmpc_oscillator", 2048, 100, 10)]
total = 0
for model, B, T, max_iter in CASES:
    prob = {"fmpc_cartpole": F.FmpcProblemCartPole, "fmpc_oscillator": F.FmpcProblemOscillator}[model](0.01)
    rng = np.random.default_rng(7)
    x0 = np.zeros((B, prob.state_dim))
    x0[:, 0] = rng.uniform(-1, 1, B)
    x0[:, 1] = rng.uniform(-0.3, 0.3, B) + (np.pi if prob.state_dim == 4 and T != 200 else 0.0)
    x0[::13, 0] = np.nan  # error exits as well
    s = F.FmpcSolverBatch(prob, B, T)
    s.config().max_iter = max_iter
    var = F.Variable.make(prob, T, B)
    digests = []
    for _ in range(reps):
        var.reset(0.0, 0.0, 0.0, 1.0, 1.0)
        try:
            s.solve(0.0, x0, var)
        except RuntimeError:
            pass
        h = hashlib.sha256()
        for a in list(s.variable().arrays()) + list(s.deltaVariable().arrays()) + [s.status(), s.iters(), s.traceDataList(), s.barrierEps(), s.partials()]:
            h.update(np.ascontiguousarray(a).tobytes())
        digests.append(h.hexdigest())
    differ = sum(d != digests[0] for d in digests)
    total += differ
    print(f"{model:16s} B={B:5d} T={T:3d} max_iter={max_iter:2d}  {','.join(k.replace('fmpc_', '').replace('_kernel', '') for k in s.kernelNames())}: "
          f"{reps} repetitions, {differ} differ from the first; status counts {np.bincount(s.status(), minlength=7).tolist()}")
print("TOTAL differing runs:", total)
sys.exit(1 if total else 0)
