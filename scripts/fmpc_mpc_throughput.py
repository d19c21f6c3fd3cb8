"""Closed-loop throughput of the device-resident FMPC loop (nmpc_hip_fmpc_mpc_run): the reference's two FMPC tests as batches.
usage: fmpc_mpc_throughput.py [B] [ticks]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmpc_amd import fmpc as F  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rng = np.random.default_rng(0)
for name, prob, T, max_iter, sim_dt, sub, fb, x0 in (
        ("cart-pole swing-up (TestFmpcCartPole: T 200, max_iter 5, 2 x 2 ms plant steps per tick, K0 feedback)",
         F.FmpcProblemCartPole(0.01), 200, 5, 0.002, 2, True,
         np.tile([0.0, np.pi, 0.0, 0.0], (B, 1)) + 0.05 * rng.standard_normal((B, 4))),
        ("Van der Pol oscillator (TestFmpcOscillator: T 400, max_iter 3, 5 ms plant step per tick)",
         F.FmpcProblemOscillator(0.01), 400, 3, 0.005, 1, False,
         np.tile([0.0, 1.0], (B, 1)) + np.abs(0.1 * rng.standard_normal((B, 2))))):
    s = F.FmpcSolverBatch(prob, B, T)
    s.config().max_iter = max_iter
    var = F.Variable.make(prob, T, B)
    var.reset(0.0, 0.0, 0.0, 1.0, 1.0)
    s.setVariable(var)
    s.mpcRun(0.0, x0, 5, sim_dt, sub, fb)  # warm-up (graph capture)
    s.setVariable(var)
    t = time.perf_counter()
    log = s.mpcRun(0.0, x0, ticks, sim_dt, sub, fb)
    dt = time.perf_counter() - t
    it = log["iters"]
    print(f"{name}: B={B}, {ticks} ticks in {dt:.3f} s = {ticks / dt:.0f} ticks/s = {B * ticks / dt / 1e6:.2f} M solves/s; "
          f"iterations per solve {it.mean():.2f} (first tick {it[:, 0].mean():.1f}, last {it[:, -1].mean():.1f}), "
          f"status counts {dict(zip(*np.unique(log['status'], return_counts=True)))}")
