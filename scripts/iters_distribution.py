"""Iteration counts to convergence (default Configuration, max_iter 500) of the tile-kernel workloads c5, c4f64, centroidal: mean, percentiles, and the
mean over workgroup groups of the maximum — how ragged they are, i.e. what a resumable / streamed schedule could win (round 6: nothing).
    python scripts/iters_distribution.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, nmpc_amd
from nmpc_amd import workloads
for name, wl in (("c5", workloads.manipulator_batch(B=8192, T=30, seed=1234)), ("c4f64", workloads.quadrotor_batch(B=8192, T=50, seed=1234)),
                 ("centroidal", workloads.centroidal_batch(B=4096, T=100, seed=1234))):
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config(); c.print_level, c.horizon_steps, c.max_iter, c.trace_level = 0, wl.T, 500, 0
    s.solve(wl.t0, wl.x0, wl.u_init)
    it = s.iters(); st = s.status()
    G = 32 if name != "centroidal" else 16
    gmax = np.array([it[i:i + G].max() for i in range(0, wl.B, G)])
    print(name, "kernel ms", round(s.computationDuration().opt, 2), "iters mean", it.mean().round(2), "median", np.median(it), "p90", np.percentile(it, 90), "p99", np.percentile(it, 99), "max", it.max(),
          "| mean over groups of the max", gmax.mean().round(1), "| status", {int(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))}, flush=True)
