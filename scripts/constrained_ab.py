"""Kernel time of box-constrained cart-pole solves (the MPC callers' shape): B = 4096, T = 100 / 200, +-15 N, max_iter 8 / 3.
usage: [NMPC_HIP_DDP_LIB=...] python scripts/constrained_ab.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import nmpc_amd
from nmpc_amd import workloads

for T, max_iter in ((100, 8), (200, 3)):
    wl = workloads.cartpole_batch(B=4096, T=T, seed=1234)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = T
    c.max_iter = max_iter
    c.with_input_constraint = True
    s.setInputLimits(np.array([-15.0]), np.array([15.0]))
    ms = []
    for _ in range(6):
        s.solve(wl.t0, wl.x0, wl.u_init)
        ms.append(s.computationDuration().opt)
    print(f"cart-pole +-15 N, B 4096, T {T}, max_iter {max_iter}: kernel {np.median(ms[1:]):.3f} ms ({s.kernelName()}), "
          f"instance-iterations {int(s.iters().sum())}")
