"""n = 6, m = 2 (planar VTOL): which kernel family is the fastest, with and without the rotor-thrust box, at two batch sizes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nmpc_amd
from nmpc_amd import workloads

for B in (1024, 8192, 32768):
    for con in (False, True):
        row = []
        for kernel in ("auto", "1w", "2w", "tile64"):
            wl = workloads.planar_vtol_batch(B=B, T=60, seed=33, constrained=con)
            s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
            c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = 6; c.with_input_constraint = con
            if con:
                s.setInputLimits(*wl.limits)
            s.setKernel(kernel)
            ms = []
            for _ in range(5):
                s.solve(wl.t0, wl.x0, wl.u_init); ms.append(s.computationDuration().opt)
            row.append(f"{kernel}: {s.kernelName().replace('ddp_solve_', '').replace('_kernel', '')} {min(ms):.3f} ms")
        print(f"planar_vtol B {B:5d} {'box' if con else 'unconstrained'}: " + " | ".join(row), flush=True)
