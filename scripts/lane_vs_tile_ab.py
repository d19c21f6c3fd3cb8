"""Shapes the tile kernel takes (5 <= n <= 15): the lane-per-instance kernel against it at large batches, with and without a box.
The tile kernel holds <= 35 instances per CU and round, the lane kernel 64 per wavefront and several wavefronts per SIMD: for small
models the lane kernel's time barely grows with the batch while the tile kernel's grows linearly."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nmpc_amd
from nmpc_amd import workloads as W

CASES = [("planar_vtol", lambda B, con: W.planar_vtol_batch(B=B, T=60, seed=33, constrained=con), 6),
         ("quadrotor", lambda B, con: W.quadrotor_batch(B=B, T=50, seed=5, constrained=con), 4),
         ("manipulator", lambda B, con: W.manipulator_batch(B=B, T=30, seed=5, constrained=con), 4),
         ("centroidal", lambda B, con: W.centroidal_batch(B=B, T=100, seed=5), 4)]
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, mk, iters in CASES:
    if only and only not in name:
        continue
    for B in ((2048, 4096, 8192, 16384, 32768) if name != "centroidal" else (1024, 4096)):
        for con in ((False, True) if name != "centroidal" else (False,)):
            row = []
            for kernel in ("auto", "tile64", "1w", "2w"):
                wl = mk(B, con)
                s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
                c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = iters; c.with_input_constraint = con
                if con:
                    s.setInputLimits(*wl.limits)
                s.setKernel(kernel)
                ms = []
                for _ in range(3):
                    s.solve(wl.t0, wl.x0, wl.u_init); ms.append(s.computationDuration().opt)
                row.append(f"{kernel}: {s.kernelName().replace('ddp_solve_', '').replace('_kernel', '')} {min(ms):8.3f} ms")
            print(f"{name:12s} B {B:5d} {'box          ' if con else 'unconstrained'}: " + " | ".join(row), flush=True)
