"""Kernel time per solve (max_iter 8) over the batch size: fp64 tile kernel against the wave-per-instance kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nmpc_amd
from nmpc_amd import workloads

for model, T in (("manipulator", 30), ("quadrotor", 50)):
    for B in (1, 16, 64, 256, 1024, 2048, 4096, 8192, 8200, 12288, 16384, 32768):
        row = []
        for kernel in (None, "wpi"):
            os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
            if kernel:
                os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
            wl = workloads.quadrotor_batch(B=B, T=T, seed=1234) if model == "quadrotor" else workloads.manipulator_batch(B=B, T=T, seed=1234)
            s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
            c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = 8
            ms = []
            for _ in range(3):
                s.solve(wl.t0, wl.x0, wl.u_init); ms.append(s.computationDuration().opt)
            row.append((s.kernelName(), min(ms), int(s.iters().sum())))
            del s
        (k0, t0, i0), (k1, t1, i1) = row
        print(f"{model:12s} B {B:6d}: {k0} {t0:8.3f} ms ({i0 / B / t0 * 1e3:7.0f} it/s)   {k1} {t1:8.3f} ms ({i1 / B / t1 * 1e3:7.0f} it/s)   ratio {t1 / t0:5.2f}", flush=True)
