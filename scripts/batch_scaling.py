"""Solve latency of the cart-pole workload (T = 100, max_iter = 8) against the batch size, per kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmpc_amd
from nmpc_amd import workloads
for kernel in (sys.argv[1:] or ["quad", "2w"]):
    os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
    for B in (256, 1024, 2048, 3072, 3968, 4096, 4160, 8192, 16384):
        wl = workloads.cartpole_batch(B=B, T=100, seed=1234)
        s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
        s.config().print_level = 0; s.config().max_iter = 8
        ms = []
        for _ in range(6):
            s.solve(wl.t0, wl.x0, wl.u_init)
            ms.append(s.computationDuration().opt)
        print(f"{s.kernelName():24s} B {B:6d}  kernel ms {min(ms):.3f}  iterations max {int(s.iters().max())}", flush=True)
