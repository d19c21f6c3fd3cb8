"""Per-phase cycle counters of a -DNMPC_AMD_PROFILE_WPI build of the wave-per-instance kernel (instance 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmpc_amd
from nmpc_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name, wl in (("quadrotor", workloads.quadrotor_batch(B=B, T=50, seed=1234)),
                 ("manipulator", workloads.manipulator_batch(B=B, T=30, seed=1234)),
                 ("centroidal", workloads.centroidal_batch(B=min(B, 256), T=100, seed=1234))):
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = 4
    for _ in range(2):
        s.solve(wl.t0, wl.x0, wl.u_init)
    q = s.qpFreeMask()[0, :6].astype(np.float64) * 16.0
    it = int(s.iters()[0])
    names = ("initial rollout", "linearise", "backward", "line search (all alphas)", "adopt candidate", "write-out")
    print(f"{name}: kernel {s.computationDuration().opt:.2f} ms, {s.kernelName()}, instance 0 ran {it} iterations")
    for n, v in zip(names, q):
        print(f"   {n:26s} {v:12.0f} cycles  ({v / max(it, 1):10.0f} per iteration)")
