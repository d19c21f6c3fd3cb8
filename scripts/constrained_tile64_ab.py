"""Box-constrained solves of the 9 <= n <= 15 shapes: fp64 tile kernel against the wave-per-instance kernel, full batch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nmpc_amd
from nmpc_amd import workloads

for model, T in (("quadrotor", 50), ("manipulator", 30), ("planar_vtol", 60)):
    res = {}
    for kernel in (None, "wpi", "tile64"):
        os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
        if kernel:
            os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
        wl = (workloads.quadrotor_batch(B=8192, T=T, seed=31, constrained=True) if model == "quadrotor" else
              workloads.manipulator_batch(B=8192, T=T, seed=32, constrained=True) if model == "manipulator" else
              workloads.planar_vtol_batch(B=8192, T=T, seed=33, constrained=True))
        s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
        c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = 4; c.with_input_constraint = True
        s.setInputLimits(*wl.limits)
        ms = []
        for _ in range(3):
            s.solve(wl.t0, wl.x0, wl.u_init); ms.append(s.computationDuration().opt)
        res[kernel] = (s.iters().copy(), s.status().copy(), s.X().copy())
        print(f"{model} box, B 8192, max_iter 4: {s.kernelName()} {min(ms):.2f} ms, iterations {int(s.iters().sum())}", flush=True)
    same = np.array_equal(res[None][0], res["wpi"][0]) and np.array_equal(res[None][1], res["wpi"][1])
    print(f"   decisions equal: {same}, max |dX| {np.abs(res[None][2] - res['wpi'][2]).max():.2e}")
