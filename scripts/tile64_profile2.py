import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd
from nmpc_amd import workloads

def run(model, B, T, n_alpha=11, max_iter=2, group=0):
    os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
    os.environ.pop("NMPC_HIP_DDP_TILE64_GROUP", None)
    if group:
        os.environ["NMPC_HIP_DDP_TILE64_GROUP"] = str(group)
    wl = workloads.quadrotor_batch(B=B, T=T, seed=1234) if model == "quadrotor" else workloads.manipulator_batch(B=B, T=T, seed=1234)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = wl.T
    c.max_iter = max_iter
    c.k_rel_norm_thre = 0.0
    c.cost_update_thre = -1e300
    if n_alpha != 11:
        c.alpha_list = np.array([1.0, 0.5, 0.25, 0.125, 0.06, 0.03, 0.015, 0.007, 0.003, 0.0015, 0.001][:n_alpha])
    ms = []
    for _ in range(3):
        s.solve(wl.t0, wl.x0, wl.u_init)
        ms.append(s.computationDuration().opt)
    d = s.computationDuration()
    print(f"{model:12s} B {B} T {T} max_iter {max_iter} n_alpha {n_alpha} group {group}: {min(ms):7.3f} ms (backward {d.backward:.3f} forward {d.forward:.3f})", flush=True)

for model, T in (("manipulator", 30), ("quadrotor", 50)):
    for n_alpha in (11, 2, 1):
        run(model, 8192, T, n_alpha)
    run(model, 8192, T, 1, max_iter=0)
    run(model, 256, T, 11)
    run(model, 256, T, 1)
    run(model, 256, T, 1, max_iter=0)
