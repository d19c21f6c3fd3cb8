"""Sequential vs step-size-parallel line search of the unconstrained quad kernel (Configuration::line_search_fan_out) on C2."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmpc_amd
from nmpc_amd import workloads
wl = workloads.cartpole_batch(B=4096, T=100, seed=1234)
for name, cfg in (("nominal (max_iter 8)", dict(max_iter=8)), ("m1 (50 forced)", dict(max_iter=50, k_rel_norm_thre=0.0, cost_update_thre=-1e300)),
                  ("m1 (8 forced)", dict(max_iter=8, k_rel_norm_thre=0.0, cost_update_thre=-1e300)), ("m2 (to convergence)", dict(max_iter=500))):
    for fan in (2, 1):
        s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
        c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.line_search_fan_out = fan
        for k, v in cfg.items():
            setattr(c, k, v)
        for _ in range(3):
            s.solve(wl.t0, wl.x0, wl.u_init)
        ms = s.computationDuration().opt
        it = s.iters(); tr = s.trace()
        nfw = tr[:, 1:, 11].sum() / max(it.sum(), 1)
        print(f"{name:22s} fan_out={fan}: kernel {ms:8.3f} ms, {it.sum() / wl.B / ms * 1e3:8.1f} batch-it/s, mean iterations {it.mean():6.2f}, max {it.max()}, forward trials per iteration {nfw:.2f}, status {np.bincount(s.status() + 1, minlength=3)}")
