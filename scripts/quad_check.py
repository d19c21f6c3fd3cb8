"""A/B check of the quad kernel against the two-wave kernel (NMPC_HIP_DDP_KERNEL=quad / 2w) on a few small workloads:
status / iteration counts / BoxQP return codes must be identical, values agree to rounding.  (The parity tests proper
are tests/test_gpu_parity.py::test_quad_kernel_vs_oracle and ::test_two_wave_and_single_wave_kernels_agree.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import nmpc_amd
from nmpc_amd import workloads
sys.path.insert(0, "tests")

def run(wl, kernel, **cfg):
    os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
    prob = nmpc_amd.make_problem(wl.model)
    s = nmpc_amd.DDPSolverBatch(prob, wl.B)
    c = s.config(); c.print_level = 0; c.horizon_steps = wl.T
    for k, v in cfg.items(): setattr(c, k, v)
    if wl.limits is not None: s.setInputLimits(*wl.limits)
    s.solve(wl.t0, wl.x0, wl.u_init)
    return dict(name=s.kernelName(), status=s.status(), iters=s.iters(), X=s.X(), U=s.U(), kff=s.kff(), Kfb=s.Kfb(),
                trace=s.trace(), qp=s.qpRetval() if wl.limits is not None else None)

def err(a, b): return float((np.abs(a - b) / (1 + np.abs(b))).max())
cases = [("cartpole", workloads.cartpole_batch(B=200, T=60, seed=11), dict(max_iter=30)),
         ("cartpole-con", workloads.cartpole_batch(B=200, T=60, seed=11, constrained=True), dict(max_iter=30, with_input_constraint=True)),
         ("cartpole-T37", workloads.cartpole_batch(B=33, T=37, seed=5), dict(max_iter=20)),
         ("cartpole-reg2", workloads.cartpole_batch(B=64, T=50, seed=6), dict(max_iter=20, reg_type=2)),
         ("bipedal", workloads.bipedal_batch(B=130, T=40, seed=12), dict(max_iter=30))]
for name, wl, cfg in cases:
    a = run(wl, "quad", **cfg); b = run(wl, "2w", **cfg)
    print(name, a["name"], b["name"], "status eq", np.array_equal(a["status"], b["status"]), "iters eq", int((a["iters"] != b["iters"]).sum()), "of", wl.B,
          "X", err(a["X"], b["X"]), "U", err(a["U"], b["U"]), "k", err(a["kff"], b["kff"]), "K", err(a["Kfb"], b["Kfb"]),
          "qp eq" if a["qp"] is None else ("qp neq %d" % int((a["qp"] != b["qp"]).sum())), flush=True)
