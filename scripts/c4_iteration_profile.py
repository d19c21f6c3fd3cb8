"""c4 (quadrotor 8192 x T 50, fp32, cost_update_thre 1e-3): kernel time as a function of max_iter for the fp32 tile kernel and for the
fp64 tile kernel's float instantiation — the differences are the cost of each iteration as the batch thins out (instances that
have converged leave the sweeps).     python scripts/c4_iteration_profile.py [B]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402


def run(kernel, B, max_iter, fp32=True, thre=1e-3):
    if kernel:
        os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
    else:
        os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
    wl = workloads.quadrotor_batch(B=B, T=50, seed=1234, fp32=fp32)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = wl.T
    c.max_iter = max_iter
    c.cost_update_thre = thre
    best = None
    for _ in range(4):
        s.solve(wl.t0, wl.x0, wl.u_init)
        d = s.computationDuration()
        if best is None or d.opt < best[0]:
            best = (d.opt, d.backward, d.forward)
    it = s.iters()
    return s.kernelName(), best, float(it.sum()) / B, int((it >= max_iter).sum()) if max_iter else 0


MI = [int(v) for v in os.environ.get("MI", "1,2,3,4,5,6,7,8").split(",")]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    for kernel, fp32, thre in ((None, True, 1e-3), ("tile64", True, 1e-3), (None, False, 1e-7)):
        prev = 0.0
        for mi in MI:
            name, (opt, bw, fw), mean_it, at_cap = run(kernel, B, mi, fp32, thre)
            print(f"{name:28s} fp32 {int(fp32)} max_iter {mi}: kernel {opt:.3f} ms (+{opt - prev:.3f}) backward {bw:.3f} forward {fw:.3f}; "
                  f"mean iterations {mean_it:.2f}, instances at the cap {at_cap} -> {mean_it / max(opt, 1e-9) * 1e3:.0f} it/s", flush=True)
            prev = opt


if __name__ == "__main__":
    main()
