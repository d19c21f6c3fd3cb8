"""Phase breakdown of the fp64 tile kernel at the bench shapes (shader-clock shares from the model wave's phase stamps)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402


def run(model, B, T, kernel=None, group=0, forced=False, max_iter=8):
    os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
    os.environ.pop("NMPC_HIP_DDP_TILE64_GROUP", None)
    if kernel:
        os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
    if group:
        os.environ["NMPC_HIP_DDP_TILE64_GROUP"] = str(group)
    wl = workloads.quadrotor_batch(B=B, T=T, seed=1234) if model == "quadrotor" else workloads.manipulator_batch(B=B, T=T, seed=1234)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = wl.T
    c.max_iter = max_iter
    if forced:
        c.k_rel_norm_thre = 0.0
        c.cost_update_thre = -1e300
    ms = []
    for _ in range(3):
        s.solve(wl.t0, wl.x0, wl.u_init)
        ms.append(s.computationDuration().opt)
    d = s.computationDuration()
    its = int(s.iters().sum())
    tr = s.trace()
    nfw = tr[:, 1:, 11].sum() / max(its, 1)
    print(f"{model:12s} B {B:5d} T {T} {'forced' if forced else 'nominal'} max_iter {max_iter} group {group}: {s.kernelName():24s} {min(ms):7.3f} ms "
          f"(backward {d.backward:.3f} forward {d.forward:.3f}) {its} instance-iterations, {its / B / min(ms) * 1e3:7.0f} batch-it/s, "
          f"iters mean {s.iters().mean():.2f} max {s.iters().max()}, forward trials / iteration {nfw:.2f}", flush=True)


if __name__ == "__main__":
    for model, T in (("manipulator", 30), ("quadrotor", 50)):
        run(model, 8192, T)
        run(model, 8192, T, forced=True)
        run(model, 8192, T, forced=True, max_iter=2)
        run(model, 8192, T, group=16)
        run(model, 8192, T, group=8)
        run(model, 4096, T)
        run(model, 2048, T)
        run(model, 256, T)
        run(model, 8192, T, kernel="wpi")
