"""Repeated solves of the same batch on the fp64 tile kernel: every run must give the same bits (and the wave-per-instance kernel's decisions)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402


def solver(model, B, T, kernel=None, group=0, max_iter=8):
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
    return wl, s


for model, T in (("manipulator", 30), ("quadrotor", 50)):
    wl, ref = solver(model, 8192, T, kernel="wpi")
    ref.solve(wl.t0, wl.x0, wl.u_init)
    r_it, r_X, r_tr = ref.iters().copy(), ref.X().copy(), ref.trace().copy()
    for group in (0, 16):
        wl, s = solver(model, 8192, T, group=group)
        for rep in range(6):
            s.solve(wl.t0, wl.x0, wl.u_init)
            it, X, tr = s.iters(), s.X(), s.trace()
            bad = np.flatnonzero(it != r_it)
            dx = np.abs(X - r_X).reshape(len(it), -1).max(axis=1)
            badx = np.flatnonzero(dx > 1e-9)
            print(f"{model} group {group} rep {rep}: {s.computationDuration().opt:.2f} ms, iteration counts differ from wpi on {bad.size} instances "
                  f"{bad[:12]}, X differs on {badx.size} {badx[:12]}", flush=True)
            if bad.size:
                b = bad[0]
                print("   first bad instance", b, "group", b // 32, "slot", b % 32, "iters", it[b], "vs", r_it[b])
                np.set_printoptions(linewidth=250, precision=6)
                print("   tile64 trace\n", tr[b, :it[b] + 1])
                print("   wpi trace\n", r_tr[b, :r_it[b] + 1])
                print("   bad slots histogram", np.bincount(bad % 32, minlength=32))
                print("   bad groups", np.unique(bad // 32)[:20], "count", np.unique(bad // 32).size)
