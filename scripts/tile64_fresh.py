"""First solve of a fresh process on the fp64 tile kernel, then the wave-per-instance kernel: any difference?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402

model, T, B = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("manipulator", 30, 8192)


def solver(kernel=None):
    os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
    if kernel:
        os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
    wl = workloads.quadrotor_batch(B=B, T=T, seed=1234) if model == "quadrotor" else workloads.manipulator_batch(B=B, T=T, seed=1234)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = wl.T
    c.max_iter = 8
    return wl, s


wl, s = solver()
runs = []
for rep in range(4):
    s.solve(wl.t0, wl.x0, wl.u_init)
    runs.append((s.iters().copy(), s.X().copy(), s.trace().copy(), s.computationDuration().opt))
wl, ref = solver("wpi")
ref.solve(wl.t0, wl.x0, wl.u_init)
r_it, r_X, r_tr = ref.iters(), ref.X(), ref.trace()
np.set_printoptions(linewidth=250, precision=6)
for rep, (it, X, tr, ms) in enumerate(runs):
    bad = np.flatnonzero(it != r_it)
    print(f"rep {rep}: {ms:.2f} ms, iteration counts differ on {bad.size} {bad[:12]}", flush=True)
    if bad.size:
        print("   bad slots histogram", np.bincount(bad % 32, minlength=32))
        print("   bad groups", np.unique(bad // 32)[:20], "count", np.unique(bad // 32).size)
        b = bad[0]
        print("   instance", b, "tile64 trace\n", tr[b, :it[b] + 1], "\n   wpi trace\n", r_tr[b, :r_it[b] + 1])
