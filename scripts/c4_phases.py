"""Kernel time and phase split of the fp32 quadrotor workload (BASELINE config 4) on the fp64 tile kernel's float instantiation and on
the fp32 tile kernel (NMPC_HIP_DDP_KERNEL=tile32), next to the fp64 solve of the same shape; max_iter as given (default 8)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402

mi = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for fp32, kernel in ((True, None), (True, "tile32"), (False, None)):
    os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
    if kernel:
        os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
    wl = workloads.quadrotor_batch(B=8192, T=50, seed=1234, fp32=fp32)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = wl.T
    c.max_iter = mi
    if fp32:
        c.cost_update_thre = 1e-3
    ms, bw, fw = [], [], []
    for _ in range(6):
        s.solve(wl.t0, wl.x0, wl.u_init)
        d = s.computationDuration()
        ms.append(d.opt)
        bw.append(d.backward)
        fw.append(d.forward)
    its = int(s.iters().sum())
    tr = s.trace()
    print(f"{wl.model:14s} {s.kernelName():24s} kernel ms min {min(ms):.3f} median {np.median(ms):.3f} backward {np.median(bw):.3f} forward "
          f"{np.median(fw):.3f}; {its} instance-iterations ({its / wl.B / min(ms) * 1e3:.0f} batch-it/s), max iterations {s.iters().max()}, "
          f"forward trials / iteration {tr[:, 1:, 11].sum() / max(its, 1):.2f}", flush=True)
