"""fp64 tile kernel, chunked linearisation (round 4) against one timestep per pass (NMPC_HIP_DDP_TILE64_CHUNK=1: round 3's
schedule) and against the wave-per-instance kernel, over the batch size.  Kernel time per solve (max_iter 8), min of 4."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nmpc_amd
from nmpc_amd import workloads

models = sys.argv[1:] or ["manipulator", "quadrotor"]
for model in models:
    T = 30 if model == "manipulator" else 50
    for B in (16, 64, 256, 512, 1024, 2048, 4096, 8192, 8200, 12288, 16384):
        row = []
        ref = None
        for kernel, chunk in (("tile64", None), ("tile64", "1"), ("wpi", None)):
            os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
            os.environ.pop("NMPC_HIP_DDP_TILE64_CHUNK", None)
            if chunk:
                os.environ["NMPC_HIP_DDP_TILE64_CHUNK"] = chunk
            wl = workloads.quadrotor_batch(B=B, T=T, seed=1234) if model == "quadrotor" else workloads.manipulator_batch(B=B, T=T, seed=1234)
            s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
            c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = 8
            ms = []
            for _ in range(4):
                s.solve(wl.t0, wl.x0, wl.u_init); ms.append(s.computationDuration().opt)
            if kernel == "tile64":
                if ref is None:
                    ref = (s.X().copy(), s.iters().copy())
                else:
                    assert np.array_equal(ref[0], s.X()) and np.array_equal(ref[1], s.iters()), "chunked and unchunked sweeps differ"
            row.append((min(ms), int(s.iters().sum())))
            del s
        (t0, i0), (t1, i1), (t2, i2) = row
        print(f"{model:12s} B {B:6d}: tile64 {t0:8.3f} ms ({i0 / B / t0 * 1e3:7.0f} it/s) | chunk=1 {t1:8.3f} ms | wpi {t2:8.3f} ms ({i2 / B / t2 * 1e3:7.0f} it/s)"
              f"   tile64/wpi {t0 / t2:5.2f}  chunked/unchunked {t0 / t1:5.2f}", flush=True)
