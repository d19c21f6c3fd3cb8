"""C3 (bipedal, n = 2, m = 1, T = 300, 1024 instances): the kernel families the shape can run on, single batch and pooled
(the A/B that was missing).   python scripts/c3_kernel_ab.py [B] [max_iter]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch  # noqa: E402
import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
MI = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = workloads.bipedal_batch(B=B, T=300, seed=1234)
prob = nmpc_amd.make_problem(wl.model)
d = [torch.from_numpy(a).cuda() for a in (wl.t0, wl.x0, wl.u_init)]
ptr = [t.data_ptr() for t in d]
ref = None
for kern in ("quad", "2w", "1w"):
    s = nmpc_amd.DDPSolverBatch(prob, B)
    c = s.config(); c.print_level, c.horizon_steps, c.max_iter = 0, wl.T, MI
    s.setKernel(kern)
    for _ in range(3):
        s.solveDevice(*ptr)
    s.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        s.solveDevice(*ptr)
    s.synchronize()
    dt = time.perf_counter() - t0
    it = s.iters()
    out = (s.iters().copy(), s.status().copy(), s.X().copy())
    if ref is None:
        ref = out
    same = np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
    dx = float(np.abs(out[2] - ref[2]).max())
    print(f"{kern:5s} {s.kernelName():26s} single batch: {n * it.sum() / B / dt:8.0f} batch-it/s  ({1e3 * dt / n:.3f} ms per solve, mean it {it.mean():.2f})"
          f"  decisions == quad: {same}, max |dX| {dx:.2e}", flush=True)
    for nh in (4, 8, 16):
        pool = nmpc_amd.DDPSolverPool(prob, B, n_handles=nh)
        pc = pool.config(); pc.print_level, pc.horizon_steps, pc.max_iter = 0, wl.T, MI
        pool.applyConfig()
        for h in pool.solvers:
            h.setKernel(kern)
        for _ in range(2 * nh):
            pool.submit(*ptr)
        pool.synchronize()
        n = 50 * nh
        t0 = time.perf_counter()
        for _ in range(n):
            pool.submit(*ptr)
        pool.synchronize()
        dt = time.perf_counter() - t0
        print(f"      pooled x{nh:2d}: {n * it.sum() / B / dt:8.0f} batch-it/s", flush=True)
        del pool
