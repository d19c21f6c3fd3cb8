import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before torch initialises the HIP runtime
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import nmpc_amd
from nmpc_amd import workloads
wl = workloads.cartpole_batch(B=4096, T=100, seed=1234)
prob = nmpc_amd.make_problem(wl.model)
d = [torch.from_numpy(a).cuda() for a in (wl.t0, wl.x0, wl.u_init)]
for n in (4, 8, 12, 16, 24):
    pool = nmpc_amd.DDPSolverPool(prob, wl.B, n_handles=n)
    c = pool.config(); c.print_level, c.horizon_steps, c.max_iter, c.ragged_schedule = 0, wl.T, 500, 0
    pool.applyConfig()
    for _ in range(n): pool.submit(*[t.data_ptr() for t in d])
    pool.synchronize()
    best = 0
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(64): pool.submit(*[t.data_ptr() for t in d])
        pool.synchronize()
        best = max(best, 64 / (time.perf_counter() - t0))
    print(f"handles {n:2d} (GPU_MAX_HW_QUEUES {os.environ.get('GPU_MAX_HW_QUEUES')}): {best:.1f} batches/s = {best * 19.95:.0f} batch-iterations/s", flush=True)
