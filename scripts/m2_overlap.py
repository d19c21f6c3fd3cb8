"""Back-to-back solves to convergence (SURVEY 8(d) mode M2) on ONE handle against the same batches dealt round-robin to
several handles, each with its own stream: a converging batch ends with a tail of a few instances that run for hundreds of
iterations on a few CUs; with the next batches queued on other streams the hardware fills the vacated CUs with their workgroups.
    python scripts/m2_overlap.py [n_batches] [handles ...]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before torch initialises the HIP runtime

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402  (device buffers)

import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402


def make(wl):
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = wl.T
    c.max_iter = 500
    return s


def run(wl, n_batches, n_handles, d):
    pool = [make(wl) for _ in range(n_handles)]
    for s in pool:  # warm-up: first launch, buffers
        s.solveDevice(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr())
    for s in pool:
        s.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n_batches):
        pool[k % n_handles].solveDevice(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr())
    for s in pool:
        s.synchronize()
    dt = time.perf_counter() - t0
    return dt, pool


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    handles = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
    wl = workloads.cartpole_batch(B=4096, T=100, seed=1234)
    dev = torch.device("cuda", 0)
    d = (torch.from_numpy(wl.t0).to(dev), torch.from_numpy(wl.x0).to(dev), torch.from_numpy(wl.u_init).to(dev))
    ref = None
    out = {}
    for nh in handles:
        dt, pool = run(wl, n_batches, nh, d)
        X, it, st = pool[-1].X(), pool[-1].iters(), pool[-1].status()
        if ref is None:
            ref = (X.copy(), it.copy(), st.copy())
        same = np.array_equal(X, ref[0]) and np.array_equal(it, ref[1]) and np.array_equal(st, ref[2])
        out[nh] = n_batches * wl.B / dt
        print(f"{n_batches} batches of {wl.B} cart-pole solves to convergence on {nh} handle(s) / stream(s): {1e3 * dt:8.2f} ms  = "
              f"{n_batches * wl.B / dt / 1e3:8.1f} k solves/s, {n_batches * float(it.sum()) / wl.B / dt:8.0f} batch-iterations/s; "
              f"results bit-identical to the single-handle run: {same}", flush=True)
        del pool
    return out


if __name__ == "__main__":
    main()
