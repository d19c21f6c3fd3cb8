"""Streamed solve (nmpc_hip_ddp_solve_stream): N cart-pole instances to convergence (default Configuration) through the 4096 slots of ONE
handle; batch-iterations/s-equivalent = instance-iterations / 4096 / device time.   python scripts/stream_throughput.py [N] [slots] [spans ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
spans = [int(a) for a in sys.argv[3:]] or [16]
wl = workloads.cartpole_batch(B=N, T=100, seed=1234)
for span in spans:
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), S)
    c = s.config(); c.print_level, c.horizon_steps, c.max_iter, c.trace_level = 0, wl.T, 500, 0
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        r = s.solveStream(wl.t0, wl.x0, wl.u_init, span=span)
        wall = time.perf_counter() - t0
        it = int(r.iters.sum())
        rate = it / S / (r.device_ms * 1e-3)
        best = max(best or 0, rate)
    st = {int(k): int(v) for k, v in zip(*np.unique(r.status, return_counts=True))}
    print(f"N {N} slots {S} span {span:3d}: {r.rounds} rounds, device {r.device_ms:8.2f} ms (wall {1e3 * wall:8.2f}), {it} instance-iterations "
          f"(mean {it / N:.2f}) -> {best:8.0f} batch-it/s-equivalent; status {st}", flush=True)
