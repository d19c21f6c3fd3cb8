"""First-light check on a GPU box: cart-pole batch through the C-ABI vs the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmpc_amd, oracle

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = 100
rng = np.random.default_rng(1234)
x0 = np.stack([rng.uniform(-1, 1, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-1, 1, B)], 1)
u0 = np.zeros((B, T, 1))

s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemCartPole(), B)
s.config().print_level = 0
t = time.time()
ok = s.solve(0.0, x0, u0)
print("gpu solve wall", time.time() - t, "dur", s.computationDuration())
t = time.time()
ok = s.solve(0.0, x0, u0)
print("gpu solve wall (2nd)", time.time() - t, "dur", s.computationDuration())
cfg = oracle.default_config()
ref = oracle.solve_batch("cartpole", cfg, x0, u0, n_threads=8, want_alpha_hist=True)
print("cpu oracle sec", ref.seconds, "total iters", ref.total_iters)
print("status equal:", np.array_equal(s.status(), ref.status), "iters equal:", np.array_equal(s.iters(), ref.iters))
print("gpu iters mean", s.iters().mean(), "cpu", ref.iters.mean())
bad = np.nonzero(s.iters() != ref.iters)[0]
print("mismatching instances:", bad[:10], s.iters()[bad[:10]], ref.iters[bad[:10]])
print("max|dX|", np.abs(s.X() - ref.X).max(), "max|dU|", np.abs(s.U() - ref.U).max(),
      "max|dcost|", np.abs(s.cost() - ref.cost).max())
print("max|dk|", np.abs(s.kff() - ref.k).max(), "max|dK|", np.abs(s.Kfb() - ref.K).max())
tr = s.trace()
print("trace row1 gpu", tr[0, 1], "\n")
r0 = oracle.solve("cartpole", cfg, x0[0], u0[0])
print("trace row1 cpu", r0.trace[1])
n = r0.trace.shape[0]
print("trace max rel diff inst0:", np.abs(tr[0, :n] - r0.trace).max())
