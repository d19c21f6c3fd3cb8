import numpy as np, sys
import nmpc_amd, oracle
from nmpc_amd import workloads
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
mi=int(sys.argv[2]) if len(sys.argv)>2 else 4
wl = workloads.quadrotor_batch(B=B, T=50, seed=1, fp32=True)
solver = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B, device=0)
solver.config().print_level = 0
solver.config().horizon_steps = wl.T
solver.config().max_iter = mi
solver.solve(wl.t0, wl.x0, wl.u_init)
print("kernel", solver.kernelName(), "ms", solver.computationDuration().opt)
cfg = oracle.default_config(horizon_steps=wl.T, max_iter=mi)
ref = oracle.solve_batch(wl.model, cfg, wl.x0, wl.u_init, t0=wl.t0, n_threads=8, want_alpha_hist=True)
print("status eq", (solver.status()==ref.status).mean(), "iters eq", (solver.iters()==ref.iters).mean())
print("gpu iters", np.bincount(solver.iters()), "ref", np.bincount(ref.iters))
for name, got, want in (("X", solver.X(), ref.X), ("U", solver.U(), ref.U), ("cost", solver.cost(), ref.cost), ("k", solver.kff(), ref.k), ("K", solver.Kfb(), ref.K)):
    err = np.abs(got - want) / (1.0 + np.abs(want))
    print(name, "max rel err %.3e"%err.max(), "median per-instance max %.3e"%np.median(err.reshape(B,-1).max(1)))
tr = solver.trace()
print(tr[0,:mi+1,:])
