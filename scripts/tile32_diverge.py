"""Where do the fp32 tile kernel and the fp32 oracle part ways?  (diagnostic, quadrotor_f32)"""
import numpy as np, sys
import nmpc_amd, oracle
from nmpc_amd import workloads
B=int(sys.argv[1]) if len(sys.argv)>1 else 8192
mi=int(sys.argv[2]) if len(sys.argv)>2 else 8
wl = workloads.quadrotor_batch(B=B, T=50, seed=1, fp32=True)
solver = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B, device=0)
solver.config().print_level = 0
solver.config().horizon_steps = wl.T
solver.config().max_iter = mi
solver.solve(wl.t0, wl.x0, wl.u_init)
tr = solver.trace()
cfg = oracle.default_config(horizon_steps=wl.T, max_iter=mi)
ref = oracle.solve_batch(wl.model, cfg, wl.x0, wl.u_init, t0=wl.t0, n_threads=8, want_alpha_hist=True)
errX = (np.abs(solver.X()-ref.X)/(1+np.abs(ref.X))).reshape(B,-1).max(1)
worst = np.argsort(-errX)[:4]
print("instances with X err > 1e-3:", (errX>1e-3).sum(), " > 1e-2:", (errX>1e-2).sum(), "status counts gpu", np.bincount(solver.status()+1, minlength=3), "ref", np.bincount(ref.status+1, minlength=3))
for b in worst:
    r1 = oracle.solve(wl.model, cfg, wl.x0[b], wl.u_init[b])
    print("instance", b, "errX", errX[b], "gpu status/iters", solver.status()[b], solver.iters()[b], "ref", r1.status, r1.iters)
    n = max(solver.iters()[b], r1.iters)+1
    for it in range(n):
        g = tr[b,it]; o = r1.trace[it] if it < len(r1.trace) else np.zeros(12)
        print("  it %d gpu cost %.7g lam %.3g a_idx %d act %.3e exp %.3e krel %.3e | ref cost %.7g lam %.3g a_idx %d act %.3e exp %.3e krel %.3e" % (it, g[1], g[2], g[9], g[6], g[7], g[5], o[1], o[2], o[9], o[6], o[7], o[5]))
