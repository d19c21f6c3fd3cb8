import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmpc_amd, oracle
from nmpc_amd import workloads
wl = workloads.vertical_batch(B=128, T=300, seed=1234, constrained=True)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.initial_lambda = 1e-6; c.with_input_constraint = True; c.max_iter = 60
s.setInputLimits(*wl.limits)
s.solve(wl.t0, wl.x0, wl.u_init)
ocfg = oracle.default_config(horizon_steps=wl.T, initial_lambda=1e-6, with_input_constraint=1, max_iter=60)
ref = oracle.solve_batch(wl.model, ocfg, wl.x0, wl.u_init, t0=wl.t0, lower=wl.limits[0], upper=wl.limits[1], n_threads=8, want_alpha_hist=True)
print("status eq", np.array_equal(s.status(), ref.status), "iters eq", np.array_equal(s.iters(), ref.iters))
bad = np.nonzero((s.iters() != ref.iters) | (s.status() != ref.status))[0]
print("bad", bad[:20], "gpu iters", s.iters()[bad[:10]], "cpu", ref.iters[bad[:10]], "status", s.status()[bad[:10]], ref.status[bad[:10]])
err = np.abs(s.X() - ref.X).reshape(wl.B, -1).max(1); print("max dX per inst (top)", np.sort(err)[-5:], np.argsort(err)[-5:])
erru = np.abs(s.U() - ref.U).reshape(wl.B, -1).max(1); print("max dU per inst (top)", np.sort(erru)[-5:], np.argsort(erru)[-5:])
if len(bad):
    b = bad[0]
    tr = s.trace()[b]; r = oracle.solve(wl.model, ocfg, wl.x0[b], wl.u_init[b], t0=wl.t0[b], lower=wl.limits[0], upper=wl.limits[1])
    n = min(r.trace.shape[0], int(s.iters()[b]) + 1)
    for i in range(n):
        if not np.allclose(tr[i], r.trace[i], rtol=1e-6, atol=1e-12):
            print("first differing trace row", i, "\n gpu", tr[i], "\n cpu", r.trace[i]); break
    print("qp ret eq", np.array_equal(s.qpRetval()[b], r.qp_retval), "free eq", np.array_equal(s.qpFreeMask()[b], r.qp_free_mask))
    d = np.nonzero(s.qpRetval()[b] != r.qp_retval)[0]; print("qp diff steps", d[:10], s.qpRetval()[b][d[:10]], r.qp_retval[d[:10]])
# full-size status check
wl = workloads.cartpole_batch(B=4096, T=100, seed=1234)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B); s.config().print_level = 0
s.solve(wl.t0, wl.x0, wl.u_init)
ocfg = oracle.default_config()
ref = oracle.solve_batch(wl.model, ocfg, wl.x0, wl.u_init, n_threads=16)
print("C2 4096: status counts gpu", np.unique(s.status(), return_counts=True), "cpu", np.unique(ref.status, return_counts=True))
print("status eq", np.array_equal(s.status(), ref.status), "iters eq", np.array_equal(s.iters(), ref.iters), "n diff", int((s.iters() != ref.iters).sum()))
bad = np.nonzero(s.iters() != ref.iters)[0]; print(bad[:10], s.iters()[bad[:10]], ref.iters[bad[:10]])
print("max dX", np.abs(s.X() - ref.X).max(), "max dU", np.abs(s.U() - ref.U).max(), "iters max", ref.iters.max(), "kernel ms", s.computationDuration().opt)
