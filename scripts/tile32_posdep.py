import numpy as np, sys
sys.path.insert(0,'.')
import nmpc_amd
from nmpc_amd import workloads
def solve(wl, lo, hi, mi):
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), hi-lo)
    c = s.config(); c.print_level=0; c.horizon_steps=wl.T; c.max_iter=mi
    s.solve(wl.t0[lo:hi], wl.x0[lo:hi], wl.u_init[lo:hi])
    return s.X().copy(), s.kff().copy(), s.trace().copy()
for mi in (0, 1, 3):
    wl = workloads.quadrotor_batch(B=75, T=20, seed=5, fp32=True)
    Xf, kf, trf = solve(wl, 0, 75, mi)
    for lo, hi in ((0, 38), (38, 75), (32, 64), (1, 75)):
        Xs, ks, trs = solve(wl, lo, hi, mi)
        dX = np.abs(Xs - Xf[lo:hi]).reshape(hi-lo, -1).max(1)
        dk = np.abs(ks - kf[lo:hi]).reshape(hi-lo, -1).max(1)
        bad = np.flatnonzero((dX > 0) | (dk > 0))
        print("max_iter", mi, "shard", lo, hi, "differing instances (global idx):", (bad + lo).tolist()[:40], "max dX %.3g max dk %.3g" % (dX.max(), dk.max()))
