import numpy as np, sys
sys.path.insert(0,'.')
import nmpc_amd, oracle
from nmpc_amd import workloads
wl = workloads.quadrotor_batch(B=384, T=50, seed=11, fp32=True)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
c = s.config(); c.print_level=0; c.horizon_steps=50; c.max_iter=60; c.cost_update_thre=1e-4
s.solve(wl.t0, wl.x0, wl.u_init)
tr = s.trace()
cfg = oracle.default_config(horizon_steps=50, max_iter=60, cost_update_thre=1e-4)
for b in (76, 206):
    r = oracle.solve("quadrotor_f32", cfg, wl.x0[b], wl.u_init[b])
    r64 = oracle.solve("quadrotor", cfg, wl.x0[b], wl.u_init[b])
    for it in range(0, 6):
        g = tr[b, it]; o = r.trace[it] if it < len(r.trace) else np.zeros(12); d = r64.trace[it] if it < len(r64.trace) else np.zeros(12)
        print("b %d it %d | gpu cost %.8g act %.4e exp %.4e ai %d | f32 oracle cost %.8g act %.4e exp %.4e ai %d | f64 oracle cost %.10g act %.4e exp %.4e ai %d" % (b, it, g[1], g[6], g[7], g[9], o[1], o[6], o[7], o[9], d[1], d[6], d[7], d[9]))
