import sys
sys.path.insert(0, ".")
import numpy as np, nmpc_amd
from nmpc_amd import workloads
wl = workloads.cartpole_batch(B=4096, T=100, seed=1234)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
c = s.config(); c.print_level = 0; c.max_iter = 8; c.with_input_constraint = False
for _ in range(3):
    s.solve(wl.t0, wl.x0, wl.u_init)
q = s.qpFreeMask().astype(np.float64) * 16.0
qi = s.qpFreeMask()
print("kernel ms", s.computationDuration().opt, s.kernelName())
bt, bw, ft, fw = q[0, 0], q[0, 1], q[0, 2], q[0, 3]
n_sec, n_bw, n_fw = int(qi[0, 9]), int(qi[0, 10]), int(qi[0, 11])
print(f"backward total {bt:.0f} cyc in {n_bw} passes ({bt/max(n_bw,1):.0f} each), linearise {bw:.0f} | forward total {ft:.0f} in {n_fw} passes ({ft/max(n_fw,1):.0f} each), wait {fw:.0f}")
