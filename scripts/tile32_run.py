"""Workload for profiler runs of the fp32 tile kernel: `reps` solves of the C4 batch (quadrotor_f32, B 8192, T 50)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmpc_amd
from nmpc_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
wl = workloads.quadrotor_batch(B=B, T=50, seed=1234, fp32=True)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = mi
for _ in range(reps):
    s.solve(wl.t0, wl.x0, wl.u_init)
print(f"kernel {s.computationDuration().opt:.3f} ms  {s.kernelName()}  B {B} max_iter {mi}  instance-iterations {int(s.iters().sum())}")
