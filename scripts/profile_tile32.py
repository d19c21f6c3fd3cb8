"""Per-phase shader-clock ticks of a -DNMPC_AMD_PROFILE_TILE32 build of the fp32 tile kernel (model wave of workgroup 0).
   build:  scripts/build_alt.sh prof32 -DNMPC_AMD_PROFILE_TILE32   run: NMPC_HIP_DDP_LIB=nmpc_amd/lib/alt/prof32.so python scripts/profile_tile32.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmpc_amd
from nmpc_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = workloads.quadrotor_batch(B=B, T=50, seed=1234, fp32=True)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = mi
for _ in range(2):
    s.solve(wl.t0, wl.x0, wl.u_init)
qq = s.qpFreeMask()[0, :24].astype(np.float64)
q = qq[:8]
q[:4] *= 16.0
q[7] *= 16.0
print(f"kernel {s.computationDuration().opt:.3f} ms, {s.kernelName()}, B {B}, max_iter {mi}, instance-iterations {int(s.iters().sum())}")
names = ("initial rollout", "backward sweeps", "line search (all step sizes)", "re-rolls")
rounds = max(q[6], 1)
for n, v in zip(names, q[:4]):
    print(f"   {n:30s} {v:12.0f} ticks  ({v / rounds:10.0f} per iteration round)")
print(f"   of the sweeps: linearisation by the model wave {q[7]:12.0f} ticks ({q[7] / rounds:10.0f} per iteration round)")
print("   matrix waves' own work inside the sweeps (both solves), by wave: " + " ".join(f"{int(v * 16 / 1000)}k" for v in qq[9:24]))
print(f"   sweeps {int(q[4])}, re-rolls {int(q[5])}, iteration rounds {int(q[6])}   (ticks of __builtin_readcyclecounter)")
