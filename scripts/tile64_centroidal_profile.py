"""Phase split of the tile kernel on centroidal motion; with a -DNMPC_AMD_PROFILE_TILE64 build (NMPC_HIP_DDP_LIB) also the role stamps.
    python scripts/tile64_centroidal_profile.py [B] [T] [max_iter]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 8
wl = workloads.centroidal_batch(B=B, T=T, seed=1234)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
c = s.config()
c.print_level = 0
c.horizon_steps = wl.T
c.max_iter = mi
for _ in range(3):
    s.solve(wl.t0, wl.x0, wl.u_init)
d = s.computationDuration()
tr = s.trace()
its = int(s.iters().sum())
print(f"centroidal B {B} T {T} max_iter {mi}: {s.kernelName()} kernel {d.opt:.3f} ms backward {d.backward:.3f} forward {d.forward:.3f}; "
      f"{its} instance-iterations, backward passes / iteration {tr[:, 1:, 10].sum() / max(its, 1):.2f}, forward trials / iteration "
      f"{tr[:, 1:, 11].sum() / max(its, 1):.2f}")
if os.environ.get("NMPC_HIP_DDP_LIB"):
    _m = s.qpFreeMask()
    q = np.concatenate([_m[0, :wl.T], _m[1, :wl.T]])[:40].astype(np.float64) * 16.0
    sweeps, passes, steps = q[9], q[10], q[4]
    print(f"  workgroup 0: {sweeps:.0f} sweeps, {passes:.0f} rollout passes, {steps:.0f} backward steps on matrix wave 1; ticks per step "
          f"{q[2] / max(steps, 1):.0f}")
    print(f"  per sweep timestep: model wave linearisation {q[0] / max(sweeps * T, 1):.0f} waiting {q[1] / max(sweeps * T, 1):.0f} | matrix wave 1 "
          f"steps {q[2] / max(sweeps * T, 1):.0f} waiting {q[3] / max(sweeps * T, 1):.0f}")
    print(f"  record stride {int(q[37] / 16.0)} doubles, group size {int(q[38] / 16.0)}, LDS doubles in front of the records {int(q[39] / 16.0)} of 20480")
    print(f"  per rollout timestep: rolling lanes compute {q[5] / max(passes * (T + 2), 1):.0f} waiting {q[6] / max(passes * (T + 2), 1):.0f} | "
          f"prefetching wave 1: prefetch {q[7] / max(passes * (T + 2), 1):.0f} waiting {q[8] / max(passes * (T + 2), 1):.0f}")
