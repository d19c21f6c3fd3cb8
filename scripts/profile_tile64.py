"""Per-role shader-clock ticks of a -DNMPC_AMD_PROFILE_TILE64 build of the fp64 tile kernel (workgroup 0).
   build:  scripts/build_alt.sh prof64 model_manipulator.hip -DNMPC_AMD_PROFILE_TILE64
   run:    NMPC_HIP_DDP_LIB=nmpc_amd/lib/alt/prof64.so python scripts/profile_tile64.py manipulator 30 [B] [max_iter]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "manipulator"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
mi = int(sys.argv[4]) if len(sys.argv) > 4 else 8
f32 = len(sys.argv) > 5 and sys.argv[5] == "f32"  # (quadrotor_f32: the kernel's float instantiation)
wl = workloads.quadrotor_batch(B=B, T=T, seed=1234, fp32=f32) if model == "quadrotor" else workloads.manipulator_batch(B=B, T=T, seed=1234)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
c = s.config()
c.print_level = 0
c.horizon_steps = wl.T
c.max_iter = mi
if f32:
    c.cost_update_thre = 1e-3
for _ in range(2):
    s.solve(wl.t0, wl.x0, wl.u_init)
_m = s.qpFreeMask()
q = np.concatenate([_m[0, :wl.T], _m[1, :wl.T]])[:40].astype(np.float64)
tick = q.copy()
for k in (0, 1, 2, 3, 5, 6, 7, 8, 11):
    tick[k] *= 16.0
steps_shift = 16.0  # counters 4, 9, 10 were shifted too
sweeps, passes, steps = q[9] * steps_shift, q[10] * steps_shift, q[4] * steps_shift
d = s.computationDuration()
print(f"{model} B {B} T {T} max_iter {mi}: kernel {d.opt:.3f} ms ({s.kernelName()}), backward {d.backward:.3f} forward {d.forward:.3f}; "
      f"workgroup 0: ~{sweeps:.0f} sweeps, ~{passes:.0f} rollout passes, ~{steps:.0f} backward steps on matrix wave 1")
print(f"  per sweep timestep:   model wave linearisation {tick[0] / max(sweeps * T, 1):8.0f}  waiting {tick[1] / max(sweeps * T, 1):8.0f}   | "
      f"matrix wave 1 steps {tick[2] / max(sweeps * T, 1):8.0f} waiting {tick[3] / max(sweeps * T, 1):8.0f}  ({tick[2] / max(steps, 1):.0f} ticks per "
      f"instance-step) | matrix wave 5 steps {tick[11] / max(sweeps * T, 1):8.0f}")
print(f"  per rollout timestep: rolling lanes compute {tick[5] / max(passes * (T + 2), 1):8.0f} waiting {tick[6] / max(passes * (T + 2), 1):8.0f}   | "
      f"prefetching wave 1: prefetch {tick[7] / max(passes * (T + 2), 1):8.0f} waiting {tick[8] / max(passes * (T + 2), 1):8.0f}")
print("  matrix waves 1 .. 7, ticks over the solve (M): steps " + " ".join(f"{q[16 + w] * 16.0 / 1e6:6.2f}" for w in range(1, 8))
      + "   waiting at the sweeps' barriers " + " ".join(f"{q[24 + w] * 16.0 / 1e6:6.2f}" for w in range(1, 8)))
print(f"  record stride {int(q[37])} doubles, group size {int(q[38])}, LDS doubles in front of the records {int(q[39])} of 20480")
