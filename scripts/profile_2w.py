"""Read the per-role barrier-wait counters of a -DNMPC_AMD_PROFILE_2W build (see ddp_kernels_2w.hpp)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmpc_amd
from nmpc_amd import workloads
wl = workloads.cartpole_batch(B=4096, T=100, seed=1234)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
s.config().print_level = 0; s.config().max_iter = 8
for _ in range(3):
    s.solve(wl.t0, wl.x0, wl.u_init)
q = s.qpFreeMask().astype(np.float64) * 16.0
print("kernel ms", s.computationDuration().opt)
for role, inst in (("master", 0), ("helper", 1)):
    bt, bw, ft, fw = q[inst, 0], q[inst, 1], q[inst, 2], q[inst, 3]
    print(f"{role}: backward total {bt:.0f} cyc, barrier wait {bw:.0f} ({bw / max(bt, 1):.2%}) | forward total {ft:.0f}, wait {fw:.0f} ({fw / max(ft, 1):.2%})")
    hw = int(s.qpFreeMask()[inst, 4])
    print(f"   HW_ID wave slot {hw & 15}, SIMD {(hw >> 4) & 3}, CU {(hw >> 8) & 15}, SE {(hw >> 13) & 7}")
