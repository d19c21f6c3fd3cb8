import sys, os
sys.path.insert(0, ".")
import numpy as np, nmpc_amd
from nmpc_amd import workloads
wl = workloads.cartpole_batch(B=4096, T=100, seed=1234)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
s.config().print_level = 0; s.config().max_iter = 8
for _ in range(3):
    s.solve(wl.t0, wl.x0, wl.u_init)
q = s.qpFreeMask().astype(np.float64) * 16.0
print("kernel ms", s.computationDuration().opt, s.kernelName())
bt, bw, ft, fw = q[0, 0], q[0, 1], q[0, 2], q[0, 3]
print(f"master: backward total {bt:.0f} cyc, linearise {bw:.0f} ({bw / max(bt, 1):.2%}) | forward total {ft:.0f}, wait {fw:.0f} ({fw / max(ft, 1):.2%})")
qi = s.qpFreeMask()
n_sec, n_bw, n_fw = int(qi[0, 9]), int(qi[0, 10]), int(qi[0, 11])
print(f"        {n_bw} backward passes ({bt / max(n_bw, 1):.0f} cyc each), {n_fw} forward passes ({ft / max(n_fw, 1):.0f} each), "
      f"{n_sec} stamped sections ({bw / max(n_sec, 1):.0f} cyc each: the linearisation of one chunk)")
for w in range(0, 4):
    hw = int(s.qpFreeMask()[0, 4 + w])
    print(f"   wave {w}: HW_ID wave slot {hw & 15}, SIMD {(hw >> 4) & 3}, CU {(hw >> 8) & 15}, SE {(hw >> 13) & 7}")
