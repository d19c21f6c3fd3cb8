"""Soak test: a long closed-loop run (device-resident MPC, cart-pole with the +-15 N box: the constrained quad kernel with its
step-size fan-out) repeated with the two-wave kernel and repeated with itself.  Run-to-run: bit-identical logs.  Kernel to
kernel: the closed loop amplifies rounding differences (a swing-up is a sensitive trajectory), so the comparison is per tick
on the iteration counts / statuses of the first ticks and on where the loops end up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmpc_amd

B, T, TICKS = 4096, 200, 1500
rng = np.random.default_rng(1)
x0 = np.tile(np.array([0.0, np.pi, 0.0, 0.0]), (B, 1)); x0[:, :2] += rng.uniform(-0.2, 0.2, (B, 2))
u = np.zeros((B, T, 1))
def run(kernel):
    if kernel: os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
    else: os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemCartPole(running_u=[0.01]), B)
    c = s.config(); c.print_level = 0; c.horizon_steps = T; c.max_iter = 3; c.with_input_constraint = True
    s.setInputLimits(np.array([-15.0]), np.array([15.0]))
    log = s.mpcRun(0.0, x0, u, TICKS, shift_warm_start=False, sim_substeps=2, sim_dt=0.002)
    return s.kernelName(), log
n1, a = run("")
n2, a2 = run("")
n3, b = run("2w")
same = all(np.array_equal(getattr(a, f), getattr(a2, f)) for f in ("x", "u0", "iters", "status"))
print(f"{n1}: {B} loops x {TICKS} ticks, repeated run bit-identical: {same}")
first = 50
print(f"{n1} vs {n3}: iteration counts equal on the first {first} ticks: {np.array_equal(a.iters[:, :first], b.iters[:, :first])}, "
      f"statuses equal: {np.array_equal(a.status[:, :first], b.status[:, :first])}, max |dx| there {np.abs(a.x[:, :first] - b.x[:, :first]).max():.2e}")
up_a = np.abs(np.mod(a.x_final[:, 1] + np.pi, 2 * np.pi) - np.pi) < 0.1
up_b = np.abs(np.mod(b.x_final[:, 1] + np.pi, 2 * np.pi) - np.pi) < 0.1
print(f"upright at the end: {int(up_a.sum())} / {B} ({n1}), {int(up_b.sum())} / {B} ({n3}); same set: {np.array_equal(up_a, up_b)}; "
      f"max |u0| {np.abs(a.u0).max():.3f} (box 15)")
