"""The device-resident receding-horizon loop, N times on fresh handles (after an fp32 loop in the same process): every log
bit-identical to the first.   python scripts/mpc_determinism_soak.py [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
wf = workloads.quadrotor_batch(B=40, T=50, seed=5, fp32=True)
sf = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wf.model), wf.B)
sf.config().print_level = 0; sf.config().horizon_steps = wf.T; sf.config().max_iter = 3; sf.config().cost_update_thre = 1e-3
sf.mpcRun(0.0, wf.x0, np.zeros_like(wf.u_init), 4, shift_warm_start=True)
B, T, ticks = 64, 300, 200
rng = np.random.default_rng(21)
x0 = np.stack([rng.uniform(-0.02, 0.02, B), rng.uniform(-0.05, 0.05, B)], 1)
x0[0] = 0.0
first, bad = None, 0
for r in range(N):
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.DDPProblemBipedal(), B)
    s.config().print_level = 0
    s.config().horizon_steps = T
    log = s.mpcRun(0.0, x0, np.zeros((B, T, 1)), ticks, shift_warm_start=True)
    d = (log.x.tobytes(), log.u0.tobytes(), log.iters.tobytes(), log.x_final.tobytes(), log.t_final.tobytes())
    if first is None:
        first = d
        assert np.array_equal(log.x[:, 0], x0), "the first tick does not start from x0"
    elif d != first:
        bad += 1
        print("run", r, "differs: instances whose first logged state is not x0:", np.flatnonzero((log.x[:, 0] != x0).any(axis=1))[:8])
print(f"bipedal shift loop, {B} x T {T} x {ticks} ticks, {N} runs on fresh handles: {bad} differ from the first")
