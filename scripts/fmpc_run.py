"""Times the batched FMPC solve (cart-pole problem of the reference's TestFmpcCartPole) and prints the mean solve time; under
rocprofv3 --kernel-trace --stats this gives the per-kernel split.  usage: fmpc_run.py [B] [T] [max_iter] [reps] [model]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmpc_amd import fmpc as F  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 5
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
model = sys.argv[5] if len(sys.argv) > 5 else "fmpc_cartpole"
prob = {"fmpc_cartpole": F.FmpcProblemCartPole, "fmpc_oscillator": F.FmpcProblemOscillator,
        "fmpc_pointmass": F.FmpcProblemPointMass}[model](0.01)
rng = np.random.default_rng(0)
n = prob.state_dim
x0 = np.zeros((B, n))
x0[:, 0] = rng.uniform(-1, 1, B)
x0[:, 1] = rng.uniform(-0.3, 0.3, B) + (np.pi if os.environ.get("SWINGUP") else 0.0)
s = F.FmpcSolverBatch(prob, B, T)
s.config().max_iter = max_iter
s.config().use_graph = not os.environ.get("NOGRAPH")
var = F.Variable.make(prob, T, B)
var.reset(0.0, 0.0, 0.0, 1.0, 1.0)
ms = []
for r in range(reps):
    s.solve(0.0, x0, var)
    ms.append(s.computationDuration().solve)
it = s.iters()
print(f"B={B} T={T} max_iter={max_iter} model={model}: solve {np.mean(ms[1:]):.3f} ms (first {ms[0]:.3f}), iterations mean {it.mean():.2f}, "
      f"status {np.bincount(s.status())}, {it.sum() / B / (np.mean(ms[1:]) * 1e-3):.1f} batch-iterations/s")
if os.environ.get("NMPC_AMD_EXTRA_HIPCC_FLAGS", "").find("NMPC_AMD_FMPC_PROFILE") >= 0:
    m = s.meritFunc()
    print(f"riccati kernel, 100 MHz ticks: backward {m[:, 0].mean() / 100:.1f} us, forward {m[:, 1].mean() / 100:.1f} us")
if os.environ.get("FMPC_TAIL_PROFILE"):
    m = s.meritFunc()
    print("tail kernel (last but one iteration), us: reductions %.1f, walk of slice 0 %.1f, wait for the slowest slice %.1f"
          % (m[:, 0].mean() / 100, m[:, 1].mean() / 100, m[:, 2].mean() / 100))
    tot = m.sum(axis=1) / 100
    print("  per workgroup (sum of the three): min %.1f  median %.1f  max %.1f;  walk min %.1f max %.1f;  reductions min %.1f max %.1f"
          % (tot.min(), np.median(tot), tot.max(), m[:, 1].min() / 100, m[:, 1].max() / 100, m[:, 0].min() / 100, m[:, 0].max() / 100))
if os.environ.get("FMPC_PROFILE2"):
    tr = s.traceDataList()
    m = s.meritFunc()
    print("backward per launch, 100 MHz ticks -> us: request %.1f, compute %.1f, barrier+commit %.1f, flush+barrier %.1f"
          % (tr[:, 0, 2].mean() / 100, m[:, 0].mean() / 100, m[:, 1].mean() / 100, m[:, 2].mean() / 100))
