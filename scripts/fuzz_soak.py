"""Digest soak for the wave-timing fuzz build (include/nmpc_amd/hip/fuzz_sched.hpp).

    python scripts/fuzz_soak.py [--reps N] [--cases small|full] [--only substring]

Runs every kernel family (six DDP families + the FMPC kernels) on the library NMPC_HIP_DDP_LIB names (default: the product
library), each case `reps` times, and prints ONE JSON line {case: {"kernel": ..., "digests": [sha256 per repetition]}}.  The caller
(tests/test_gpu_fuzz_sched.py, scripts/fuzz_experiment.sh) runs it once per library and compares: the fuzz build perturbs wave timing
only, so every digest of every repetition must equal the product build's.  A race between waves shows as a digest that differs."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nmpc_amd  # noqa: E402
from nmpc_amd import workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--cases", choices=("small", "full"), default="small")
ap.add_argument("--only", default="")
args = ap.parse_args()

FORCED = dict(k_rel_norm_thre=0.0, cost_update_thre=-1e300)  # forced iterations: every line search back-tracks through the list
W = workloads
big = args.cases == "full"
# (label, workload, configuration, NMPC_HIP_DDP_KERNEL or None)
CASES = [
    ("quad c2", lambda: W.cartpole_batch(B=4096 if big else 1024, T=100, seed=1234), dict(max_iter=8), None),
    ("quad c2 forced", lambda: W.cartpole_batch(B=4096 if big else 512, T=100, seed=3), dict(max_iter=12, **FORCED), None),
    ("quad c2 box", lambda: W.cartpole_batch(B=4096 if big else 512, T=100, seed=1234, constrained=True), dict(max_iter=6, with_input_constraint=True), None),
    # the step-size-parallel line search (automatic only for long solves: the cases above never run it) with wide passes, accepted
    # rollouts adopted from the fan-out scratch, step sizes taken from the cost-only waves; then without the scratch; then the
    # sequential search of the same kernel family — and the resumable launches of the ragged schedule
    ("quad c2 fan-out forced", lambda: W.cartpole_batch(B=4096 if big else 512, T=100, seed=3), dict(max_iter=50, line_search_fan_out=1, ragged_schedule=-1, **FORCED), None),
    ("quad c2 fan-out 25 step sizes", lambda: W.cartpole_batch(B=4096 if big else 512, T=100, seed=3),
     dict(max_iter=40, line_search_fan_out=1, ragged_schedule=-1, alpha_list=np.power(10.0, np.linspace(0, -3, 25)), **FORCED), None),
    ("quad c2 fan-out no scratch", lambda: W.cartpole_batch(B=512, T=100, seed=3), dict(max_iter=50, line_search_fan_out=1, ragged_schedule=-1, **FORCED), "env:NMPC_HIP_DDP_FAN_SCRATCH=0"),
    ("quad c2 sequential forced", lambda: W.cartpole_batch(B=512, T=100, seed=3), dict(max_iter=50, line_search_fan_out=2, ragged_schedule=-1, **FORCED), None),
    ("quad c2 to convergence", lambda: W.cartpole_batch(B=4096 if big else 1000, T=100, seed=1500), dict(max_iter=500, ragged_schedule=-1), None),
    ("quad c2 ragged schedule", lambda: W.cartpole_batch(B=4096 if big else 1000, T=100, seed=1500), dict(max_iter=500, ragged_schedule=1), None),
    ("quad c2 box ragged schedule", lambda: W.cartpole_batch(B=520, T=100, seed=610, constrained=True), dict(max_iter=90, with_input_constraint=True, ragged_schedule=1), None),
    ("quad bipedal ragged schedule", lambda: W.bipedal_batch(B=260, T=300, seed=5), dict(max_iter=60, ragged_schedule=1), None),
    ("two-wave ragged schedule", lambda: W.cartpole_batch(B=1500, T=100, seed=77), dict(max_iter=100, ragged_schedule=1), "2w"),
    ("quad bipedal", lambda: W.bipedal_batch(B=1024 if big else 256, T=300, seed=7), dict(max_iter=4), None),
    ("two-wave", lambda: W.cartpole_batch(B=8192 if big else 1024, T=100, seed=99), dict(max_iter=6), "2w"),
    ("two-wave box", lambda: W.cartpole_batch(B=8192 if big else 512, T=100, seed=98, constrained=True), dict(max_iter=5, with_input_constraint=True), "2w"),
    ("lane 1w", lambda: W.cartpole_batch(B=512, T=100, seed=17), dict(max_iter=6), "1w"),
    ("lane vertical", lambda: W.vertical_batch(B=256, T=300, seed=5), dict(max_iter=4, with_input_constraint=True), None),
    ("tile32 c4", lambda: W.quadrotor_batch(B=8192 if big else 1024, T=50, seed=5, fp32=True), dict(max_iter=4, cost_update_thre=1e-3), "tile32"),
    ("tile32 c4 forced", lambda: W.quadrotor_batch(B=8192 if big else 1024, T=50, seed=6, fp32=True), dict(max_iter=8, **FORCED), "tile32"),
    ("tile32 c4 box", lambda: W.quadrotor_batch(B=2048 if big else 512, T=50, seed=5, fp32=True, constrained=True),
     dict(max_iter=3, cost_update_thre=1e-3, with_input_constraint=True), "tile32"),
    ("tile64f c4", lambda: W.quadrotor_batch(B=8192 if big else 1024, T=50, seed=5, fp32=True), dict(max_iter=6, cost_update_thre=1e-3), "tile64"),
    ("tile64f c4 forced", lambda: W.quadrotor_batch(B=8192 if big else 1500, T=50, seed=8, fp32=True), dict(max_iter=12, **FORCED), "tile64"),
    ("tile64 c5", lambda: W.manipulator_batch(B=8192 if big else 1024, T=30, seed=5), dict(max_iter=5), "tile64"),
    ("tile64 c5 forced ragged", lambda: W.manipulator_batch(B=8200 if big else 1100, T=30, seed=9), dict(max_iter=12, **FORCED), "tile64"),
    ("tile64 quadrotor forced", lambda: W.quadrotor_batch(B=8192 if big else 1024, T=50, seed=11), dict(max_iter=12, **FORCED), "tile64"),
    ("tile64 quadrotor box", lambda: W.quadrotor_batch(B=4096 if big else 512, T=50, seed=5, constrained=True), dict(max_iter=3, with_input_constraint=True), "tile64"),
    ("tile64 centroidal", lambda: W.centroidal_batch(B=1024 if big else 300, T=100, seed=5), dict(max_iter=6), None),
    ("tile64 vtol", lambda: W.planar_vtol_batch(B=4096 if big else 700, T=60, seed=5), dict(max_iter=6), None),
    ("wpi manipulator", lambda: W.manipulator_batch(B=512 if big else 128, T=30, seed=5), dict(max_iter=4), "wpi"),
]


def make(wl, cfg):
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config()
    c.print_level = 0
    c.horizon_steps = wl.T
    for k, v in cfg.items():
        setattr(c, k, v)
    if wl.limits is not None and cfg.get("with_input_constraint"):
        s.setInputLimits(*wl.limits)
    return s


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:24]


out = {}
for label, mk, cfg, kernel in CASES:
    if args.only and args.only not in label:
        continue
    os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
    os.environ.pop("NMPC_HIP_DDP_FAN_SCRATCH", None)
    if kernel and kernel.startswith("env:"):  # (the knobs are read when a handle is created)
        k, v = kernel[4:].split("=")
        os.environ[k] = v
    elif kernel:
        os.environ["NMPC_HIP_DDP_KERNEL"] = kernel
    wl = mk()
    s = make(wl, cfg)
    digests = []
    for r in range(args.reps):
        h = s if r % 2 == 0 else make(wl, cfg)  # alternately the reused handle and a fresh one
        h.solve(wl.t0, wl.x0, wl.u_init)
        digests.append(sha(h.X(), h.U(), h.cost(), h.iters(), h.status(), h.kff(), h.trace()))
    out[label] = {"kernel": s.kernelName(), "digests": digests}
    print(f"{label:28s} {s.kernelName():28s} {len(set(digests))} distinct digest(s) in {args.reps} repetitions", file=sys.stderr, flush=True)
os.environ.pop("NMPC_HIP_DDP_KERNEL", None)
os.environ.pop("NMPC_HIP_DDP_FAN_SCRATCH", None)

# streamed solves (nmpc_hip_ddp_solve_stream): a queue through the slots of one handle — resumable launches numbered per instance, the
# rollout-only launches of refilled slots, extraction / compaction / refill kernels between them
if not args.only or "stream" in args.only:
    for label, kern, N, S, cfg in (("quad c2 stream", "quad", 3000 if big else 1500, 512, dict(max_iter=120, trace_level=0)),
                                   ("quad c2 box stream", "quad", 1500, 256, dict(max_iter=60, trace_level=0, with_input_constraint=True)),
                                   ("two-wave stream", "2w", 1200, 256, dict(max_iter=80, trace_level=0))):
        wl = W.cartpole_batch(B=N, T=100, seed=31, constrained=bool(cfg.get("with_input_constraint")))
        digests = []
        for r in range(args.reps):
            s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), S)
            c = s.config()
            c.print_level, c.horizon_steps = 0, wl.T
            for k, v in cfg.items():
                setattr(c, k, v)
            if wl.limits is not None and cfg.get("with_input_constraint"):
                s.setInputLimits(*wl.limits)
            s.setKernel(kern)
            res = s.solveStream(wl.t0, wl.x0, wl.u_init, span=(8, 16, 5)[r % 3])  # (the round length does not show in the results)
            digests.append(sha(res.X, res.U, res.cost, res.iters, res.status, res.trace_last))
        out[label] = {"kernel": s.kernelName(), "digests": digests}
        print(f"{label:28s} {s.kernelName():28s} {len(set(digests))} distinct digest(s) in {args.reps} repetitions", file=sys.stderr, flush=True)

if not args.only or "fmpc" in args.only:
    from nmpc_amd import fmpc as F
    for label, cls, B, T, it, ric in (("fmpc cartpole fused", F.FmpcProblemCartPole, 1024 if big else 300, 200 if big else 60, 4, "fused"),
                                      ("fmpc cartpole quad", F.FmpcProblemCartPole, 1024 if big else 300, 200 if big else 60, 4, "quad"),
                                      ("fmpc cartpole lane", F.FmpcProblemCartPole, 512, 60, 3, "lane"),
                                      ("fmpc pointmass", F.FmpcProblemPointMass, 300, 40, 4, None)):
        os.environ.pop("NMPC_HIP_FMPC_RICCATI", None)
        if ric:
            os.environ["NMPC_HIP_FMPC_RICCATI"] = ric
        prob = cls()
        rng = np.random.default_rng(B + T)
        n, m, g = prob.state_dim, prob.input_dim, prob.ineq_dim
        x0 = 0.2 * rng.standard_normal((B, n))
        digests = []
        s = F.FmpcSolverBatch(prob, B, T)
        s.config().max_iter = it
        for r in range(args.reps):
            var = F.Variable.make(prob, T, B)
            var.reset(0.0, 0.0, 0.0, 1.0, 1.0)
            s.solve(np.zeros(B), x0, var)
            v = s.variable()
            digests.append(sha(*v.arrays(), s.iters(), s.status(), s.traceDataList()))
        out[label] = {"kernel": ",".join(s.kernelNames()), "digests": digests}
        print(f"{label:28s} {len(set(digests))} distinct digest(s) in {args.reps} repetitions", file=sys.stderr, flush=True)
    os.environ.pop("NMPC_HIP_FMPC_RICCATI", None)

print(json.dumps(out), flush=True)
