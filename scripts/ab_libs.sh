#!/bin/bash
# A/B of library builds on the same GPU box: usage scripts/ab_libs.sh <model> <T> <libA|main> <libB> ...
MODEL=$1; T=$2; shift 2
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "main" ]; then unset NMPC_HIP_DDP_LIB; else export NMPC_HIP_DDP_LIB=$L; fi
  python - "$MODEL" "$T" "$L" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, nmpc_amd
from nmpc_amd import workloads
model, T, lib = sys.argv[1], int(sys.argv[2]), sys.argv[3]
wl = workloads.quadrotor_batch(B=8192, T=T, seed=1234) if model == "quadrotor" else workloads.manipulator_batch(B=8192, T=T, seed=1234)
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = 8
ms = []; bw = []; fw = []
for _ in range(8):
    s.solve(wl.t0, wl.x0, wl.u_init); d = s.computationDuration(); ms.append(d.opt); bw.append(d.backward); fw.append(d.forward)
print(f"{lib:34s} {model}: kernel ms min {min(ms):.3f} median {np.median(ms):.3f}  backward {np.median(bw):.3f} forward {np.median(fw):.3f}  iterations {int(s.iters().sum())}")
PY
done; done
