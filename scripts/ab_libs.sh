for rep in 1 2; do
for L in "" nmpc_amd/lib/alt/pair.so; do
  export NMPC_HIP_DDP_LIB=$L; [ -z "$L" ] && unset NMPC_HIP_DDP_LIB
  echo "== lib ${L:-main}"
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, nmpc_amd
from nmpc_amd import workloads
for model, T in (("quadrotor", 50),):
    wl = workloads.quadrotor_batch(B=8192, T=T, seed=1234)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config(); c.print_level = 0; c.horizon_steps = wl.T; c.max_iter = 8
    ms = []; bw = []; fw = []
    for _ in range(8):
        s.solve(wl.t0, wl.x0, wl.u_init); d = s.computationDuration(); ms.append(d.opt); bw.append(d.backward); fw.append(d.forward)
    print(f"{model}: kernel ms min {min(ms):.3f} median {np.median(ms):.3f}  backward {np.median(bw):.3f} forward {np.median(fw):.3f}")
PY
done; done
