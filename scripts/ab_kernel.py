"""A/B of library builds on ONE GPU box: the same workload on each library in turn (a fresh process per library and repetition,
interleaved), kernel time by HIP events.   python scripts/ab_kernel.py <c2|m1|c2box|c3|c4|c4f64|c5|centroidal> <lib|main> [<lib> ...]"""
import json, os, subprocess, sys

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, nmpc_amd
from nmpc_amd import workloads as W
name = sys.argv[1]
forced = dict(k_rel_norm_thre=0.0, cost_update_thre=-1e300)
wl, cfg = {"c2": (lambda: W.cartpole_batch(B=4096, T=100, seed=1234), dict(max_iter=8)),
           "m1": (lambda: W.cartpole_batch(B=4096, T=100, seed=1234), dict(max_iter=50, **forced)),
           "c2box": (lambda: W.cartpole_batch(B=4096, T=100, seed=1234, constrained=True), dict(max_iter=8, with_input_constraint=True)),
           "c3": (lambda: W.bipedal_batch(B=1024, T=300, seed=1234), dict(max_iter=8)),
           "c4": (lambda: W.quadrotor_batch(B=8192, T=50, seed=1234, fp32=True), dict(max_iter=8, cost_update_thre=1e-3)),
           "c4f64": (lambda: W.quadrotor_batch(B=8192, T=50, seed=1234), dict(max_iter=8)),
           "c5": (lambda: W.manipulator_batch(B=8192, T=30, seed=1234), dict(max_iter=8)),
           "centroidal": (lambda: W.centroidal_batch(B=4096, T=100, seed=1234), dict(max_iter=8))}[name]
wl = wl()
s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
c = s.config(); c.print_level = 0; c.horizon_steps = wl.T
for k, v in cfg.items(): setattr(c, k, v)
if wl.limits is not None and cfg.get("with_input_constraint"): s.setInputLimits(*wl.limits)
ms = []
for _ in range(int(sys.argv[2])):
    s.solve(wl.t0, wl.x0, wl.u_init); ms.append(s.computationDuration().opt)
import hashlib
h = hashlib.sha256(); [h.update(np.ascontiguousarray(a).tobytes()) for a in (s.X(), s.U(), s.iters(), s.status())]
print(json.dumps({"min": min(ms[2:]), "median": float(np.median(ms[2:])), "kernel": s.kernelName(), "digest": h.hexdigest()[:12], "iters": int(s.iters().sum())}))
'''
name, libs = sys.argv[1], sys.argv[2:]
res = {l: [] for l in libs}
for rep in range(3):
    for l in libs:
        env = dict(os.environ)
        env.pop("NMPC_HIP_DDP_LIB", None)
        if l != "main":
            env["NMPC_HIP_DDP_LIB"] = os.path.abspath(l)
        r = subprocess.run([sys.executable, "-c", WORKER, name, "22"], capture_output=True, text=True, env=env)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", r.stderr[-500:])
            continue
        res[l].append(json.loads(line[-1]))
for l in libs:
    if res[l]:
        print(f"{name:10s} {os.path.basename(os.path.dirname(l)) if l != 'main' else 'main':24s} kernel ms min {min(d['min'] for d in res[l]):.4f}  medians "
              + " ".join(f"{d['median']:.4f}" for d in res[l]) + f"  {res[l][0]['kernel']} digest {res[l][0]['digest']} iters {res[l][0]['iters']}")
