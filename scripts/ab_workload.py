"""A/B of library builds on one GPU box for a bench workload: python scripts/ab_workload.py <c2|c2m2|c4|c4f64|c5|centroidal|c3> <lib|main> ...
(one process per library and repetition, interleaved; kernel ms by HIP events, backward / forward split)."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
W = {"c2": ("cartpole_batch", {}, 4096, 100, None), "c2m2": ("cartpole_batch", {}, 4096, 100, "m2"), "c4": ("quadrotor_batch", dict(fp32=True), 8192, 50, 1e-3), "c4f64": ("quadrotor_batch", {}, 8192, 50, None), "c5": ("manipulator_batch", {}, 8192, 30, None),
     "centroidal": ("centroidal_batch", {}, 4096, 100, None), "c3": ("bipedal_batch", {}, 1024, 300, None)}
if sys.argv[1] == "--worker":
    sys.path.insert(0, ROOT)
    import numpy as np, nmpc_amd
    from nmpc_amd import workloads
    gen, kw, B, T, thre = W[sys.argv[2]]
    wl = getattr(workloads, gen)(B=B, T=T, seed=1234, **kw)
    s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
    c = s.config(); c.print_level, c.horizon_steps, c.max_iter = 0, wl.T, 8
    if thre == "m2":  # the reference's default Configuration: to convergence, max_iter 500
        c.max_iter = 500
    elif thre is not None:
        c.cost_update_thre = thre
    ms, bw, fw = [], [], []
    for _ in range(10):
        s.solve(wl.t0, wl.x0, wl.u_init); d = s.computationDuration(); ms.append(d.opt); bw.append(d.backward); fw.append(d.forward)
    print(f"{sys.argv[3]:34s} {sys.argv[2]}: {s.kernelName()} kernel ms min {min(ms):.3f} median {np.median(ms):.3f}  backward {np.median(bw):.3f} forward {np.median(fw):.3f}  "
          f"iterations {int(s.iters().sum())} -> {s.iters().sum() / wl.B / (np.median(ms) * 1e-3):.0f} batch-it/s (kernel)")
else:
    for rep in range(2):
        for lib in sys.argv[2:]:
            env = dict(os.environ)
            env.pop("NMPC_HIP_DDP_LIB", None)
            if lib != "main":
                env["NMPC_HIP_DDP_LIB"] = os.path.abspath(lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", sys.argv[1], lib], env=env, capture_output=True, text=True, cwd=ROOT)
            print((r.stdout.strip().splitlines() or [r.stderr[-500:]])[-1], flush=True)
