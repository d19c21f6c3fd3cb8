"""A/B of the natural-layout gains of the fp64 tile kernel (kBig: the centroidal problem): the wave's slots factorised TOGETHER, a column
per lane (gainsPrepare / gainsBatch / gainsComplete, round 6) against every slot over the whole wave (stepGainsNatural; a library built
with -DNMPC_AMD_AB_NATURAL_GAINS, or the round-5 library).  One process per library (NMPC_HIP_DDP_LIB); prints a digest of every output
and the kernel time per case — the arithmetic is the same operation for operation, so the digests are expected to agree.
    python scripts/centroidal_gains_ab.py [lib_a lib_b ...]      (no arguments: worker mode on the library in NMPC_HIP_DDP_LIB)"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

CASES = [  # B, T, seed, max_iter, group cap (0: default), extra config
    (4096, 100, 1234, 8, 0, {}),
    (256, 100, 1234, 8, 0, {}),
    (1024, 100, 7, 8, 0, {}),
    (64, 100, 3, 4, 32, {}),
    (96, 60, 5, 6, 32, {"reg_type": 2}),
    (8960, 40, 11, 3, 0, {}),     # 35 instances per group: a fifth slot on some waves (whole-wave factorisation)
    (65, 7, 2, 4, 32, {}),
    (128, 100, 9, 60, 0, {}),     # to convergence: regularisation retries, failed pivots
]


def worker():
    import numpy as np
    import nmpc_amd
    from nmpc_amd import workloads
    out = []
    for B, T, seed, mi, cap, cfg in CASES:
        if cap:
            os.environ["NMPC_HIP_DDP_TILE64_GROUP"] = str(cap)
        else:
            os.environ.pop("NMPC_HIP_DDP_TILE64_GROUP", None)
        wl = workloads.centroidal_batch(B=B, T=T, seed=seed)
        s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
        s.setKernel("tile64")
        c = s.config(); c.print_level, c.horizon_steps, c.max_iter = 0, wl.T, mi
        for k, v in cfg.items():
            setattr(c, k, v)
        ms, bw, fw = [], [], []
        for _ in range(3):
            s.solve(wl.t0, wl.x0, wl.u_init)
            d = s.computationDuration()
            ms.append(d.opt); bw.append(d.backward); fw.append(d.forward)
        h = hashlib.sha256()
        for f in (s.X(), s.U(), s.cost(), s.kff(), s.Kfb(), s.iters(), s.status(), s.trace()):
            h.update(np.ascontiguousarray(f).tobytes())
        out.append({"case": [B, T, seed, mi, cap, cfg], "kernel": s.kernelName(), "ms": min(ms), "bw": min(bw), "fw": min(fw), "digest": h.hexdigest()[:16],
                    "iters": int(s.iters().sum()), "failed": int((s.status() < 0).sum())})
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) < 2:
        worker()
        sys.exit(0)
    res = {}
    for lib in sys.argv[1:]:
        env = dict(os.environ, NMPC_HIP_DDP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True, env=env, cwd=ROOT)
        if r.returncode != 0:
            print(lib, "FAILED", r.stderr[-2000:])
            continue
        res[lib] = json.loads(r.stdout.strip().splitlines()[-1])
    libs = list(res)
    for k in range(len(CASES)):
        row = [res[l][k] for l in libs]
        same = len({x["digest"] for x in row}) == 1
        print(f"case {row[0]['case']}: " + "  |  ".join(f"{x['ms']:8.3f} ms (bw {x['bw']:.2f} fw {x['fw']:.2f}) it {x['iters']} fail {x['failed']} {x['digest']}" for x in row)
              + ("   SAME BITS" if same else "   DIFFERENT"), flush=True)
