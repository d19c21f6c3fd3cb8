"""Which configuration / field differs under the wave-timing fuzz build: the cases of the GPU suite that failed under
scripts/fuzz_suite.sh, field by field.

    python scripts/fuzz_probe.py worker [--reps N]     one JSON line {case: {field: [digest per repetition]}} for NMPC_HIP_DDP_LIB
    python scripts/fuzz_probe.py [seeds...]            runs the worker on the product library and on each fuzz library and prints
                                                       the (case, field) pairs whose digests are not all equal"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
FIELDS = ("X", "U", "cost", "kff", "Kfb", "trace", "iters", "status", "dV")
FORCED = dict(k_rel_norm_thre=0.0, cost_update_thre=-1e300)


def worker(reps):
    import nmpc_amd
    from nmpc_amd import workloads as W

    def make(wl, **cfg):
        s = nmpc_amd.DDPSolverBatch(nmpc_amd.make_problem(wl.model), wl.B)
        c = s.config()
        c.print_level, c.horizon_steps = 0, wl.T
        for k, v in cfg.items():
            setattr(c, k, v)
        if wl.limits is not None and cfg.get("with_input_constraint"):
            s.setInputLimits(*wl.limits)
        return s

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]

    out = {}

    def run(label, wl, env=None, **cfg):
        for k in ("NMPC_HIP_DDP_FAN_SCRATCH", "NMPC_HIP_DDP_KERNEL"):
            os.environ.pop(k, None)
        os.environ.update(env or {})
        d = {f: [] for f in FIELDS}
        for r in range(reps):
            s = make(wl, **cfg)
            s.solve(wl.t0, wl.x0, wl.u_init)
            for f in FIELDS:
                d[f].append(sha(getattr(s, f)()))
            if r == reps - 1:  # the handle again
                s.solve(wl.t0, wl.x0, wl.u_init)
                for f in FIELDS:
                    d[f].append(sha(getattr(s, f)()))
        out[label] = d
        print(label, file=sys.stderr, flush=True)

    wl = W.cartpole_batch(B=512, T=100, seed=3)
    cfgs = {"conv500": dict(max_iter=500), "forced50": dict(max_iter=50, **FORCED),
            "forced40x25": dict(max_iter=40, alpha_list=np.power(10.0, np.linspace(0, -3, 25)), **FORCED)}
    for cname, cfg in cfgs.items():
        for mname, fan, env in (("seq", 2, None), ("fan", 1, None), ("fan-noscratch", 1, {"NMPC_HIP_DDP_FAN_SCRATCH": "0"})):
            for rag in (-1,) if cname != "conv500" else (-1, 1):
                run(f"fanout/{cname}/{mname}/ragged{rag}", wl, env, line_search_fan_out=fan, ragged_schedule=rag, **cfg)
    for B, it, con in ((520, 90, True), (17, 40, False), (300, 70, False)):
        wl = W.cartpole_batch(B=B, T=100, seed=B + it, constrained=con)
        for rag in (-1, 0, 1):
            run(f"ragged/{B}-{it}-{con}/ragged{rag}", wl, None, max_iter=it, with_input_constraint=con, ragged_schedule=rag)
    print(json.dumps(out), flush=True)


def main(seeds):
    from nmpc_amd import build as b
    libs = [("product", None)] + [(f"fuzz{s}", b.build_fuzz(int(s))) for s in seeds]
    res = {}
    for name, lib in libs:
        env = dict(os.environ)
        env.pop("NMPC_HIP_DDP_LIB", None)
        if lib:
            env["NMPC_HIP_DDP_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    ref = res["product"]
    bad = 0
    for case in ref:
        for f in FIELDS:
            want = set(ref[case][f])
            line = []
            for name, _ in libs:
                got = res[name][case][f]
                if set(got) != want or len(want) != 1:
                    line.append(f"{name}: {sum(g not in want or len(want) != 1 for g in got)}/{len(got)}")
            if line:
                bad += 1
                print(f"{case:44s} {f:7s} differs  " + "  ".join(line))
    # equivalences the suite asserts within one library: seq == fan == fan-noscratch, ragged == whole
    for name, _ in libs:
        for case in ref:
            group, cfg, *rest = case.split("/")
            first = next(c for c in ref if c.startswith(group + "/" + cfg + "/"))
            for f in FIELDS:
                if set(res[name][case][f]) != set(res[name][first][f]):
                    print(f"[{name}] {case} != {first} in {f}")
    print(f"{bad} (case, field) pairs differ between libraries")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker(int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3)
    else:
        main(sys.argv[1:] or ["1", "2", "3"])
