"""Where two builds of the library part ways on one case: the first differing iteration of every instance and the trace columns that
differ there.

    python scripts/fuzz_diff.py worker <out.npz>      solves the cases on NMPC_HIP_DDP_LIB and saves the arrays
    python scripts/fuzz_diff.py [seeds...]            product against each fuzz library (and each library against itself, twice)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
FORCED = dict(k_rel_norm_thre=0.0, cost_update_thre=-1e300)
COLS = ("iter", "cost", "lambda", "dlambda", "alpha", "k_rel", "upd_act", "upd_exp", "upd_ratio", "alpha_idx", "n_bw", "n_fw")


def worker(path):
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

    out = {}

    def run(label, wl, env=None, **cfg):
        for k in ("NMPC_HIP_DDP_FAN_SCRATCH", "NMPC_HIP_DDP_KERNEL"):
            os.environ.pop(k, None)
        os.environ.update(env or {})
        for r in range(2):
            s = make(wl, **cfg)
            s.solve(wl.t0, wl.x0, wl.u_init)
            for f in ("trace", "cost", "iters", "status", "X"):
                out[f"{label}|{r}|{f}"] = np.array(getattr(s, f)())

    wl = W.cartpole_batch(B=512, T=100, seed=3)
    for mname, fan, env in (("seq", 2, None), ("fan", 1, None), ("fan-noscratch", 1, {"NMPC_HIP_DDP_FAN_SCRATCH": "0"})):
        run(f"forced50/{mname}", wl, env, line_search_fan_out=fan, ragged_schedule=-1, max_iter=50, **FORCED)
    wl = W.cartpole_batch(B=520, T=100, seed=610, constrained=True)
    run("box520", wl, None, max_iter=90, with_input_constraint=True, ragged_schedule=-1)
    wl = W.cartpole_batch(B=300, T=100, seed=370)
    run("conv300", wl, None, max_iter=70, ragged_schedule=-1)
    np.savez(path, **out)


def main(seeds):
    from nmpc_amd import build as b
    # (a seed, or the name of a variant library under nmpc_amd/lib/ built beforehand — e.g. fuzz1_reopen_fan_adopt)
    libs = [("product", None)] + [(f"fuzz{s}", b.build_fuzz(int(s))) if s.isdigit() else (s, b.variant_path(s)) for s in seeds]
    res = {}
    tmp = tempfile.mkdtemp()
    for name, lib in libs:
        env = dict(os.environ)
        env.pop("NMPC_HIP_DDP_LIB", None)
        if lib:
            env["NMPC_HIP_DDP_LIB"] = lib
        path = os.path.join(tmp, name + ".npz")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", path], capture_output=True, text=True, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[name] = dict(np.load(path))
    labels = sorted({k.rsplit("|", 2)[0] for k in res["product"]})
    for lab in labels:
        for name, _ in libs:
            a = {k.split("|")[0] + "|" + k.split("|")[2]: v for k, v in res[name].items() if k.split("|")[1] == "0"}
            bb = {k.split("|")[0] + "|" + k.split("|")[2]: v for k, v in res[name].items() if k.split("|")[1] == "1"}
            cmp2(a, bb, lab, f"{name} run 0 vs run 1")
        p = {k.split("|")[0] + "|" + k.split("|")[2]: v for k, v in res["product"].items() if k.split("|")[1] == "0"}
        for name, _ in libs[1:]:
            a = {k.split("|")[0] + "|" + k.split("|")[2]: v for k, v in res[name].items() if k.split("|")[1] == "0"}
            cmp2(p, a, lab, f"product vs {name}")
    # within the product library: the modes the suite compares
    p = {k.split("|")[0] + "|" + k.split("|")[2]: v for k, v in res["product"].items() if k.split("|")[1] == "0"}
    for name, _ in libs:
        a = {k.split("|")[0] + "|" + k.split("|")[2]: v for k, v in res[name].items() if k.split("|")[1] == "0"}
        for m in ("fan", "fan-noscratch"):
            cmp2({k.replace("forced50/seq", "x"): v for k, v in a.items()}, {k.replace(f"forced50/{m}", "x"): v for k, v in a.items()}, "x", f"[{name}] seq vs {m}")


def cmp2(a, b, lab, what):
    ta, tb = a[f"{lab}|trace"], b[f"{lab}|trace"]
    diff = (ta != tb) & ~(np.isnan(ta) & np.isnan(tb))
    inst = np.flatnonzero(diff.any(axis=(1, 2)))
    line = f"{what:26s} {lab:24s}: {inst.size:4d} of {ta.shape[0]} instances differ in the trace"
    for f in ("cost", "iters", "status", "X"):
        x, y = a[f"{lab}|{f}"], b[f"{lab}|{f}"]
        line += f", {int((x != y).reshape(x.shape[0], -1).any(axis=1).sum())} in {f}"
    print(line)
    firsts = {}
    for i in inst:
        it = int(np.flatnonzero(diff[i].any(axis=1))[0])
        cols = tuple(COLS[c] for c in np.flatnonzero(diff[i, it]))
        firsts.setdefault(cols, []).append((int(i), it))
    for cols, lst in sorted(firsts.items(), key=lambda kv: -len(kv[1]))[:5]:
        i, it = lst[0]
        c = [COLS.index(x) for x in cols]
        print(f"      first difference in {cols}: {len(lst)} instances; instance {i} (workgroup {i // 16}, slot {i % 16}) iteration {it}: "
              + ", ".join(f"{COLS[k]} {ta[i, it, k]!r} vs {tb[i, it, k]!r}" for k in c[:4])
              + f" | first-difference iterations {min(x[1] for x in lst)}..{max(x[1] for x in lst)}; workgroups {sorted({x[0] // 16 for x in lst})[:12]}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker(sys.argv[2])
    else:
        main(sys.argv[1:] or ["1", "2"])
