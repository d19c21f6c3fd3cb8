"""DESIGN.md section 5's table from a default bench line (profiles/<tag>_bench_default.json: headline + secondary legs with the PMC traffic of
profiles/hbm_traffic.json).   python scripts/design_table.py profiles/r06_bench_default.json"""
import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]


def row(name, kernel, val, r, extra=""):
    tr = r.get("traffic")
    fused = r.get("fused_lower_bound_bytes_per_launch")
    over = f"{tr / fused:.2f}" if tr and fused else "—"
    hbm = f"{r.get('hbm_frac_measured'):.2f}" if r.get("hbm_frac_measured") else "—"
    return f"| {name} | {kernel} | {val}{extra} | {r['kernel_ms_avg']:.3f} | {r['frac']:.2f} | {over} | {hbm} |"


print("| workload | kernel | it/s (batch-iterations) | kernel ms | contract frac | measured traffic / fused bound | HBM frac measured |")
print("|---|---|---|---|---|---|---|")
st = c.get("m2_stream", {})
extra = (f" (M1 {c['m1_value'] / 1e3:.1f} k, CPU {c['m1'].get('cpu_value', 0):.0f}; M2 {c['m2_value']:.0f}, CPU {c['m2'].get('cpu_value', 0):.0f}; M2 pooled ×8 "
         f"**{c['m2_overlapped_value'] / 1e3:.2f} k**; M2 streamed {c.get('m2_stream_value', 0) / 1e3:.2f} k at 32 768 instances, "
         f"**{c.get('m2_stream_sustained_value', 0) / 1e3:.1f} k** at 262 144)")
print(row("**c2** cart-pole 4096 × T 100 fp64 (BASELINE metric)", "quad", f"**{d['value'] / 1e3:.1f} k**", d["roofline"], extra))
names = {"c3": ("c3 bipedal 1024 × T 300", "quad"), "c4": ("c4 quadrotor 8192 × T 50 fp32, thre 1e-3", "tile64<float>, twelve waves"),
         "c4f64": ("c4f64 quadrotor fp64", "tile64"), "c5": ("c5 manipulator 8192 × T 30", "tile64"), "centroidal": ("centroidal 4096 × T 100", "tile64, batched gains"),
         "fmpc": ("fmpc cart-pole 4096 × T 200 × 5", "fused Riccati, delta, tail (three launches per iteration)")}
for k, (nm, kern) in names.items():
    v = d["secondary"][k]
    r = dict(v["roofline"])
    r.setdefault("kernel_ms_avg", 0.0)
    ex = ""
    if v.get("pooled"):
        ex = f" (pooled ×{v['pooled']['handles']} {v['pooled']['value'] / 1e3:.1f} k)"
    val = f"{v['value'] / 1e3:.2f} k" if v["value"] >= 1000 else f"{v['value']:.0f}"
    print(row(nm, kern, val, r, ex))
print()
print("cpu_baseline:", d["cpu_baseline"]["value"], "on", d["cpu_baseline"]["cores"], "threads")
