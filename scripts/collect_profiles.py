#!/usr/bin/env python3
"""Turn one scripts/profile_round.sh session (gpurun_out/profile_<tag>/) into the tracked summaries under profiles/.

  profiles/<tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of `python bench.py --steps 10 --warmup 2`
  profiles/<tag>_pmc_summary.txt         per-launch means of the PMC passes (each set collected in its own run)
  profiles/<tag>_counter_calibration.txt FETCH_SIZE / WRITE_SIZE on a known 1 GiB read / write at 8 B per lane
  profiles/<tag>_ubench.txt              per-instruction prices (clock, fp64 FMA / div / sqrt / sin / sincos; FMA latency
                                         vs issue interval for a lone wavefront; recipFast accuracy)
  profiles/<tag>_roles.txt               master / helper cycles and barrier-wait shares (profiling build of the 2-wave kernel)
  profiles/<tag>_bench.json              the bench line of the same session (unprofiled run)
  profiles/hbm_traffic.json              corrected HBM bytes per launch, read by bench.py for roofline.traffic
"""
import csv, collections, glob, json, os, shutil, subprocess, sys

tag, src = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "ubench_clock.txt"), os.path.join(dst, f"{tag}_ubench.txt"))
if os.path.exists(os.path.join(src, "roles.txt")):  # per-role cycle counters of the -DNMPC_AMD_PROFILE_2W build
    shutil.copy(os.path.join(src, "roles.txt"), os.path.join(dst, f"{tag}_roles.txt"))
for extra in ("batch_scaling.txt", "ubench_mfma4.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, f"{tag}_{extra}"))

def means(pattern_file, kernel_pat):
    out = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(src, pattern_file))):
        for r in csv.DictReader(open(f)):
            if kernel_pat in r["Kernel_Name"]:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}

pm, cnt = means("pmc?_counter_collection.csv", "ddp_solve")
cal_r, _ = means("calF_counter_collection.csv", "read_k")
cal_w, _ = means("calW_counter_collection.csv", "write_k")
GiB_KB = 1024.0 * 1024.0
fetch_scale = GiB_KB / cal_r["FETCH_SIZE"]     # actual KB per reported KB on a known 1 GiB read
write_scale = GiB_KB / cal_w["WRITE_SIZE"]
with open(os.path.join(dst, f"{tag}_counter_calibration.txt"), "w") as f:
    f.write("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on scripts/ubench_hbm_counters (1 GiB streamed, 8 B per lane,\n"
            "the access width of the DDP kernels), separate runs per counter:\n")
    f.write(f"  read_k : FETCH_SIZE = {cal_r['FETCH_SIZE']:.1f} KB reported for 1048576 KB read  -> scale x{fetch_scale:.4f}\n")
    f.write(f"  write_k: WRITE_SIZE = {cal_w['WRITE_SIZE']:.1f} KB reported for 1048576 KB written -> scale x{write_scale:.4f}\n")
    f.write("(FETCH_SIZE under-reports a coalesced read by 2x on gfx950, as MI355X_MICROARCH.md §HBM states for 16 B/lane;\n"
            " the same factor holds at 8 B/lane.  WRITE_SIZE is 1:1.)\n")
bench_line = [l for l in open(os.path.join(src, "bench.txt")) if l.startswith("{")][-1]
bench = json.loads(bench_line)
json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
hbm_bytes = (pm["FETCH_SIZE"] * fetch_scale + pm["WRITE_SIZE"] * write_scale) * 1024.0
with open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w") as f:
    f.write(f"kernel: {bench['roofline']['kernel']} (bench.py --steps 10 --warmup 2, batch 4096, 8 iterations per launch)\n")
    f.write("per-launch means; SQ_* cycle counters are in quad-cycles (x4 = shader cycles), summed over all waves of the launch\n")
    for k in sorted(pm):
        f.write(f"  {k:28s} n={cnt[k]:3d} mean={pm[k]:18.1f}\n")
    wc = pm.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        f.write("\nderived:\n")
        f.write(f"  VALU-active share of wave cycles   {pm['SQ_ACTIVE_INST_VALU'] / wc:6.3f}\n")
        f.write(f"  s_waitcnt (memory) share           {pm['SQ_WAIT_ANY'] / wc:6.3f}\n")
        f.write(f"  issue-stall share                  {pm['SQ_WAIT_INST_ANY'] / wc:6.3f}\n")
        f.write(f"  VALU instructions per wave         {pm['SQ_INSTS_VALU'] / max(pm.get('SQ_WAVES', 64.0), 1.0):12.0f}\n")
        f.write(f"  fp64 FMA+MUL+ADD per wave          {(pm['SQ_INSTS_VALU_FMA_F64'] + pm['SQ_INSTS_VALU_MUL_F64'] + pm['SQ_INSTS_VALU_ADD_F64']) / max(pm.get('SQ_WAVES', 64.0), 1.0):12.0f}\n")
        f.write(f"  MFMA f64 instructions              {pm.get('SQ_INSTS_VALU_MFMA_F64', 0):12.0f}\n")
    f.write(f"\nHBM-side traffic per launch (corrected): fetch {pm['FETCH_SIZE'] * fetch_scale * 1024 / 1e6:.1f} MB"
            f" + write {pm['WRITE_SIZE'] * write_scale * 1024 / 1e6:.1f} MB = {hbm_bytes / 1e6:.1f} MB\n")
    f.write(f"L2 hit rate TCC_HIT/(HIT+MISS) = {pm['TCC_HIT'] / (pm['TCC_HIT'] + pm['TCC_MISS']):.3f}\n")
json.dump({"batch": bench["config"].get("batch", 4096) if isinstance(bench.get("config"), dict) else 4096,
           "iterations_per_step": bench["config"]["iterations_per_step"],
           "hbm_bytes_per_launch": hbm_bytes,
           "source": f"profiles/{tag}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes; "
                     f"FETCH_SIZE x{fetch_scale:.2f} per profiles/{tag}_counter_calibration.txt)"},
          open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_pmc_summary.txt")).read())
