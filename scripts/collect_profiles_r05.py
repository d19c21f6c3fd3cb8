#!/usr/bin/env python3
"""Turn one scripts/profile_r05.sh session (gpurun_out/profile_<tag>/) into the tracked summaries under profiles/:

  profiles/<tag>_kernel_stats_<wl>.csv     rocprofv3 --kernel-trace --stats of `python bench.py --workload <wl> --steps 10 --warmup 2`
  profiles/<tag>_pmc_summary_<wl>.txt      per-launch means of the PMC passes (each set collected in its own run)
  profiles/<tag>_counter_calibration.txt   FETCH_SIZE / WRITE_SIZE on a known 1 GiB read / write
  profiles/<tag>_bench_<wl>.json           unprofiled bench line of the same session
  profiles/hbm_traffic.json                corrected HBM bytes per launch PER WORKLOAD, read by bench.py for roofline.traffic
"""
import csv, collections, glob, json, os, shutil, sys

tag, src = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
sys.path.insert(0, root)
from nmpc_amd import build as hip_build  # noqa: E402

# the hash of the device sources the session ran on (scripts/profile_r05.sh writes it next to its outputs): bench.py refuses
# the traffic entries on any other sources
_hash_file = os.path.join(src, "source_hash.txt")
SOURCE_HASH = open(_hash_file).read().strip() if os.path.exists(_hash_file) else hip_build.source_hash()


def means(pattern, kernel_pat):
    out = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(src, pattern))):
        for r in csv.DictReader(open(f)):
            if kernel_pat in r["Kernel_Name"]:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


cal_r, _ = means("calF_counter_collection.csv", "read_k")
cal_w, _ = means("calW_counter_collection.csv", "write_k")
GiB_KB = 1024.0 * 1024.0
fetch_scale = GiB_KB / cal_r["FETCH_SIZE"]
write_scale = GiB_KB / cal_w["WRITE_SIZE"]
with open(os.path.join(dst, f"{tag}_counter_calibration.txt"), "w") as f:
    f.write("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on scripts/ubench_hbm_counters (1 GiB streamed), separate runs per counter:\n")
    f.write(f"  read_k : FETCH_SIZE = {cal_r['FETCH_SIZE']:.1f} KB reported for 1048576 KB read  -> scale x{fetch_scale:.4f}\n")
    f.write(f"  write_k: WRITE_SIZE = {cal_w['WRITE_SIZE']:.1f} KB reported for 1048576 KB written -> scale x{write_scale:.4f}\n")
    f.write("(FETCH_SIZE under-reports a coalesced read by 2x on gfx950, as MI355X_MICROARCH.md §HBM states; WRITE_SIZE is 1:1.)\n")

traffic = {}
for wl in ("c2", "c4", "c3", "c5", "c4f64", "centroidal"):
    stats = os.path.join(src, f"stats_{wl}_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats_{wl}.csv"))
    bench_file = os.path.join(src, f"bench_{wl}.txt")
    bench = None
    if os.path.exists(bench_file):
        lines = [l for l in open(bench_file) if l.startswith("{")]
        if lines:
            bench = json.loads(lines[-1])
            json.dump(bench, open(os.path.join(dst, f"{tag}_bench_{wl}.json"), "w"), indent=1)
    pm, cnt = means(f"pmc?_{wl}_counter_collection.csv", "ddp_solve")
    if not pm:
        continue
    hbm_bytes = (pm.get("FETCH_SIZE", 0.0) * fetch_scale + pm.get("WRITE_SIZE", 0.0) * write_scale) * 1024.0
    with open(os.path.join(dst, f"{tag}_pmc_summary_{wl}.txt"), "w") as f:
        kname = bench["roofline"]["kernel"] if bench else "ddp_solve_*"
        f.write(f"kernel: {kname} (bench.py --workload {wl} --steps 10 --warmup 2; one launch = one solve of the batch, max_iter 8)\n")
        f.write("per-launch means; SQ_* cycle counters are in quad-cycles (x4 = shader cycles), summed over all waves of the launch\n")
        for k in sorted(pm):
            f.write(f"  {k:30s} n={cnt[k]:3d} mean={pm[k]:18.1f}\n")
        f.write("\nderived:\n")
        f.write(f"  HBM traffic per launch (FETCH x{fetch_scale:.2f} + WRITE x{write_scale:.2f})  {hbm_bytes / 1e6:10.1f} MB\n")
        if "TCC_HIT" in pm:
            f.write(f"  L2 hit rate                        {pm['TCC_HIT'] / (pm['TCC_HIT'] + pm['TCC_MISS']):6.3f}\n")
        wc = pm.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            f.write(f"  VALU-active share of wave cycles   {pm['SQ_ACTIVE_INST_VALU'] / wc:6.3f}\n")
            f.write(f"  s_waitcnt / barrier share          {pm['SQ_WAIT_ANY'] / wc:6.3f}\n")
            f.write(f"  issue-stall share                  {pm['SQ_WAIT_INST_ANY'] / wc:6.3f}\n")
            f.write(f"  VALU instructions per wave         {pm['SQ_INSTS_VALU'] / pm['SQ_WAVES']:10.0f}\n")
            if pm.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                # busy cycles are per SIMD-pipe; 1024 SIMDs; wave cycles / waves = kernel duration in quad-cycles
                dur_cycles = 4.0 * wc / pm["SQ_WAVES"]
                f.write(f"  matrix-core busy share (1024 SIMDs) {pm['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / dur_cycles:6.3f}\n")
                f.write(f"  VALU issue share (1024 SIMDs)       {4.0 * pm['SQ_ACTIVE_INST_VALU'] / 1024.0 / dur_cycles:6.3f}\n")
        if bench:
            rf = bench["roofline"]
            f.write(f"  fused lower bound per launch       {rf['fused_lower_bound_bytes_per_launch'] / 1e6:10.1f} MB  -> traffic / bound = "
                    f"{hbm_bytes / rf['fused_lower_bound_bytes_per_launch']:.2f}\n")
            f.write(f"  contract (staged) bytes per launch {rf['algorithmic_bytes_per_launch'] / 1e6:10.1f} MB\n")
    if bench:
        traffic[wl] = {"hbm_bytes_per_launch": hbm_bytes, "batch": int(bench["metric"].split("batch=")[1].split(",")[0]),
                       "iterations_per_step": bench["config"]["iterations_per_step"],
                       "source_hash": SOURCE_HASH, "cost_update_thre": bench["config"].get("cost_update_thre"),
                       "source": f"profiles/{tag}_pmc_summary_{wl}.txt (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes; FETCH_SIZE "
                                 f"x{fetch_scale:.2f} per profiles/{tag}_counter_calibration.txt)"}
# FMPC: several kernels per iteration -> one table, per kernel: launches, HBM bytes per launch, SQ shares
stats = os.path.join(src, "stats_fmpc_kernel_stats.csv")
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats_fmpc.csv"))
    bench = None
    bench_file = os.path.join(src, "bench_fmpc.txt")
    if os.path.exists(bench_file):
        lines = [l for l in open(bench_file) if l.startswith("{")]
        if lines:
            bench = json.loads(lines[-1])
            json.dump(bench, open(os.path.join(dst, f"{tag}_bench_fmpc.json"), "w"), indent=1)
    avg_us = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(stats))}
    with open(os.path.join(dst, f"{tag}_pmc_summary_fmpc.txt"), "w") as f:
        f.write("bench.py --workload fmpc --steps 10 --warmup 2 (4096 cart-pole FMPC instances, T = 200, max_iter 5); per-launch means,\n")
        f.write("each counter set collected in its own run; SQ_* cycle counters in quad-cycles summed over the waves of a launch\n\n")
        f.write("%-28s %9s %11s %11s %8s %9s %9s %9s\n" % ("kernel", "avg_us", "fetch_MB", "write_MB", "L2_hit", "GB/s", "VALU_act", "waitcnt"))
        for short in ("fmpc_barrier_kernel", "fmpc_coeff_kernel", "fmpc_riccati_fused_kernel", "fmpc_riccati_quad_kernel", "fmpc_riccati_kernel", "fmpc_delta_kernel",
                      "fmpc_step_length_kernel", "fmpc_update_kernel", "fmpc_tail_kernel", "fmpc_transpose_kernel"):
            pm, cnt = means("pmc?_fmpc_counter_collection.csv", short)
            if not pm:
                continue
            fetch = pm.get("FETCH_SIZE", 0.0) * fetch_scale * 1024.0
            write = pm.get("WRITE_SIZE", 0.0) * write_scale * 1024.0
            us = next((v for k, v in avg_us.items() if short in k), float("nan"))
            hit = pm["TCC_HIT"] / (pm["TCC_HIT"] + pm["TCC_MISS"]) if pm.get("TCC_HIT") else float("nan")
            wc = pm.get("SQ_WAVE_CYCLES", 0.0)
            f.write("%-28s %9.1f %11.1f %11.1f %8.3f %9.0f %9.3f %9.3f\n" % (
                short, us, fetch / 1e6, write / 1e6, hit, (fetch + write) / (us * 1e-6) / 1e9,
                pm["SQ_ACTIVE_INST_VALU"] / wc if wc else float("nan"), pm["SQ_WAIT_ANY"] / wc if wc else float("nan")))
            if short in ("fmpc_riccati_fused_kernel", "fmpc_riccati_quad_kernel", "fmpc_riccati_kernel") and bench and short == bench["roofline"]["kernel"]:
                traffic["fmpc"] = {"hbm_bytes_per_launch": fetch + write, "batch": int(bench["metric"].split("batch=")[1].split(",")[0]),
                                   "horizon": int(bench["metric"].split("T=")[1]), "source_hash": SOURCE_HASH,
                                   "source": f"profiles/{tag}_pmc_summary_fmpc.txt ({short}; FETCH_SIZE x{fetch_scale:.2f}, "
                                             "WRITE_SIZE 1:1, separate passes)"}
        if bench:
            f.write("\nalgorithmic bytes per launch of the Riccati kernel (bench accounting): %.1f MB\n"
                    % (bench["roofline"]["algorithmic_bytes_per_launch"] / 1e6))
json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
# the bench lines of this session were printed before this file existed in its new state: give them the session's own traffic
for wl, entry in traffic.items():
    path = os.path.join(dst, f"{tag}_bench_{wl}.json")
    if not os.path.exists(path):
        continue
    b = json.load(open(path))
    rf = b["roofline"]
    rf["traffic"] = entry["hbm_bytes_per_launch"]
    rf["traffic_source"] = entry["source"]
    if rf.get("fused_lower_bound_bytes_per_launch"):
        rf["traffic_over_fused_bound"] = rf["traffic"] / rf["fused_lower_bound_bytes_per_launch"]
    rf["hbm_frac_measured"] = rf["traffic"] / (rf["kernel_ms_avg"] * 1e-3) / 1e9 / rf["peak"]
    json.dump(b, open(path, "w"), indent=1)
for extra in ("fanout_ab.txt", "batch_scaling.txt", "ubench_mfma_f32.txt", "ubench_mfma_f64_16.txt", "tile64_phases.txt", "constrained_ab.txt",
              "mpc_throughput.txt", "tile64_batch_scaling.txt", "m2_overlap.txt", "constrained_tile64_ab.txt", "tile64_soak.txt",
              "tile64_chunk_ab.txt", "tile64_centroidal.txt", "c4_dispatch_sweep.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, f"{tag}_{extra}"))
print(json.dumps(traffic, indent=1))
