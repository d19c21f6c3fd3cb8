#!/usr/bin/env python3
"""Turn one scripts/profile_r06.sh session (gpurun_out/profile_<tag>/) into the tracked summaries under profiles/: what
collect_profiles_r05.py writes (kernel stats, PMC summaries, calibration, bench lines, hbm_traffic.json keyed on the device-source hash)
plus the round's own files (stream throughput, bipedal kernel A/B, tile kernel phase split, the default bench line).
    python scripts/collect_profiles_r06.py <tag> <gpurun_out/profile_tag>"""
import json, os, shutil, subprocess, sys
tag, src = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run([sys.executable, os.path.join(root, "scripts", "collect_profiles_r05.py"), tag, src], check=True)
for name in ("stream_throughput.txt", "c3_kernel_ab.txt", "tile_phases.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(root, "profiles", f"{tag}_{name}"))
f = os.path.join(src, "bench_default.txt")
if os.path.exists(f):
    lines = [l for l in open(f) if l.startswith("{")]
    if lines:
        json.dump(json.loads(lines[-1]), open(os.path.join(root, "profiles", f"{tag}_bench_default.json"), "w"), indent=1)
