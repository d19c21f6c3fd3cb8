#!/bin/bash
# Per-kernel times of the batched FMPC solve (rocprofv3 --kernel-trace --stats), printed as a table.
# usage (on the GPU box, from the repo root): scripts/fmpc_profile.sh [B] [T] [max_iter]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/fmpc_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fmpc -- python scripts/fmpc_run.py ${1:-4096} ${2:-200} ${3:-5} 10 > $OUT/run.log 2>&1
cat $OUT/run.log | tail -1
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("%-72s %7s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for r in rows:
    print("%-72s %7s %12.1f %10.2f %6s" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
