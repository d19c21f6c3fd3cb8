"""Overlap structure of a rocprofv3 kernel trace: per solve kernel its queue, start, duration, and how many solve kernels ran beside it.
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python scripts/m2_overlap.py 8 4
    python scripts/trace_overlap.py <dir> [name substring, default ddp_solve]"""
import csv, glob, os, sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "ddp_solve"
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append(r)
if not rows:
    sys.exit("no kernel trace rows under " + d)
cols = rows[0].keys()
qcol = "Queue_Id" if "Queue_Id" in cols else None
scol = "Stream_Id" if "Stream_Id" in cols else None
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get(qcol, "?"), r.get(scol, "?")) for r in rows]
k.sort()
t0 = k[0][0]
solve = [x for x in k if pat in x[2]]
print(f"{len(k)} dispatches, {len(solve)} matching '{pat}'; columns: {list(cols)}")
# skip the warm-up: keep the last N where N = count in the timed region is unknown -> print all, compactly
for s, e, name, q, st in solve[-40:]:
    beside = sum(1 for s2, e2, *_ in solve if s2 < e and e2 > s) - 1
    print(f"  start {1e-6 * (s - t0):9.3f} ms  dur {1e-6 * (e - s):8.3f} ms  queue {q:>3} stream {st:>3}  overlapping solve kernels {beside}  {name[:40]}")
# occupancy histogram: time with n solve kernels in flight
ev = sorted([(s, 1) for s, *_ in solve] + [(e, -1) for _, e, *_ in solve])
cur, last, hist = 0, ev[0][0], {}
for t, dlt in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += dlt
    last = t
tot = sum(hist.values())
print("time with n solve kernels in flight:", {n: f"{1e-6 * v:.2f} ms ({100.0 * v / tot:.0f} %)" for n, v in sorted(hist.items())})
