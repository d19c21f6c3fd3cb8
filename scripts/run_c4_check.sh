cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
NMPC_HIP_DDP_LIB=$PWD/nmpc_amd/lib/alt/prof32_pipe.so python scripts/profile_tile32.py 8192 8 | grep -v matrix
timeout 800 python -m pytest tests/test_gpu_fp32.py tests/test_gpu_multirank.py -q 2>&1 | tail -3
OUT=gpurun_out/c4b; mkdir -p $OUT
BENCH="python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-modes"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pmcD -- $BENCH > $OUT/pmcD.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $OUT -o pmcE -- $BENCH > $OUT/pmcE.log 2>&1
python - <<PY
import csv, glob, collections
out = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/pmc?_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "ddp_solve" in r["Kernel_Name"]:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v)/len(v) for k, v in out.items()}
print(m, "traffic MB", (m["FETCH_SIZE"]*2 + m["WRITE_SIZE"])*1024/1e6)
PY
python bench.py --workload c4 --steps 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 value', d['value'], 'kernel ms', d['roofline']['kernel_ms_avg'], 'fp32tol', d['config']['fp32_tolerance_value'])"
python bench.py --workload c4f64 --steps 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4f64 value', d['value'], 'kernel ms', d['roofline']['kernel_ms_avg'])"
python bench.py --workload c5 --steps 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 value', d['value'], 'kernel ms', d['roofline']['kernel_ms_avg'])"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "wave_per_instance or quadrotor or manipulator or centroidal" 2>&1 | tail -3
