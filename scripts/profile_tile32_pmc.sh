#!/bin/bash
# PMC passes over the fp32 tile kernel (quadrotor_f32, B = 8192); usage: scripts/profile_tile32_pmc.sh <tag> [max_iter]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/tile32_pmc_${1:-r02}
MI=${2:-8}
mkdir -p $OUT
CMD="python scripts/tile32_run.py 8192 $MI 6"
rocprofv3 -L > $OUT/counters_available.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -o pmcA -- $CMD > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT -o pmcB -- $CMD > $OUT/pmcB.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 --output-format csv -d $OUT -o pmcC -- $CMD > $OUT/pmcC.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pmcD -- $CMD > $OUT/pmcD.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $OUT -o pmcE -- $CMD > $OUT/pmcE.log 2>&1
python - <<PY
import csv, glob, collections
out = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/pmc?_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "tile32" in r["Kernel_Name"]:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fo:
    fo.write("ddp_solve_tile32_kernel<quadrotor_f32>  B = 8192, max_iter $MI: per-launch means (SQ cycle counters in quad-cycles)\n")
    for k in sorted(out):
        fo.write(f"  {k:32s} n={len(out[k]):2d} mean={sum(out[k]) / len(out[k]):18.1f}\n")
print(open("$OUT/summary.txt").read())
PY
grep -E "tile32|Name" $OUT/stats_kernel_stats.csv | head -5
tail -2 $OUT/pmcB.log $OUT/pmcC.log
