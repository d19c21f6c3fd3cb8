#!/bin/bash
# PMC passes over the wave-per-instance kernel (quadrotor + manipulator at B = 8192); usage: scripts/profile_wpi_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/wpi_pmc_${1:-r01}
mkdir -p $OUT
CMD="python scripts/config_throughput.py 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -o pmcA -- $CMD > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH --output-format csv -d $OUT -o pmcB -- $CMD > $OUT/pmcB.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU --output-format csv -d $OUT -o pmcC -- $CMD > $OUT/pmcC.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pmcD -- $CMD > $OUT/pmcD.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $OUT -o pmcE -- $CMD > $OUT/pmcE.log 2>&1
python - <<PY
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pmc?_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "wpi" in r["Kernel_Name"]:
            model = "quadrotor" if "Quadrotor" in r["Kernel_Name"] else "manipulator"
            out[model][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fo:
    for model, cs in out.items():
        fo.write(f"ddp_solve_wpi_kernel<{model}>  B = 8192, max_iter 4: per-launch means (SQ cycle counters in quad-cycles)\n")
        for k in sorted(cs):
            fo.write(f"  {k:28s} n={len(cs[k]):2d} mean={sum(cs[k]) / len(cs[k]):18.1f}\n")
print(open("$OUT/summary.txt").read())
PY
