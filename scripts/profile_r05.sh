#!/bin/bash
# Round-5 profiling session for profiles/: per workload kernel-trace stats + HBM-traffic counters of the bench command, SQ counter
# passes for the headline (c2), the tile kernels (c4, c5, centroidal) and FMPC, counter calibration, unprofiled bench lines.
# usage (on the GPU box, via gpurun): scripts/profile_r05.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05}
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
python -c "from nmpc_amd import build; print(build.source_hash())" > $OUT/source_hash.txt
[ -x scripts/ubench_hbm_counters ] || hipcc --offload-arch=gfx950 -O2 -w scripts/ubench_hbm_counters.hip -o scripts/ubench_hbm_counters
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o calF -- ./scripts/ubench_hbm_counters > $OUT/calF.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o calW -- ./scripts/ubench_hbm_counters > $OUT/calW.log 2>&1
for WL in c2 c4 c3 c5 c4f64 centroidal; do
  BENCH="python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-extra-modes --no-secondary --min-seconds 0"
  # (the kernel-trace pass runs 200 steps: over 32 launches from a cold start the average is ~5 % above the steady state the bench
  # line's HIP events see — 435 against 415 us on the headline kernel, minimum 415 — and the roofline is quoted on this average)
  STATS_BENCH="python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --no-extra-modes --no-secondary --min-seconds 0"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats_$WL -- $STATS_BENCH > $OUT/stats_$WL.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pmcD_$WL -- $BENCH > $OUT/pmcD_$WL.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $OUT -o pmcE_$WL -- $BENCH > $OUT/pmcE_$WL.log 2>&1
done
for WL in c2 c4 c5 centroidal; do
  BENCH="python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-extra-modes --no-secondary --min-seconds 0"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -o pmcA_$WL -- $BENCH > $OUT/pmcA_$WL.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT -o pmcB_$WL -- $BENCH > $OUT/pmcB_$WL.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 --output-format csv -d $OUT -o pmcC_$WL -- $BENCH > $OUT/pmcC_$WL.log 2>&1
done
BENCH="python bench.py --workload fmpc --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats_fmpc -- python bench.py --workload fmpc --steps 100 --warmup 10 --no-cpu-baseline > $OUT/stats_fmpc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pmcD_fmpc -- $BENCH > $OUT/pmcD_fmpc.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $OUT -o pmcE_fmpc -- $BENCH > $OUT/pmcE_fmpc.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -o pmcA_fmpc -- $BENCH > $OUT/pmcA_fmpc.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT -o pmcB_fmpc -- $BENCH > $OUT/pmcB_fmpc.log 2>&1
python bench.py --workload fmpc > $OUT/bench_fmpc.txt 2>&1
python bench.py --workload c4 --steps 100 > $OUT/bench_c4.txt 2>&1
for WL in c3 c5 c4f64 centroidal; do
  python bench.py --workload $WL --steps 30 > $OUT/bench_$WL.txt 2>&1
done
python bench.py --no-secondary > $OUT/bench_c2.txt 2>&1
python scripts/batch_scaling.py > $OUT/batch_scaling.txt 2>&1
python scripts/constrained_ab.py > $OUT/constrained_ab.txt 2>&1
python scripts/mpc_throughput.py > $OUT/mpc_throughput.txt 2>&1
# raw counter dumps are large: keep the csv files the collector reads
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
