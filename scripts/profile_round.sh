#!/bin/bash
# One profiling session for profiles/: kernel-trace stats of the bench command, PMC passes, counter calibration.
# usage (on the GPU box, via gpurun): scripts/profile_round.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r01}
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
# microbenchmark binaries (git-ignored): build what is missing
build_ubench() { [ -x scripts/$2 ] || hipcc --offload-arch=gfx950 -O2 -w scripts/$1 -o scripts/$2; }
build_ubench ubench_clock.hip ubench_clock
build_ubench ubench_latency.hip ubench_latency
build_ubench ubench_recip.hip ubench_recip
build_ubench ubench_hbm_counters.hip ubench_hbm_counters
build_ubench ubench_mfma_f64_4x4.hip ubench_mfma4
build_ubench ubench_mfma4_rate.hip ubench_mfma4_rate
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
echo "== bench (unprofiled)" > $OUT/bench.txt
python bench.py --steps 50 --warmup 5 >> $OUT/bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT -o pmcA -- $BENCH > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU --output-format csv -d $OUT -o pmcB -- $BENCH > $OUT/pmcB.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $OUT -o pmcC -- $BENCH > $OUT/pmcC.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pmcD -- $BENCH > $OUT/pmcD.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $OUT -o pmcE -- $BENCH > $OUT/pmcE.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o calF -- ./scripts/ubench_hbm_counters > $OUT/calF.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o calW -- ./scripts/ubench_hbm_counters > $OUT/calW.log 2>&1
./scripts/ubench_clock > $OUT/ubench_clock.txt 2>&1
./scripts/ubench_latency >> $OUT/ubench_clock.txt 2>&1
./scripts/ubench_recip >> $OUT/ubench_clock.txt 2>&1
if [ -f nmpc_amd/lib/alt/prof.so ]; then
  echo "== quad kernel (default for this batch)" > $OUT/roles.txt
  NMPC_HIP_DDP_LIB=$PWD/nmpc_amd/lib/alt/prof.so python scripts/profile_quad.py >> $OUT/roles.txt 2>&1
  echo "== two-wave kernel (NMPC_HIP_DDP_KERNEL=2w)" >> $OUT/roles.txt
  NMPC_HIP_DDP_KERNEL=2w NMPC_HIP_DDP_LIB=$PWD/nmpc_amd/lib/alt/prof.so python scripts/profile_2w.py >> $OUT/roles.txt 2>&1
fi
python scripts/batch_scaling.py > $OUT/batch_scaling.txt 2>&1
./scripts/ubench_mfma4 > $OUT/ubench_mfma4.txt 2>&1
./scripts/ubench_mfma4_rate >> $OUT/ubench_mfma4.txt 2>&1
ls $OUT
