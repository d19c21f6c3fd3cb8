#!/bin/bash
# PMC passes over the bench kernel (each counter set in its own rocprofv3 run; no trace domains mixed in).
# usage: scripts/profile_pmc.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r01}; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT -o $name -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $OUT/$name.log 2>&1
}
BENCH_ARGS="$*"
run A SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
run B SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC
run C SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH
run D FETCH_SIZE
run E WRITE_SIZE TCC_HIT TCC_MISS
ls -R $OUT | head -30
