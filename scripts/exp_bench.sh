#!/bin/bash
# usage: exp_bench.sh <variant names...>: bench lines of the tile-kernel workloads on the product library and on variant libraries
cd "$GRAFT_REPO_ROOT"
for V in "" "$@"; do
  if [ -n "$V" ]; then export NMPC_HIP_DDP_LIB=$PWD/nmpc_amd/lib/$V/libnmpc_hip_ddp.so; else unset NMPC_HIP_DDP_LIB; fi
  for WL in c4 c4f64 c5 centroidal; do
    python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-extra-modes --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${V:-product}', '$WL', round(d['value'], 1), 'it/s', round(d['ms_per_step'], 3), 'ms', d['roofline'].get('kernel'))"
  done
done
