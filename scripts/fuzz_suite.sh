#!/bin/bash
# The whole GPU suite + the determinism soak against the wave-timing fuzz libraries, one seed after the other.
# usage (GPU box): scripts/fuzz_suite.sh [seeds...] > profiles/r05_fuzz_suite.txt      (default seeds: 1 2 3)
cd "$(dirname "$0")/.."
SEEDS=${@:-1 2 3}
for s in $SEEDS; do
  LIB=$(python -c "from nmpc_amd import build as b; print(b.build_fuzz($s))") || exit 1
  echo "== fuzz seed $s: $LIB"
  export NMPC_HIP_DDP_LIB=$LIB
  # (the fuzz test itself compares product against fuzz and manages the variable on its own; the bench-contract, phase-duration and
  # pool-rate tests assert TIMINGS, which the sleeps distort: deselected)
  python -m pytest tests -m gpu -q --deselect tests/test_gpu_mpc.py::test_phase_durations_on_every_kernel_family --deselect tests/test_gpu_fuzz_sched.py --deselect tests/test_gpu_bench_contract.py --deselect tests/test_gpu_ragged.py::test_solver_pool_with_the_ragged_schedule_overlaps_more 2>&1 | tail -4
  python scripts/determinism_soak.py 20 2>&1 | tail -12
  unset NMPC_HIP_DDP_LIB
done
