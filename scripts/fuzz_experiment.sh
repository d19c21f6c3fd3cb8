#!/bin/bash
# The experiment behind tests/test_gpu_fuzz_sched.py: does the wave-timing fuzz build find a missing barrier that the product
# timing does not?  Four libraries, the same digest soak of the tile-kernel cases at full size (scripts/fuzz_soak.py):
#   product                                 the reference digests
#   race_nofuzz  (-DNMPC_AMD_AB_REOPEN_ADOPT_RACE)                       the barrier of commit 2b8d598 removed, product timing
#   race_fuzz    (-DNMPC_AMD_AB_REOPEN_ADOPT_RACE -DNMPC_AMD_FUZZ_SCHED) the same race under fuzzed wave timing
#   fuzz1        (-DNMPC_AMD_FUZZ_SCHED=1)                               the shipped sources under fuzzed wave timing: must be clean
# usage (GPU box): scripts/fuzz_experiment.sh [reps] > profiles/r05_fuzz_reopened_race.txt
REPS=${1:-6}
cd "$(dirname "$0")/.."
python - <<PY
from nmpc_amd import build as b
b.build_fuzz(1)
b.build_variant("race_fuzz", ["-DNMPC_AMD_FUZZ_SCHED=1", "-DNMPC_AMD_AB_REOPEN_ADOPT_RACE"])
b.build_variant("race_nofuzz", ["-DNMPC_AMD_AB_REOPEN_ADOPT_RACE"])
PY
run() {  # name, library ("" = product)
  if [ -n "$2" ]; then export NMPC_HIP_DDP_LIB=$PWD/$2; else unset NMPC_HIP_DDP_LIB; fi
  python scripts/fuzz_soak.py --reps $REPS --cases full --only tile64 2> /tmp/fuzz_$1.err > /tmp/fuzz_$1.json
  echo "== $1 ($(grep -c . /tmp/fuzz_$1.err) cases)"; cat /tmp/fuzz_$1.err
}
run product ""
run race_nofuzz nmpc_amd/lib/race_nofuzz/libnmpc_hip_ddp.so
run race_fuzz nmpc_amd/lib/race_fuzz/libnmpc_hip_ddp.so
run fuzz1 nmpc_amd/lib/fuzz1/libnmpc_hip_ddp.so
unset NMPC_HIP_DDP_LIB
python - <<PY
import json
ref = json.load(open("/tmp/fuzz_product.json"))
print("== repetitions whose digest differs from the product build's first (of $REPS per case)")
for name in ("product", "race_nofuzz", "race_fuzz", "fuzz1"):
    got = json.load(open("/tmp/fuzz_%s.json" % name))
    bad = {c: sum(d != ref[c]["digests"][0] for d in v["digests"]) for c, v in got.items()}
    print("%-12s total %3d   %s" % (name, sum(bad.values()), {c: n for c, n in bad.items() if n}))
PY
