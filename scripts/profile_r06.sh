#!/bin/bash
# Round-6 profiling session for profiles/: the round-5 session (per workload kernel-trace stats + HBM-traffic counters of the bench
# command, SQ counter passes, counter calibration, unprofiled bench lines) plus the round's own measurements — streamed solves,
# the bipedal kernel A/B, the centroidal gains A/B is run separately (scripts/centroidal_gains_ab.py needs the A/B libraries).
# usage (on the GPU box, via gpurun): scripts/profile_r06.sh [tag]
TAG=${1:-r06}
bash "$(dirname "$0")/profile_r05.sh" $TAG
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/profile_$TAG
{ python scripts/stream_throughput.py 32768 4096 8 16; python scripts/stream_throughput.py 131072 4096 8; python scripts/stream_throughput.py 262144 4096 4 6 8 12 16; } 2>&1 | grep -v amdgpu.ids > $OUT/stream_throughput.txt
python scripts/c3_kernel_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/c3_kernel_ab.txt
for WL in c4 c4f64 c5 centroidal c3; do python scripts/ab_workload.py $WL main; done 2>&1 | grep -v amdgpu.ids > $OUT/tile_phases.txt
python bench.py > $OUT/bench_default.txt 2> $OUT/bench_default.err
du -sh $OUT
