#!/bin/bash
# A/B library: rebuild ONE translation unit with extra flags and link it with the current objects of the others into
# nmpc_amd/lib/alt/<name>.so (load it with NMPC_HIP_DDP_LIB=...).   usage: scripts/build_alt.sh <name> <source.hip> [flags...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
mkdir -p nmpc_amd/lib/alt
OBJ=nmpc_amd/lib/alt/${NAME}_$(basename $SRC .hip).o
EXTRA=""
[ "$SRC" = "builtin_models.hip" ] && EXTRA="-mllvm --amdgpu-mfma-vgpr-form"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $EXTRA "$@" -c nmpc_amd/csrc/$SRC -o $OBJ
OBJS=""
for o in nmpc_amd/lib/obj/*.o; do
  if [ "$(basename $o)" = "$(basename $SRC .hip).o" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o nmpc_amd/lib/alt/$NAME.so $OBJS -Wl,-rpath,/opt/rocm/lib
echo nmpc_amd/lib/alt/$NAME.so
