#!/bin/bash
# Development aid: builds build_variants/libomnihip_<name>.so with extra compiler flags for ONE kernel file
# (all other objects are the in-tree ones).  usage: tools/build_variant.sh <name> <file.hip> <flags...>
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; FILE="$2"; shift 2
mkdir -p "$ROOT/build_variants"
cd "$ROOT/omni-video_amd/csrc"
BASE="${FILE%.hip}"
EXTRA=""
[ "$BASE" = "attn" ] && EXTRA="-fno-slp-vectorize"
[ "$BASE" = "sampler" ] && EXTRA="-ffp-contract=off"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -fno-gpu-rdc $EXTRA "$@" -c "$FILE" -o "$ROOT/build_variants/${BASE}_${NAME}.o"
OBJS=""
for o in api norm gemm gemm256 attn elementwise sampler vae t5 video fp8 peer; do
  if [ "$o" = "$BASE" ]; then OBJS="$OBJS $ROOT/build_variants/${BASE}_${NAME}.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_variants/libomnihip_${NAME}.so" $OBJS
echo "$ROOT/build_variants/libomnihip_${NAME}.so"
