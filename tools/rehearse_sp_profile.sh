#!/bin/bash
# Two sequence-parallel ranks of bench.py as two processes on ONE GPU (gloo rendezvous, peer-mapped
# exchange), rank 0 under rocprofv3 --kernel-trace --stats: shows which kernels a sharded step launches.
# usage: tools/rehearse_sp_profile.sh <outdir> [extra bench.py flags]
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$(mkdir -p "$1" && cd "$1" && pwd)"; shift
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 WORLD_SIZE=2 LOCAL_WORLD_SIZE=2 HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
ARGS="--gpus 2 --parallel sp --rehearse-on-one-gpu --sp-comm peer --no-vae --no-cpu-baseline $*"
cd /tmp
RANK=1 LOCAL_RANK=1 timeout -k 10 300 python "$ROOT/bench.py" $ARGS > "$OUT/rank1.out" 2> "$OUT/rank1.err" &
P1=$!
RANK=0 LOCAL_RANK=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o r0 -- python "$ROOT/bench.py" $ARGS \
    > "$OUT/rank0.json" 2> "$OUT/rank0.err"
RC0=$?
wait $P1
RC1=$?
echo "rank0 rc=$RC0 rank1 rc=$RC1"
exit $(( RC0 != 0 || RC1 != 0 ))
