#!/bin/bash
# Per-dispatch timeline of one steady-state frame of a tile-row shard under rocprofv3:  tools/trace_shard.sh OUTDIR G r
set -o pipefail
OUT=$1; G=$2; R=$3
ROOT=$(pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
cd /tmp || exit 1
GSR_SLOTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 "$ROOT/tools/shard_timing.py" bicycle $G $R > "$OUT/shard.log" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
python3 "$ROOT/tools/frame_timeline.py" "$OUT/trace_kernel_trace.csv" 8
