#!/bin/bash
# One steady-state batch's per-dispatch timeline under rocprofv3 (run on the GPU box from the repo root):
#   tools/trace_batch.sh OUTDIR K [batches] [camera-set 0|1]
set -o pipefail
OUT=$1; shift
ROOT=$(pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
cd /tmp || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 "$ROOT/tools/batch_profile.py" "$@" > "$OUT/run.log" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
python3 "$ROOT/tools/frame_timeline.py" "$OUT/trace_kernel_trace.csv" 3
