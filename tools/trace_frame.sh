#!/bin/bash
# One steady-state frame's per-dispatch timeline under rocprofv3 (run on the GPU box from the repo root):
#   tools/trace_frame.sh OUTDIR [bench.py args ...]      GSR_LIB_PATH selects the library
set -o pipefail
OUT=$1; shift
ROOT=$(pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
cd /tmp || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-psnr --legs= --frames-in-flight 1 --steps 30 --warmup 5 "$@" > "$OUT/bench.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
python3 "$ROOT/tools/frame_timeline.py" "$OUT/trace_kernel_trace.csv" +20
