#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh gpurun_out/prof_r2 [workload]   then, back home:  tools/summarize_profiles.py gpurun_out/prof_r2 profiles/r2 [workload]
# One --kernel-trace --stats pass, then the PMC counters in SEPARATE passes (never combined with sys/runtime traces);
# the program follows `--` directly (no env/bash hop).  Steps are joined with && so nothing runs after a failure.
set -o pipefail
OUT=${1:-gpurun_out/prof}
WL=${2:-bicycle}
ROOT=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
cd /tmp || exit 1
# one view per launch sequence, one frame in flight: what bench.py's stage times and roofline object describe (kernels of overlapping
# frames share the machine, their durations are not attributable; a launch sequence of four views is four frames per kernel)
B="$ROOT/bench.py --workload $WL --no-cpu-baseline --no-psnr --legs= --frames-in-flight 1 --views-per-launch 1 ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 $B --steps 30 --warmup 5 > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" &&
for spec in "FETCH_SIZE:FETCH_SIZE" "WRITE_SIZE:WRITE_SIZE" \
            "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
            "sq2:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "sq3:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    name=${spec%%:*}; ctrs=${spec#*:}
    echo "pmc pass $name: $ctrs"
    timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT" -o "pmc_$name" -- python3 $B --steps 10 --warmup 3 > /dev/null 2> "$OUT/pmc_$name.err" || exit 1
done
ls "$OUT"
