#!/bin/bash
# Per-kernel register / LDS / occupancy table of the library's HIP sources (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh [file.hip ...]   (default: every .hip under csrc/)
cd "$(dirname "$0")/../torch-gaussian-splatting-rasterizer_amd/csrc" || exit 1
files=${@:-$(ls *.hip)}
for f in $files; do
    extra=""
    case $f in preprocess.hip|helpers.hip) extra="-ffp-contract=off";; blend*.hip) extra="-fno-slp-vectorize";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $extra -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    awk -v F=$f '/Function Name:/ {for (i = 1; i <= NF; ++i) if ($i == "Name:") name = $(i + 1)} /TotalSGPRs:/ {s=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {l=$(NF-1); cmd="echo " name " | c++filt"; cmd | getline d; close(cmd); sub(/\(.*/, "", d); printf "%-14s vgpr %3s sgpr %3s lds %6s scratch %3s occ %2s  %s\n", F, v, s, l, sc, o, d}'
done
