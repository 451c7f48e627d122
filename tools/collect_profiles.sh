#!/bin/bash
# Copies what tools/profile_round.sh left under gpurun_out/profile_<round>[_<workload>]/ into profiles/ under the judged names:
#   profiles/<round>[_<workload>]_{bench.json, bench_kernel_stats.csv, bench_kernel_summary.txt, hbm_traffic_pmc.json, mfma_util_pmc.json}
set -euo pipefail
R=${1:-r05}
for d in gpurun_out/profile_${R} gpurun_out/profile_${R}_*; do
    [ -d "$d" ] || continue
    w=${d#gpurun_out/profile_${R}}
    p=profiles/${R}${w}
    grep '^{' "$d/bench.json" | tail -1 > "${p}_bench.json"
    cp "$d/kernel_stats.csv" "${p}_bench_kernel_stats.csv"
    cp "$d/kernel_summary.txt" "${p}_bench_kernel_summary.txt"
    cp "$d/hbm_traffic_pmc.json" "${p}_hbm_traffic_pmc.json"
    cp "$d/mfma_util_pmc.json" "${p}_mfma_util_pmc.json"
    echo "$p"
done
