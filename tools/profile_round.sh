#!/bin/bash
# Regenerates the judged artifacts under profiles/ for round $1 (e.g. r01) on a GPU box.  Run from the repo root through
# gpurun; everything lands in gpurun_out/profile_$1/ and is copied into profiles/ afterwards.
#   1. the default bench line                          -> bench.json
#   2. rocprofv3 --kernel-trace --stats of that command -> kernel_stats.csv + kernel_summary.txt
#   3. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) -> hbm_traffic_pmc.json
#   4. one PMC pass for the matrix-core utilisation     -> mfma_util_pmc.json
set -euo pipefail
R=${1:-r01}
W=${2:-c2_base_bf16}      # bench.py --workload (second argument; the default bench line is C2)
ST=${3:-100}              # timed steps
OUT=gpurun_out/profile_$R
[ "$W" != c2_base_bf16 ] && OUT=gpurun_out/profile_${R}_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 bench.py --workload $W --steps $ST > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $W --steps $ST --no-cpu-baseline --no-live-traffic --no-extra-workloads > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name '*_kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 tools/rocprof_summary.py $OUT/trace $((ST + 10)) > $OUT/kernel_summary.txt
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --workload $W --no-cpu-baseline --no-roofline --no-live-traffic --no-extra-workloads --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --workload $W --no-cpu-baseline --no-roofline --no-live-traffic --no-extra-workloads --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over 'bench.py --workload $W --steps 3 --warmup 1' ($R); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); L2<->fabric traffic, Infinity-Cache hits included" > $OUT/hbm_traffic_pmc.json
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --workload $W --no-cpu-baseline --no-roofline --no-live-traffic --no-extra-workloads --steps 3 --warmup 1 > $OUT/pmc_mfma.log 2>&1
python3 tools/pmc_mfma.py $OUT/pmc_mfma "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE over 'bench.py --workload $W --steps 3 --warmup 1' ($R): busy cycles of the matrix pipes (summed over the 4 SIMDs of every CU) against the cycles the CUs had work" > $OUT/mfma_util_pmc.json
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
cat $OUT/bench.json
head -14 $OUT/kernel_summary.txt
