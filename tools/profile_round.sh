#!/bin/bash
# Profiles of one round, run on the MI355X box:  bash tools/profile_round.sh r02
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command (per-kernel average durations)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, of tools/pmc_run.py for every BASELINE config
#      (HBM-side bytes per launch, calibrated on the stream copies of the same run; tools/pmc_parse.py)
# Results land under gpurun_out/<round>/prof/; copy what is to be judged into profiles/<round>/.
set -e -o pipefail
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$R/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PARTS=${2:-trace,pmc}
if [[ $PARTS == *trace* ]]; then
echo "[profile] bench.py under rocprofv3 --kernel-trace --stats"
rm -rf "$OUT/bench_trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 "$ROOT/bench.py" > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
cp "$(find "$OUT/bench_trace" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_kernel_stats.csv"
python3 "$ROOT/tools/trace_stats.py" "$OUT/bench_trace" "--window=$OUT/bench_under_rocprof.json" > "$OUT/bench_kernel_stats_by_grid.csv"
tail -1 "$OUT/bench_kernel_stats_by_grid.csv" | cut -c1-260
head -4 "$OUT/bench_kernel_stats_by_grid.csv" | cut -c1-200
fi
if [[ $PARTS != *pmc* ]]; then exit 0; fi
for cfg in "4 72 10000" "4 72 12500" "4 128 12500" "8 72 20000"; do
  set -- $cfg
  tag=np$1_nlev$2_e$3
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "[profile] $tag $ctr"
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_${tag}_$ctr" -- python3 "$ROOT/tools/pmc_run.py" --np $1 --nlev $2 --elems $3 > "$OUT/pmc_${tag}_$ctr.log" 2>&1
  done
  python3 "$ROOT/tools/pmc_parse.py" "$OUT/pmc_${tag}_FETCH_SIZE" "$OUT/pmc_${tag}_WRITE_SIZE" > "$OUT/pmc_traffic_$tag.json"
  grep -E "hbm_bytes_per_launch|factor_8B_lane" "$OUT/pmc_traffic_$tag.json" || true
done
# keep the summaries, drop the bulky per-dispatch traces of the counter passes
find "$OUT" -name "*kernel_trace.csv" -path "*pmc_*" -delete || true
find "$OUT" -name "*agent_info.csv" -delete || true
du -sh "$OUT"
