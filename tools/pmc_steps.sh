# HBM-side traffic of the step-loop kernels (NP=8, NLEV=128; the NLEV=72 loop is measured live by bench.py):  bash tools/pmc_steps.sh r05
set -e
R=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R/pmc_steps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "8 72 20000" "4 128 12500"; do
  set -- $cfg
  tag=np$1_nlev$2_e$3
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${tag}_$ctr -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py --np $1 --nlev $2 --elems $3 --steps 20 > $OUT/${tag}_$ctr.log 2>&1
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_parse.py $OUT/${tag}_FETCH_SIZE $OUT/${tag}_WRITE_SIZE > $GRAFT_REPO_ROOT/gpurun_out/$R/pmc_traffic_steps_$tag.json
  grep -E "steps_hbm_bytes_per_launch|\"hbm_bytes_per_launch|steps_kernel\"" $GRAFT_REPO_ROOT/gpurun_out/$R/pmc_traffic_steps_$tag.json | head -4
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
