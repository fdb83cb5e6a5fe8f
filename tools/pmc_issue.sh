set -e
R=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R/issue
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/pmc_issue.py run > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/pmc_issue.py run > $OUT/b.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_issue.py parse $OUT/a $OUT/b > $GRAFT_REPO_ROOT/gpurun_out/$R/pmc_issue.json
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
tail -3 $OUT/a.log; tail -3 $OUT/b.log
python3 -c "
import json; j=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/$R/pmc_issue.json'))
for k,v in j.items():
    print(k[:70]); print('   ', {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,dict)}); print('   ', {a:round(b['per_element_call'],1) for a,b in v.items() if isinstance(b,dict)})
"
