#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/rp_sq -- python3 $REPO/tools/run_forward.py 3 > $OUT/rp_sq.log 2>&1 || echo failed >> $OUT/rp_sq.log
python3 $REPO/tools/summarize_rocprof.py $OUT/rp_sq $OUT/rocprof_pmc_sq_summary.txt
find $OUT/rp_sq -name '*.csv' -size +2M -delete 2>/dev/null
true
