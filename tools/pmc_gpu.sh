#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
SET=${PMC_SET:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"}
timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/rp_sq2 -- python3 $REPO/tools/run_forward.py 3 ${PMC_MODE:-bf16} > $OUT/rp_sq2.log 2>&1 || echo failed >> $OUT/rp_sq2.log
python3 $REPO/tools/summarize_rocprof.py $OUT/rp_sq2 $OUT/rocprof_pmc_sq2_summary.txt
find $OUT/rp_sq2 -name '*.csv' -size +2M -delete 2>/dev/null
true
