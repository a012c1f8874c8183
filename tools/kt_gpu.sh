#!/bin/bash
# rocprofv3 kernel trace of a few DetBenchPredict forwards (no counters) -> per-kernel summary
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_kt2 -- python3 $REPO/tools/run_forward.py 3 > $OUT/rp_kt2.log 2>&1 || echo failed >> $OUT/rp_kt2.log
python3 $REPO/tools/summarize_rocprof.py $OUT/rp_kt2 $OUT/rocprof_kt2_summary.txt
find $OUT/rp_kt2 -name '*.csv' -size +2M -delete 2>/dev/null
true
