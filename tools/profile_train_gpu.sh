#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of the pretrain step (BASELINE config 5, tools/pretrain_bench.py).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
ARGS="$REPO/tools/pretrain_bench.py --steps 5 --warmup 2 ${TRAIN_BENCH_FLAGS:-}"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_train -- python3 $ARGS > $OUT/rp_train.log 2>&1 || echo "kt failed" >> $OUT/rp_train.log
python3 $REPO/tools/summarize_rocprof.py $OUT/rp_train $OUT/rocprof_train_kernel_trace_summary.txt
python3 $REPO/tools/trace_overlap.py $OUT/rp_train 60 > $OUT/rocprof_train_overlap.txt 2>&1
find $OUT/rp_train -name '*.csv' -size +2M -delete 2>/dev/null
true
