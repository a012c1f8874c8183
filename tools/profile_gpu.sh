#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats, then FETCH_SIZE and WRITE_SIZE PMC passes (separate runs).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
# one batch at a time, one launch chain, eager: every kernel is its own serialized dispatch, so the per-kernel averages are
# comparable with the HIP-event times behind roofline.achieved (the timed bench itself keeps three batches in flight)
# usage: bash tools/profile_gpu.sh [tag] [extra bench.py args, e.g. --model tf_efficientdet_d2 --image 768 --batch 32]
TAG=${1:-d0}; shift
rm -rf $OUT/rp_kt $OUT/rp_fetch $OUT/rp_write           # one configuration per directory: the summaries glob everything under them
ARGS="$REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-graph --in-flight 1 --sub-batches 1 $@"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_kt -- python3 $ARGS > $OUT/rp_kt.log 2>&1 || echo "kt failed" >> $OUT/rp_kt.log
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rp_fetch -- python3 $ARGS > $OUT/rp_fetch.log 2>&1 || echo "fetch failed" >> $OUT/rp_fetch.log
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/rp_write -- python3 $ARGS > $OUT/rp_write.log 2>&1 || echo "write failed" >> $OUT/rp_write.log
python3 $REPO/tools/summarize_rocprof.py $OUT/rp_kt $OUT/${TAG}_rocprofv3_kernel_trace_stats.txt
python3 $REPO/tools/summarize_rocprof.py $OUT/rp_fetch $OUT/${TAG}_rocprofv3_pmc_FETCH_SIZE.txt
python3 $REPO/tools/summarize_rocprof.py $OUT/rp_write $OUT/${TAG}_rocprofv3_pmc_WRITE_SIZE.txt
# keep the merged-back payload small
find $OUT/rp_kt $OUT/rp_fetch $OUT/rp_write -name '*.csv' -size +2M -delete 2>/dev/null
true
