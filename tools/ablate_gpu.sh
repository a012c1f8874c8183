#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for d in 0 1 2 3; do
  EFFDET_DEBUG_SKIP=$d timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-out gpurun_out/abl_$d.txt > gpurun_out/abl_$d.log 2>&1
done
