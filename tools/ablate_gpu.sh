#!/bin/bash
# Phase ablation of the fused MBConv kernels (skip expand / depthwise / both): uses the separate -DEFFDET_ABLATE build
# (libeffdet_hip_ablate.so, `make -C ood_object_detection_amd/csrc ablate`); the product library has no such switch.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
make -C ood_object_detection_amd/csrc ablate > gpurun_out/abl_build.log 2>&1 || exit 1
for d in 0 1 2 3; do
  EFFDET_DEBUG_SKIP=$d timeout -k 10 300 python tools/ablate_run.py --steps 5 --warmup 2 --no-cpu-baseline --profile-out gpurun_out/abl_$d.txt > gpurun_out/abl_$d.log 2>&1
done
