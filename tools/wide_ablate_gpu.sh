#!/bin/bash
# Phase ablation of mbconv_wide.hip (variant builds libeffdet_hip_abl<N>.so, see the Makefile's `variant` target) + SQ counters
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
SHAPES="40,40,80,480,3,1 40,40,80,480,5,1 40,40,112,672,5,1 40,40,112,672,5,2 20,20,192,1152,5,1 20,20,192,1152,3,1"
echo "== product" > gpurun_out/wide_abl.txt
EFFDET_GEOM_DEBUG=1 timeout -k 10 120 python tools/mbconv_layers.py 64 10 $SHAPES 2>&1 | grep -v amdgpu.ids >> gpurun_out/wide_abl.txt
for t in "$@"; do
  echo "== variant $t (ablN: WIDE_ABLATE=N - 1 no hand-off, 2 no depthwise, 4 no expand, 16 no SiLU; gN: WIDE_G=N)" >> gpurun_out/wide_abl.txt
  EFFDET_LIB_VARIANT=libeffdet_hip_$t.so timeout -k 10 120 python tools/mbconv_layers.py 64 10 $SHAPES 2>&1 | grep -v amdgpu.ids >> gpurun_out/wide_abl.txt
done
cat gpurun_out/wide_abl.txt
