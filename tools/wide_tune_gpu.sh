#!/bin/bash
# Geometry sweep of mbconv_wide.hip (variant build -DWIDE_TUNE -> libeffdet_hip_tune.so): waves per workgroup, workgroups per CU
# (LDS-limited residency is what the launch gets; per_cu only gates the LDS check), bands
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=gpurun_out/wide_tune.txt
: > $OUT
run() {  # shape, list of "nw,pc,nb"
  shape=$1; shift
  for t in "$@"; do
    r=$(EFFDET_WIDE_TUNE=$t EFFDET_GEOM_DEBUG=1 EFFDET_LIB_VARIANT=libeffdet_hip_tune.so timeout -k 10 60 python tools/mbconv_layers.py 64 10 $shape 2>&1 | grep -v amdgpu | grep -v "^sum" | tr '\n' ' ' | sed -e 's/mbconv_wide H=[0-9]* W=[0-9]* Cin=[0-9]* mid=[0-9]* k=[0-9] s=[0-9]: //' -e 's/arg0 *H=\([0-9]*\) W=[0-9]* Cin=\([0-9]*\) mid=[0-9]* k=\([0-9]\) s=\([0-9]\) B=64://' | cut -c1-200)
    echo "$shape tune=$t :: $r" >> $OUT
  done
}
run 40,40,80,480,3,1   15,1,2 15,1,1 10,1,1 10,1,2 10,1,4 6,2,1 6,2,2 5,3,1 5,3,2 3,5,1 3,5,2 2,8,1
run 40,40,80,480,5,1   15,1,2 15,1,1 10,1,2 10,1,4 6,2,2 5,2,1 5,2,2 3,4,1 3,4,2 2,6,1
run 40,40,112,672,5,1  14,1,4 14,1,1 14,1,2 7,2,1 7,2,2 7,2,4 6,2,1 6,2,2 3,4,1 3,4,2 2,6,1 2,6,2
run 40,40,112,672,5,2  14,1,4 14,1,1 14,1,2 7,1,1 7,1,2 6,1,2 3,3,1 3,3,2 2,4,1 2,4,2
run 20,20,192,1152,5,1 9,1,1 12,1,1 12,1,2 8,2,1 6,2,1 6,2,2 4,4,1 4,4,2 3,5,1 3,5,2 2,8,1
run 20,20,192,1152,3,1 9,1,1 12,1,1 12,1,2 8,2,1 6,2,1 4,4,1 4,4,2 3,5,1 2,8,1
cat $OUT
