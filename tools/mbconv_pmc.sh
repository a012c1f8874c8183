#!/bin/bash
# SQ / TA counters of one MBConv front launch: bash tools/mbconv_pmc.sh <tag> H W Cin mid k s [B] [reps] [gated]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
python3 $REPO/tools/mbconv_probe.py "$@" > $OUT/mbp_$TAG.txt 2>&1
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rm -rf $OUT/mbp_rp_$i
  timeout -k 10 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/mbp_rp_$i -- python3 $REPO/tools/mbconv_probe.py "$@" > $OUT/mbp_rp_$i.log 2>&1 || echo failed >> $OUT/mbp_rp_$i.log
  python3 $REPO/tools/summarize_rocprof.py $OUT/mbp_rp_$i $OUT/mbp_${TAG}_pmc$i.txt > /dev/null
  rm -rf $OUT/mbp_rp_$i
done
true
