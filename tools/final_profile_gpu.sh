# Round-4 profile set (run on the GPU box: bash tools/final_profile_gpu.sh [part]); parts keep single gpurun calls short
set -o pipefail
cd $GRAFT_REPO_ROOT
R=r04_x
PART=${1:-all}
traffic() {   # tag, workload, extra bench args...
  local tag=$1 wl=$2; shift 2
  bash tools/profile_gpu.sh $tag "$@" && python tools/pmc_traffic.py gpurun_out/${tag}_rocprofv3_pmc_FETCH_SIZE.txt gpurun_out/${tag}_rocprofv3_pmc_WRITE_SIZE.txt 0 profiles/pmc_traffic_$(echo $wl | tr / _).json $wl >> gpurun_out/${R}_pmc.log 2>&1
  cp profiles/pmc_traffic_$(echo $wl | tr / _).json gpurun_out/
}
if [ $PART = all ] || [ $PART = d0 ]; then
  traffic ${R} tf_efficientdet_d0/640/64/bf16/90
  echo "d0 bf16 profile done"
  traffic ${R}_accurate tf_efficientdet_d0/640/64/accurate/90 --dtype accurate
  echo "d0 accurate profile done"
  python bench.py --profile-out gpurun_out/${R}_per_launch_events.txt > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench.err
  echo "d0 bench done"; cut -c1-200 gpurun_out/${R}_bench_line.json
  python bench.py --dtype accurate --no-cpu-baseline --no-extras --profile-out gpurun_out/${R}_accurate_per_launch_events.txt > gpurun_out/${R}_accurate_bench_line.json 2>> gpurun_out/${R}_bench.err
  echo "d0 accurate bench done"; cut -c1-200 gpurun_out/${R}_accurate_bench_line.json
fi
if [ $PART = all ] || [ $PART = d2d4 ]; then
  traffic ${R}_d2_768_b32 tf_efficientdet_d2/768/32/bf16/90 --model tf_efficientdet_d2 --image 768 --batch 32
  python bench.py --model tf_efficientdet_d2 --image 768 --batch 32 --no-cpu-baseline --profile-out gpurun_out/${R}_d2_768_b32_per_launch_events.txt > gpurun_out/${R}_d2_768_b32_bench_line.json 2>> gpurun_out/${R}_bench.err
  echo "d2 done"; cut -c1-200 gpurun_out/${R}_d2_768_b32_bench_line.json
  traffic ${R}_d4_1024_b8_softnms tf_efficientdet_d4/1024/8/bf16/90 --model tf_efficientdet_d4 --image 1024 --batch 8 --soft-nms
  python bench.py --model tf_efficientdet_d4 --image 1024 --batch 8 --soft-nms --auroc-surrogate --no-cpu-baseline --profile-out gpurun_out/${R}_d4_1024_b8_softnms_per_launch_events.txt > gpurun_out/${R}_d4_1024_b8_softnms_bench_line.json 2>> gpurun_out/${R}_bench.err
  echo "d4 done"; cut -c1-200 gpurun_out/${R}_d4_1024_b8_softnms_bench_line.json
fi
if [ $PART = all ] || [ $PART = sq ]; then
  PMC_SET="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_BUSY_CYCLES" bash tools/pmc_gpu.sh && python tools/pmc_sq_table.py gpurun_out/rocprof_pmc_sq2_summary.txt 3 > gpurun_out/${R}_pmc_SQ_utilisation.txt
  PMC_MODE=accurate PMC_SET="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_BUSY_CYCLES" bash tools/pmc_gpu.sh && python tools/pmc_sq_table.py gpurun_out/rocprof_pmc_sq2_summary.txt 3 > gpurun_out/${R}_accurate_pmc_SQ_utilisation.txt
  echo "sq done"
fi
if [ $PART = all ] || [ $PART = train ]; then
  python tools/pretrain_bench.py --graph --steps 20 > gpurun_out/${R}_pretrain_bench_line_b8.json 2> gpurun_out/${R}_pretrain.err
  cut -c1-300 gpurun_out/${R}_pretrain_bench_line_b8.json
  python tools/pretrain_bench.py --graph --steps 10 --batch 64 > gpurun_out/${R}_pretrain_bench_line_b64.json 2>> gpurun_out/${R}_pretrain.err
  python tools/pretrain_bench.py --graph --steps 10 --model efficientdet_d0 > gpurun_out/${R}_pretrain_bench_line_b8_pad0_efficientdet_d0.json 2>> gpurun_out/${R}_pretrain.err
  python tools/train_sections.py > gpurun_out/${R}_train_sections.txt 2>> gpurun_out/${R}_pretrain.err
  echo "train done"
fi
