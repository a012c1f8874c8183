set -o pipefail
cd $GRAFT_REPO_ROOT
W0=tf_efficientdet_d0/640/64/bf16/90
bash tools/profile_gpu.sh r03_x && python tools/pmc_traffic.py gpurun_out/r03_x_rocprofv3_pmc_FETCH_SIZE.txt gpurun_out/r03_x_rocprofv3_pmc_WRITE_SIZE.txt 0 profiles/pmc_traffic_tf_efficientdet_d0_640_64_bf16_90.json $W0 > gpurun_out/r03_x_pmc.log 2>&1
cp profiles/pmc_traffic_tf_efficientdet_d0_640_64_bf16_90.json gpurun_out/
echo "d0 profile done"
python bench.py --profile-out gpurun_out/r03_x_per_launch_events.txt > gpurun_out/r03_x_bench_line.json 2> gpurun_out/r03_x_bench.err
echo "d0 bench done"; cut -c1-200 gpurun_out/r03_x_bench_line.json
bash tools/profile_gpu.sh r03_x_d2_768_b32 --model tf_efficientdet_d2 --image 768 --batch 32 && python tools/pmc_traffic.py gpurun_out/r03_x_d2_768_b32_rocprofv3_pmc_FETCH_SIZE.txt gpurun_out/r03_x_d2_768_b32_rocprofv3_pmc_WRITE_SIZE.txt 0 profiles/pmc_traffic_tf_efficientdet_d2_768_32_bf16_90.json tf_efficientdet_d2/768/32/bf16/90 >> gpurun_out/r03_x_pmc.log 2>&1
cp profiles/pmc_traffic_tf_efficientdet_d2_768_32_bf16_90.json gpurun_out/
python bench.py --model tf_efficientdet_d2 --image 768 --batch 32 --no-cpu-baseline --profile-out gpurun_out/r03_x_d2_768_b32_per_launch_events.txt > gpurun_out/r03_x_d2_768_b32_bench_line.json 2>> gpurun_out/r03_x_bench.err
echo "d2 done"; cut -c1-200 gpurun_out/r03_x_d2_768_b32_bench_line.json
bash tools/profile_gpu.sh r03_x_d4_1024_b8_softnms --model tf_efficientdet_d4 --image 1024 --batch 8 --soft-nms && python tools/pmc_traffic.py gpurun_out/r03_x_d4_1024_b8_softnms_rocprofv3_pmc_FETCH_SIZE.txt gpurun_out/r03_x_d4_1024_b8_softnms_rocprofv3_pmc_WRITE_SIZE.txt 0 profiles/pmc_traffic_tf_efficientdet_d4_1024_8_bf16_90.json tf_efficientdet_d4/1024/8/bf16/90 >> gpurun_out/r03_x_pmc.log 2>&1
cp profiles/pmc_traffic_tf_efficientdet_d4_1024_8_bf16_90.json gpurun_out/
python bench.py --model tf_efficientdet_d4 --image 1024 --batch 8 --soft-nms --auroc-surrogate --no-cpu-baseline --profile-out gpurun_out/r03_x_d4_1024_b8_softnms_per_launch_events.txt > gpurun_out/r03_x_d4_1024_b8_softnms_bench_line.json 2>> gpurun_out/r03_x_bench.err
echo "d4 done"; cut -c1-200 gpurun_out/r03_x_d4_1024_b8_softnms_bench_line.json
PMC_SET="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_BUSY_CYCLES" bash tools/pmc_gpu.sh && python tools/pmc_sq_table.py gpurun_out/rocprof_pmc_sq2_summary.txt 3 > gpurun_out/r03_x_pmc_SQ_utilisation.txt
echo "sq done"
