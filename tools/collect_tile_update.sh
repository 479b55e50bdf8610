#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun) from the repo root after a change of the time-tiled kernel only: re-collects what depends on
# it — the tile:* entries of valu.json (the other entries, and the bench lines made with them, stay), the in-loop histogram
# tables and the tile-vs-relaunched-fused counters — under gpurun_out/r03t/.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03t; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
SQA="SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
SQC="SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_CVT"
echo '{}' > $OUT/valu_tile.json
for S in a b; do eval C=\$SQ$(echo $S | tr a-z A-Z); timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_sq${S}_fused64 -- python3 $R/tools/pmc_workload_fused.py 1000000 f64 96 8 > $OUT/pmc_sq${S}_fused64.log 2>&1; echo "fused64 $S"; done
for S in a b; do eval C=\$SQ$(echo $S | tr a-z A-Z); timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_sq${S}_co2 -- python3 $R/tools/pmc_workload_fused.py 1000000 f64 96 8 co2 > $OUT/pmc_sq${S}_co2.log 2>&1; echo "co2 $S"; done
python3 $R/tools/pmc_valu.py $OUT/sq_counters_tile_f64_1M.csv $OUT/valu_tile.json 96 1000000 8 $OUT/pmc_sqa_fused64 $OUT/pmc_sqb_fused64 $OUT/pmc_sqa_co2 $OUT/pmc_sqb_co2 > /dev/null
for S in a b c; do eval C=\$SQ$(echo $S | tr a-z A-Z); timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_sq${S}_fused32 -- python3 $R/tools/pmc_workload_fused.py 4000000 f32 96 8 > $OUT/pmc_sq${S}_fused32.log 2>&1; echo "fused32 $S"; done
python3 $R/tools/pmc_valu.py $OUT/sq_counters_tile_f32_4M.csv $OUT/valu_tile.json 96 4000000 8 $OUT/pmc_sqa_fused32 $OUT/pmc_sqb_fused32 $OUT/pmc_sqc_fused32 > /dev/null
rm -rf $OUT/pmc_sq*
cd $R
bash tools/pmc_tile_vs_ksteps.sh gpurun_out/r03t/pmc_tile 4000000 32
python3 tools/tiled_hist_bench.py --small 2>&1 | grep -v amdgpu.ids > $OUT/in_loop_hist_config5_shard_f32.txt; echo hist1
python3 tools/tiled_hist_bench.py --members 1000000 --dtype f64 2>&1 | grep -v amdgpu.ids > $OUT/in_loop_hist_1M_f64.txt; echo hist2
ls $OUT
