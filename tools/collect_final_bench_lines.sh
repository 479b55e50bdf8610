#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun) from the repo root, AFTER `tools/collect_profiles.sh <tag> quick` has produced the
# counter files and they were copied to profiles/: re-collects the fp32 SQ passes (instructions, packed share), merges
# them into profiles/valu.json, then regenerates EVERY bench line and the kernel trace under gpurun_out/<tag>k/, so that no
# committed bench line was produced with an older valu.json / traffic.json than the one committed beside it.
# rocprofv3 is always given `python3 script` directly after `--`; --pmc passes carry --kernel-trace only.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${TAG}k; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cp $R/profiles/valu.json $OUT/valu.json
SQA="SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
SQC="SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_CVT"
for S in a b c; do eval C=\$SQ$(echo $S | tr a-z A-Z); timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_sq${S}_fused32 -- python3 $R/tools/pmc_workload_fused.py 4000000 f32 96 > $OUT/pmc_sq${S}_fused32.log 2>&1; echo "fused32 $S"; done
python3 $R/tools/pmc_valu.py $OUT/sq_counters_f32_4M.csv $OUT/valu.json 96 $OUT/pmc_sqa_fused32 $OUT/pmc_sqb_fused32 $OUT/pmc_sqc_fused32 > /dev/null
for S in a c; do eval C=\$SQ$(echo $S | tr a-z A-Z); timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_sq${S}_fused32c -- python3 $R/tools/pmc_workload_fused.py 4000000 f32 96 multigas comp > $OUT/pmc_sq${S}_fused32c.log 2>&1; echo "fused32 comp $S"; done
python3 $R/tools/pmc_valu.py $OUT/sq_counters_f32_4M_compensated.csv $OUT/valu.json 96 $OUT/pmc_sqa_fused32c $OUT/pmc_sqc_fused32c > /dev/null
for S in a c; do eval C=\$SQ$(echo $S | tr a-z A-Z); timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_sq${S}_step32 -- python3 $R/tools/pmc_workload.py 4000000 multigas f32 > $OUT/pmc_sq${S}_step32.log 2>&1; echo "step32 $S"; done
python3 $R/tools/pmc_valu.py $OUT/sq_counters_step_f32_4M.csv $OUT/valu.json 1 $OUT/pmc_sqa_step32 $OUT/pmc_sqc_step32 > /dev/null
for S in a b; do eval C=\$SQ$(echo $S | tr a-z A-Z); timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_sq${S}_small -- python3 $R/tools/pmc_workload_fused.py 10000 f64 750 co2 small > $OUT/pmc_sq${S}_small.log 2>&1; echo "small $S"; done
python3 $R/tools/pmc_valu.py $OUT/sq_counters_small_co2_f64_10k.csv $OUT/valu.json 750 $OUT/pmc_sqa_small $OUT/pmc_sqb_small > /dev/null
timeout -k 10 300 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/pmc_sqa_small3 -- python3 $R/tools/pmc_workload_fused.py 10000 f64 750 multigas small > $OUT/pmc_sqa_small3.log 2>&1
python3 $R/tools/pmc_valu.py $OUT/sq_counters_small_multigas_f64_10k.csv $OUT/valu.json 750 $OUT/pmc_sqa_small3 > /dev/null
cp $OUT/valu.json $R/profiles/valu.json
cd $R
python3 bench.py --no-cpu-baseline --mode fused > $OUT/bench_config3_fused.json 2>/dev/null
python3 bench.py --no-cpu-baseline --mode graph > $OUT/bench_config3_graph.json 2>/dev/null; echo bench4
python3 bench.py --no-cpu-baseline --workload config2 > $OUT/bench_config2_per_step.json 2>/dev/null
python3 bench.py --no-cpu-baseline --workload config2 --mode graph > $OUT/bench_config2_graph.json 2>/dev/null
python3 bench.py --no-cpu-baseline --workload config2 --mode auto > $OUT/bench_config2_auto.json 2>/dev/null
python3 bench.py --no-cpu-baseline --workload config2 --mode ksteps > $OUT/bench_config2_ksteps.json 2>/dev/null
python3 bench.py --no-cpu-baseline --workload config2 --mode fused > $OUT/bench_config2_fused.json 2>/dev/null; echo bench8
python3 bench.py --no-cpu-baseline --workload config4 > $OUT/bench_config4_per_gpu_shard.json 2>/dev/null
python3 bench.py --no-cpu-baseline --workload config5 --dtype f32 --steps 300 > $OUT/bench_config5_f32_per_gpu_shard.json 2>/dev/null
python3 bench.py --no-cpu-baseline --dtype f32 > $OUT/bench_config3_f32.json 2>/dev/null
python3 bench.py --no-cpu-baseline --dtype f32 --mode fused > $OUT/bench_config3_f32_fused.json 2>/dev/null
python3 bench.py --no-cpu-baseline --workload config5 --dtype f32 --mode fused --no-trajectory > $OUT/bench_config5_f32_fused_no_trajectory.json 2>/dev/null; echo bench13
python3 bench.py --no-cpu-baseline --workload config5 --dtype f32 --mode fused --no-trajectory --compensated > $OUT/bench_config5_f32_fused_no_trajectory_compensated.json 2>/dev/null
python3 bench.py --no-cpu-baseline --members 10000 --mode auto > $OUT/bench_multigas_10k_auto.json 2>/dev/null      # three gases at config 2's size: the octet form
python3 bench.py --no-cpu-baseline --members 8000 --mode auto > $OUT/bench_multigas_8k_auto.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-hbm-resident --workload config5 --dtype f32 --members 100000000 --no-trajectory > $OUT/bench_config5_whole_on_one_gpu_f32_per_step.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-hbm-resident --workload config5 --dtype f32 --members 100000000 --no-trajectory --mode fused > $OUT/bench_config5_whole_on_one_gpu_f32_fused.json 2>/dev/null; echo bench15
# the default line last, right before its twin under rocprofv3
python3 bench.py > $OUT/bench_config3.json 2> $OUT/bench_config3.err; echo bench1
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config3_driver_call_20_steps.json 2>/dev/null
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-hbm-resident --no-live-traffic > $OUT/bench_config3_under_rocprof.json 2> $OUT/trace.err
for k in kernel_stats domain_stats; do src=$(ls -t $OUT/trace/*/*_$k.csv 2>/dev/null | head -1); [ -n "$src" ] && cp "$src" $OUT/${k}_bench_config3.csv; done
rm -rf $OUT/pmc_sq*_fused32 $OUT/pmc_sq*_fused32c $OUT/pmc_sq*_step32 $OUT/pmc_sq*_small* $OUT/trace
ls $OUT
