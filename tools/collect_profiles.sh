#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun) from the repo root: produces every measurement DESIGN.md quotes
# under gpurun_out/<tag>/.  Usage: bash tools/collect_profiles.sh r06 [quick|rest|sweeps]
# (quick = the default bench line, the kernel trace and every PMC pass; rest = the other bench lines and the sweeps)
# rocprofv3 is always given the program itself after `--` (python3 script), never a shell or env wrapper, and the
# --pmc passes carry --kernel-trace only (no sys/hip/hsa trace domains).
set -u
TAG=${1:-r06}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$QUICK" != "rest" ] && [ "$QUICK" != "sweeps" ]; then
echo "== bench (default workload) =="
python3 $R/bench.py > $OUT/bench_config3.json 2> $OUT/bench_config3.err || echo "bench failed"
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config3_driver_call_20_steps.json 2>/dev/null || echo "bench (20 steps) failed"
echo "== bench under rocprofv3 --kernel-trace --stats =="
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-hbm-resident --no-live-traffic \
    > $OUT/bench_config3_under_rocprof.json 2> $OUT/trace.err || echo "trace failed"
echo "== PMC passes: HBM traffic (FETCH_SIZE, WRITE_SIZE separately) =="
for N in 1000000 8000000; do
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$N -- python3 $R/tools/pmc_workload.py $N > $OUT/pmc_fetch_$N.log 2>&1
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$N -- python3 $R/tools/pmc_workload.py $N > $OUT/pmc_write_$N.log 2>&1
  python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch_$N $OUT/pmc_write_$N 134217728 config3:f64:$N $OUT/traffic.json > /dev/null
done
echo "== PMC passes: SQ counters (VALU issue) of the per-step, fused and small-ensemble kernels =="
SQA="SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/pmc_sqa_step -- python3 $R/tools/pmc_workload.py 1000000 > $OUT/pmc_sqa_step.log 2>&1
timeout -k 10 400 rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $OUT/pmc_sqb_step -- python3 $R/tools/pmc_workload.py 1000000 > $OUT/pmc_sqb_step.log 2>&1
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/pmc_sqa_fused64 -- python3 $R/tools/pmc_workload_fused.py 1000000 f64 96 > $OUT/pmc_sqa_fused64.log 2>&1
timeout -k 10 400 rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $OUT/pmc_sqb_fused64 -- python3 $R/tools/pmc_workload_fused.py 1000000 f64 96 > $OUT/pmc_sqb_fused64.log 2>&1
python3 $R/tools/pmc_valu.py $OUT/sq_counters_f64_1M.csv $OUT/valu.json 96 $OUT/pmc_sqa_step $OUT/pmc_sqb_step $OUT/pmc_sqa_fused64 $OUT/pmc_sqb_fused64 > /dev/null
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/pmc_sqa_fused32 -- python3 $R/tools/pmc_workload_fused.py 4000000 f32 96 > $OUT/pmc_sqa_fused32.log 2>&1
timeout -k 10 400 rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $OUT/pmc_sqb_fused32 -- python3 $R/tools/pmc_workload_fused.py 4000000 f32 96 > $OUT/pmc_sqb_fused32.log 2>&1
SQC="SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_CVT"
timeout -k 10 400 rocprofv3 --pmc $SQC --kernel-trace --output-format csv -d $OUT/pmc_sqc_fused32 -- python3 $R/tools/pmc_workload_fused.py 4000000 f32 96 > $OUT/pmc_sqc_fused32.log 2>&1
python3 $R/tools/pmc_valu.py $OUT/sq_counters_f32_4M.csv $OUT/valu.json 96 $OUT/pmc_sqa_fused32 $OUT/pmc_sqb_fused32 $OUT/pmc_sqc_fused32 > /dev/null
# the compensated fp32 form of the fused kernel (round 6): its own instruction stream
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/pmc_sqa_fused32c -- python3 $R/tools/pmc_workload_fused.py 4000000 f32 96 multigas comp > $OUT/pmc_sqa_fused32c.log 2>&1
timeout -k 10 400 rocprofv3 --pmc $SQC --kernel-trace --output-format csv -d $OUT/pmc_sqc_fused32c -- python3 $R/tools/pmc_workload_fused.py 4000000 f32 96 multigas comp > $OUT/pmc_sqc_fused32c.log 2>&1
python3 $R/tools/pmc_valu.py $OUT/sq_counters_f32_4M_compensated.csv $OUT/valu.json 96 $OUT/pmc_sqa_fused32c $OUT/pmc_sqc_fused32c > /dev/null
# the other kernels bench.py can be asked to price: fp32 per-step (config 5), CO2-only (config 2)
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/pmc_sqa_step32 -- python3 $R/tools/pmc_workload.py 4000000 multigas f32 > $OUT/pmc_sqa_step32.log 2>&1
timeout -k 10 400 rocprofv3 --pmc $SQC --kernel-trace --output-format csv -d $OUT/pmc_sqc_step32 -- python3 $R/tools/pmc_workload.py 4000000 multigas f32 > $OUT/pmc_sqc_step32.log 2>&1
python3 $R/tools/pmc_valu.py $OUT/sq_counters_step_f32_4M.csv $OUT/valu.json 1 $OUT/pmc_sqa_step32 $OUT/pmc_sqc_step32 > /dev/null
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/pmc_sqa_co2 -- python3 $R/tools/pmc_workload.py 1000000 co2 > $OUT/pmc_sqa_co2.log 2>&1
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/pmc_sqa_co2_fused -- python3 $R/tools/pmc_workload_fused.py 1000000 f64 96 co2 > $OUT/pmc_sqa_co2_fused.log 2>&1
python3 $R/tools/pmc_valu.py $OUT/sq_counters_co2_f64_1M.csv $OUT/valu.json 96 $OUT/pmc_sqa_co2 $OUT/pmc_sqa_co2_fused > /dev/null
# BASELINE configs[1]: the small-ensemble kernel (4 and 1 lanes per member) and the fused kernel at 10k CO2-only members
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/pmc_sqa_small -- python3 $R/tools/pmc_workload_fused.py 10000 f64 750 co2 small > $OUT/pmc_sqa_small.log 2>&1
timeout -k 10 400 rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $OUT/pmc_sqb_small -- python3 $R/tools/pmc_workload_fused.py 10000 f64 750 co2 small > $OUT/pmc_sqb_small.log 2>&1
python3 $R/tools/pmc_valu.py $OUT/sq_counters_small_co2_f64_10k.csv $OUT/valu.json 750 $OUT/pmc_sqa_small $OUT/pmc_sqb_small > /dev/null
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/pmc_sqa_small3 -- python3 $R/tools/pmc_workload_fused.py 10000 f64 750 multigas small > $OUT/pmc_sqa_small3.log 2>&1
python3 $R/tools/pmc_valu.py $OUT/sq_counters_small_multigas_f64_10k.csv $OUT/valu.json 750 $OUT/pmc_sqa_small3 > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_config2 -- python3 $R/bench.py --workload config2 --mode auto --no-cpu-baseline \
    > $OUT/bench_config2_auto_under_rocprof.json 2> $OUT/trace_config2.err || echo "config2 trace failed"
echo "== the streamed histogram pipeline under --kernel-trace --stats; LDS conflicts of the histogram kernels; fused traffic =="
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_config5_demo -- python3 $R/tools/config5_demo.py > $OUT/trace_config5_demo.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_lds_hist -- python3 $R/tools/pmc_workload_fused.py 4000000 f32 96 > $OUT/pmc_lds_hist.log 2>&1
python3 $R/tools/pmc_valu.py $OUT/sq_counters_lds_f32_4M.csv $OUT/lds_valu_scratch.json 96 $OUT/pmc_lds_hist > /dev/null 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_fused -- python3 $R/tools/pmc_workload_fused_traffic.py > $OUT/pmc_fetch_fused.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_fused -- python3 $R/tools/pmc_workload_fused_traffic.py > $OUT/pmc_write_fused.log 2>&1
python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch_fused $OUT/pmc_write_fused 134217728 fused:f64:1000000 $OUT/traffic.json fused > /dev/null
# keep what is judged, drop the raw per-dispatch tables: gpurun brings back at most 64 MiB
for n in fetch_1000000 write_1000000 fetch_8000000 write_8000000 fetch_fused write_fused; do
  src=$(ls -t $OUT/pmc_$n/*/*counter_collection.csv 2>/dev/null | head -1)
  case $n in fetch_fused) dst=pmc_fetch_fused_1000000;; write_fused) dst=pmc_write_fused_1000000;; *) dst=pmc_$n;; esac
  [ -n "$src" ] && head -40 "$src" > $OUT/$dst.head.csv
done
for pair in trace:bench_config3 trace_config2:bench_config2_auto trace_config5_demo:config5_demo_streamed_pipeline; do
  t=${pair%%:*}; name=${pair##*:}
  for k in kernel_stats domain_stats; do
    src=$(ls -t $OUT/$t/*/*_$k.csv 2>/dev/null | head -1); [ -n "$src" ] && cp "$src" $OUT/${k}_$name.csv
  done
done
for d in $OUT/pmc_* $OUT/trace $OUT/trace_config2 $OUT/trace_config5_demo; do [ -d "$d" ] && rm -rf "$d"; done
fi
if [ "$QUICK" != "quick" ]; then
if [ "$QUICK" != "sweeps" ]; then          # (sweeps = the tables only: tools/collect_final_bench_lines.sh regenerates every bench line)
echo "== other bench lines =="
python3 $R/bench.py --no-cpu-baseline --mode fused > $OUT/bench_config3_fused.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --mode graph > $OUT/bench_config3_graph.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config2 > $OUT/bench_config2_per_step.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config2 --mode graph > $OUT/bench_config2_graph.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config2 --mode auto > $OUT/bench_config2_auto.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config2 --mode ksteps > $OUT/bench_config2_ksteps.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config2 --mode fused > $OUT/bench_config2_fused.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config4 > $OUT/bench_config4_per_gpu_shard.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config5 --dtype f32 --steps 300 > $OUT/bench_config5_f32_per_gpu_shard.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --dtype f32 > $OUT/bench_config3_f32.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --dtype f32 --mode fused > $OUT/bench_config3_f32_fused.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config5 --dtype f32 --mode fused --no-trajectory > $OUT/bench_config5_f32_fused_no_trajectory.json 2>/dev/null
fi
echo "== sweeps =="
python3 $R/tools/sweep.py --members 100000,250000,500000,1000000,2000000,4000000 --modes per_step,fused 2>&1 | grep -v amdgpu.ids > $OUT/sweep_members.txt
python3 $R/tools/sweep.py --members 8000000 --scenario-steps 330 --modes per_step,fused 2>&1 | grep -v amdgpu.ids >> $OUT/sweep_members.txt
python3 $R/tools/fp32_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/fp32_vs_fp64_sweep_1M.txt
python3 $R/tools/hist_forms_bench.py --small 2>&1 | grep -v amdgpu.ids > $OUT/in_loop_hist_config5_shard_f32.txt
python3 $R/tools/hist_forms_bench.py --members 1000000 --dtype f64 2>&1 | grep -v amdgpu.ids > $OUT/in_loop_hist_1M_f64.txt
python3 $R/tools/small_ensemble_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/small_ensemble_ab.txt
python3 $R/tools/config5_demo.py 2>&1 | grep -v amdgpu.ids > $OUT/config5_shard_end_to_end.txt
python3 $R/tools/packed_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/packed_ab.txt
python3 $R/tools/summary_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/summary_timing.txt
[ -x $R/tools/microbench/hbm_rates ] && $R/tools/microbench/hbm_rates > $OUT/hbm_rates.txt 2>&1     # (built in the container: hipcc --offload-arch=gfx950 -O3)
fi
ls $OUT
