#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun) from the repo root: produces every measurement DESIGN.md quotes
# under gpurun_out/<tag>/.  Usage: bash tools/collect_profiles.sh r01
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== bench (default workload) =="
python3 $R/bench.py > $OUT/bench_config3.json 2> $OUT/bench_config3.err || echo "bench failed"
echo "== bench under rocprofv3 --kernel-trace --stats =="
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline \
    > $OUT/bench_config3_under_rocprof.json 2> $OUT/trace.err || echo "trace failed"
echo "== PMC passes (FETCH_SIZE, WRITE_SIZE separately) =="
for N in 1000000 8000000; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$N -- python3 $R/tools/pmc_workload.py $N > $OUT/pmc_fetch_$N.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$N -- python3 $R/tools/pmc_workload.py $N > $OUT/pmc_write_$N.log 2>&1
  python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch_$N $OUT/pmc_write_$N 134217728 config3:f64:$N $OUT/traffic.json > /dev/null
done
echo "== other bench lines =="
python3 $R/bench.py --no-cpu-baseline --mode fused > $OUT/bench_config3_fused.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --mode graph > $OUT/bench_config3_graph.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config2 > $OUT/bench_config2_per_step.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config2 --mode graph > $OUT/bench_config2_graph.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config2 --mode fused > $OUT/bench_config2_fused.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config4 > $OUT/bench_config4_per_gpu_shard.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --workload config5 --dtype f32 --steps 300 > $OUT/bench_config5_f32_per_gpu_shard.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --dtype f32 > $OUT/bench_config3_f32.json 2>/dev/null
echo "== sweeps =="
python3 $R/tools/sweep.py --members 100000,250000,500000,1000000,2000000,4000000 --modes per_step,fused 2>&1 | grep -v amdgpu.ids > $OUT/sweep_members.txt
python3 $R/tools/sweep.py --members 8000000 --scenario-steps 330 --modes per_step,fused 2>&1 | grep -v amdgpu.ids >> $OUT/sweep_members.txt
python3 $R/tools/sweep.py --members 1000000 --modes per_step,fused --stats 2>&1 | grep -v amdgpu.ids > $OUT/sweep_stats_on.txt
python3 $R/tools/fp32_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/fp32_vs_fp64_sweep_1M.txt
ls $OUT
