#!/usr/bin/env bash
# Run in the build container after `gpurun -- bash tools/collect_profiles.sh r03` (and again after
# tools/collect_final_bench_lines.sh): copies what is judged from the scratch gpurun_out/<tag>/ into profiles/<round>/ (and the
# two counter files bench.py reads into profiles/).   Usage: bash tools/copy_collected.sh [r04|r04k]
set -u
TAG=${1:-r05}; S=gpurun_out/$TAG; D=profiles/${TAG%k}; mkdir -p $D
for f in $S/bench_*.json $S/sq_counters_*.csv; do [ -f "$f" ] && cp "$f" $D/; done
# the tables come from collect_profiles.sh only; *_ab.txt are hand-labelled A/B records: never overwritten from scratch
[ "$TAG" = "${TAG%k}" ] && for f in $S/*.txt; do case "$f" in *_ab.txt) ;; *) [ -f "$f" ] && cp "$f" $D/;; esac; done
for f in valu.json traffic.json; do [ -f $S/$f ] && cp $S/$f $D/$f && cp $S/$f profiles/$f; done
for n in fetch_1000000 write_1000000 fetch_8000000 write_8000000 fetch_fused write_fused; do
  src=$(ls -t $S/pmc_$n/*/*counter_collection.csv 2>/dev/null | head -1)
  case $n in fetch_fused) dst=pmc_fetch_fused_1000000;; write_fused) dst=pmc_write_fused_1000000;; *) dst=pmc_$n;; esac
  [ -n "$src" ] && head -40 "$src" > $D/$dst.head.csv
done
pick() { ls -t $S/$1/*/*$2 2>/dev/null | head -1; }     # the newest: gpurun merges into the scratch tree, older runs stay
k=$(pick trace _kernel_stats.csv); [ -n "$k" ] && cp "$k" $D/kernel_stats_bench_config3.csv
k=$(pick trace _domain_stats.csv); [ -n "$k" ] && cp "$k" $D/domain_stats_bench_config3.csv
k=$(pick trace_config5_demo _kernel_stats.csv); [ -n "$k" ] && cp "$k" $D/kernel_stats_config5_demo_streamed_pipeline.csv
k=$(pick trace_config2 _kernel_stats.csv); [ -n "$k" ] && cp "$k" $D/kernel_stats_bench_config2_auto.csv
ls $D | wc -l
