#!/usr/bin/env bash
# Run in the build container after `gpurun -- bash tools/collect_profiles.sh r06` (and again after
# tools/collect_final_bench_lines.sh): copies what is judged from the scratch gpurun_out/<tag>/ into profiles/<round>/ (and the
# two counter files bench.py reads into profiles/).   Usage: bash tools/copy_collected.sh [r06|r06k]
set -u
TAG=${1:-r06}; S=gpurun_out/$TAG; D=profiles/${TAG%k}; mkdir -p $D
for f in $S/bench_*.json $S/sq_counters_*.csv $S/pmc_*.head.csv $S/kernel_stats_*.csv $S/domain_stats_*.csv; do [ -f "$f" ] && cp "$f" $D/; done
# the tables come from collect_profiles.sh only; *_ab.txt are hand-labelled A/B records: never overwritten from scratch
[ "$TAG" = "${TAG%k}" ] && for f in $S/*.txt; do case "$f" in *_ab.txt) ;; *) [ -f "$f" ] && cp "$f" $D/;; esac; done
for f in valu.json traffic.json; do [ -f $S/$f ] && cp $S/$f $D/$f && cp $S/$f profiles/$f; done
ls $D | wc -l
