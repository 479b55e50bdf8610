#!/usr/bin/env bash
# Run ON THE GPU BOX from the repo root (via gpurun): the evidence behind profiles/r06/hbm_low_mode.txt.
#   bash tools/hbm_low_mode.sh [runs] [pmc_runs] [realloc_trials]
# Phase 1: `runs` fresh processes of tools/hbm_low_mode.py (8M-member beyond-the-cache rate + the cache-resident 1M rate + driver
#          clocks) -> low_mode_runs.jsonl;  phase 2: ONE process that re-creates the engine 12 times -> low_mode_cycles.jsonl;
# phase 3: the same tool under rocprofv3 --pmc (the program directly after `--`, --kernel-trace only, probe off), four counter
#          sets x `pmc_runs` processes x 3 engine cycles each, reduced on the box to one table per set (tools/pmc_low_mode.py).
set -u
RUNS=${1:-30}; PMC_RUNS=${2:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/low_mode_runs.jsonl
for i in $(seq 1 $RUNS); do
  python3 $R/tools/hbm_low_mode.py --config3 --tag run$i 2>/dev/null | grep '^{' >> $OUT/low_mode_runs.jsonl
  [ $((i % 5)) -eq 0 ] && echo "phase 1: $i / $RUNS"
done
python3 $R/tools/hbm_low_mode.py --cycles 12 --batches 5 --tag cycles 2>/dev/null | grep '^{' > $OUT/low_mode_cycles.jsonl
echo "phase 2 done"
export FIVEEQ_SIDE_STREAM_PROBE=0
SETA="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE GRBM_COUNT"
SETB="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE GRBM_EA_BUSY"
SETC="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
SETD="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_IB_STALL_sum GRBM_GUI_ACTIVE GRBM_TC_BUSY"
for S in A B C D; do
  eval C=\$SET$S
  for i in $(seq 1 $PMC_RUNS); do
    timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_low_$S/run$i -- python3 $R/tools/hbm_low_mode.py --cycles 3 --batches 3 --launches 40 --tag pmc$S$i \
        > $OUT/pmc_low_${S}_run$i.log 2>&1 || echo "pmc $S run $i failed"
  done
  python3 $R/tools/pmc_low_mode.py $OUT/pmc_low_$S 3 > $OUT/low_mode_pmc_$S.txt 2>&1
  grep -h '^{' $OUT/pmc_low_${S}_run*.log > $OUT/low_mode_pmc_${S}_lines.jsonl
  rm -rf $OUT/pmc_low_$S $OUT/pmc_low_${S}_run*.log
  echo "phase 3 set $S done"
done
# phase 4 (bash tools/hbm_low_mode.sh 0 0 16 runs this alone): which counters tell the slow placements of the stored-trajectory
# buffers from the fast ones — ONE process per counter set re-allocates C and T in turn under the profiler
# (tools/placement_probe.py realloc --only C,T); per trial the mean step_kernel duration and the counters per dispatch.
TRIALS=${3:-0}
if [ "$TRIALS" -gt 0 ]; then
SETE="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
SETF="TCC_WRITE_REQ_LATENCY_sum TCC_WRITE_REQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE GRBM_EA_BUSY"
SETG="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum GRBM_GUI_ACTIVE GRBM_TC_BUSY"
for S in E F G; do
  eval C=\$SET$S
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_realloc_$S -- python3 $R/tools/placement_probe.py realloc --only C,T --trials $TRIALS --batches 1 \
      > $OUT/pmc_realloc_$S.log 2>&1 || echo "pmc realloc $S failed"
  python3 $R/tools/pmc_low_mode.py $OUT/pmc_realloc_$S $TRIALS > $OUT/realloc_pmc_$S.txt 2>&1
  grep -h '^{' $OUT/pmc_realloc_$S.log > $OUT/realloc_pmc_${S}_lines.jsonl
  rm -rf $OUT/pmc_realloc_$S $OUT/pmc_realloc_$S.log
  echo "set $S done"
done
fi
ls $OUT
