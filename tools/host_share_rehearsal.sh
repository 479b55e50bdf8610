#!/bin/bash
# Host-side rehearsal of the multi-GPU run on ONE card: N ranks of the driver's workload (1M members each) share the GPU and
# exchange over gloo; what is measured is the HOST enqueue time per step with all N host threads enqueuing at once
# (timing.host_enqueue_us_per_step), against one un-shared rank's step time.  N <= 5: the GPU pool's process guard allows six processes on a card and the launcher's agent is one.
#   tools/host_share_rehearsal.sh [out_dir] [ranks...]
OUT=${1:-gpurun_out/r04}; shift
RANKS=${@:-"1 2 4 5"}
mkdir -p $OUT
export FIVEEQ_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for n in $RANKS; do
  for mode in per_step graph; do
    python3 bench.py --gpus $n --steps 20 --warmup 5 --timed-s 1.5 --no-cpu-baseline --no-hbm-resident --kernel-batches 1 \
        --mode $mode > $OUT/host_share_n${n}_${mode}.json 2> $OUT/host_share_n${n}_${mode}.err || exit 1
    python3 - $OUT/host_share_n${n}_${mode}.json <<'PY'
import json, sys
d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
t = d["timing"]
print(f"ranks {d['n_gpus']} mode {d['config']['mode']:8s} host enqueue {t['host_enqueue_us_per_step']:6.2f} us/step (min {t['host_enqueue_us_per_step_min']:6.2f})  "
      f"shared-card step {d['ms_per_step']*1e3:7.2f} us  host_share(shared card) {t['host_share']:.3f}")
PY
  done
done
