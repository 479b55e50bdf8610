#!/usr/bin/env bash
# Rehearsal of the multi-GPU run on ONE card (run ON THE GPU BOX via gpurun, from the repo root) — round 6, VERDICT r05 item 1.
# The pool's process guard allows SIX processes on a card, so "eight ranks on one card" cannot run: the ranks-as-processes legs
# stop at five (+ the launcher's agent), and eight enqueuing host threads are rehearsed as 4 processes x 2 threads and as 8 threads
# of one process.  Everything goes through gloo (ranks sharing a card cannot use RCCL).
#   (1) bench.py --gpus N for N = 1 2 4 5, per-step and graph: the line, the per-rank CPU sets, the host enqueue share
#   (2) tools/host_threads_rehearsal.py: 1 / 2 / 4 / 8 threads of one process; 4 processes x 2 threads behind a common barrier
#   (3) BASELINE configs[3] END TO END AT FULL SIZE: example/run_sharded.py, 5 ranks x 2M = 10M members drawn on the device, the
#       exchange over gloo, its CSV checked against ONE engine's 10M-member run + np.percentile, bit for bit
OUT=${1:-gpurun_out/r06}
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2
if [ "$PART" = "all" ]; then
echo "== (1) bench.py --gpus N over gloo ==" | tee $OUT/host_share_rehearsal.txt
export FIVEEQ_BENCH_BACKEND=gloo
for n in 1 2 4 5; do
  for mode in per_step graph; do
    python3 bench.py --gpus $n --steps 20 --warmup 5 --timed-s 1.5 --no-cpu-baseline --no-hbm-resident --no-live-traffic --kernel-batches 1 \
        --mode $mode > $OUT/host_share_n${n}_${mode}.json 2> $OUT/host_share_n${n}_${mode}.err || exit 1
    python3 - $OUT/host_share_n${n}_${mode}.json <<'PY' | tee -a $OUT/host_share_rehearsal.txt
import json, sys
d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
t = d["timing"]
print(f"ranks {d['n_gpus']} mode {d['config']['mode']:8s} host enqueue {t['host_enqueue_us_per_step']:6.2f} us/step (min {t['host_enqueue_us_per_step_min']:6.2f})  "
      f"shared-card step {d['ms_per_step']*1e3:7.2f} us  host_share(shared card) {t['host_share']:.3f}")
PY
  done
done
python3 - $OUT >> $OUT/host_share_rehearsal.txt <<'PY'
import json, sys
for n in (2, 4, 5):
    d = json.loads([ln for ln in open(f"{sys.argv[1]}/host_share_n{n}_per_step.json") if ln.startswith("{")][-1])
    print(f"ranks {n}: pids {[x['pid'] for x in d['config']['devices']]}")
    print(f"         cpus {[x['cpus'] for x in d['config']['devices']]}")
    print(f"         side streams probed/passed {[(x['side_streams']['probed'], x['side_streams']['passed']) for x in d['config']['devices']]}")
    print(f"         per-rank host enqueue us/step {[round(v, 2) for v in d['timing']['per_rank_host_enqueue_us']]}")
PY
echo "== (2) host threads ==" | tee -a $OUT/host_share_rehearsal.txt
python3 tools/host_threads_rehearsal.py 1 2 4 8 2>/dev/null | tee -a $OUT/host_share_rehearsal.txt
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29541 tools/host_threads_rehearsal.py 1 2 2>/dev/null | tee -a $OUT/host_share_rehearsal.txt
fi
echo "== (3) config 4 end to end: 5 ranks x 2M members, gloo ==" | tee $OUT/config4_end_to_end.txt
T0=$(date +%s.%N)
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=5 --master-addr 127.0.0.1 --master-port 29542 example/run_sharded.py \
    --members 10000000 --backend gloo --mode per_step --out $OUT/config4_summary.csv 2> $OUT/config4_run.err || { echo "run_sharded failed"; tail -5 $OUT/config4_run.err; exit 1; }
echo "whole job (launcher start to exit, 5 ranks on one card): $(python3 -c "import time,sys; print(round(time.time() - float(sys.argv[1]), 1))" $T0) s" | tee -a $OUT/config4_end_to_end.txt
cat $OUT/config4_summary.csv | tee -a $OUT/config4_end_to_end.txt
python3 tools/check_config4_csv.py $OUT/config4_summary.csv 10000000 249,499,749 2>/dev/null | tee -a $OUT/config4_end_to_end.txt
