#!/bin/bash
# SQ counters of the packed time-tiled kernel against the packed fused kernel relaunched every K steps (same state round trip):
#   bash tools/pmc_tile_vs_ksteps.sh <outdir> [members] [K]
# run from the repo root on the GPU box; writes <outdir>/tile_vs_ksteps_packed.csv (tools/pmc_reduce.py format).
OUT=$1; N=${2:-4000000}; K=${3:-32}
R=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQA="SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
SQD="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_WAVES"
SQE="SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"
SQF="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS SQ_WAVES"
i=0
for set in "$SQA" "$SQB" "$SQD" "$SQE" "$SQF"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pass$i -- python3 $R/tools/pmc_workload_tile.py $N $K 96 f32 1 > $R/$OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 tools/pmc_reduce.py K$K $OUT/pass1 $OUT/pass2 $OUT/pass3 $OUT/pass4 $OUT/pass5 > $OUT/tile_vs_ksteps_packed.csv
rm -rf $OUT/pass[1-5]
