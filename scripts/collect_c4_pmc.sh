#!/bin/bash
# usage: scripts/collect_c4_pmc.sh <tag>   (run on the GPU box via gpurun) -> gpurun_out/<tag>/...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r02}/c4_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -o b -- python $R/bench.py --workload c4 --no-cpu-baseline --steps 1 --warmup 1 > $OUT/$tag.log 2>&1
done
ls $OUT/*
