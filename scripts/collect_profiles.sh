#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel stats + separate PMC passes.
# usage: scripts/collect_profiles.sh <tag>   -> gpurun_out/<tag>/...
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T0=$(date +%s); timeout 900 python $R/bench.py 2>/dev/null | tail -1 > $OUT/bench_n1.json; echo "default bench.py run: $(( $(date +%s) - T0 )) s wall" > $OUT/bench_wall.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $R/bench.py --no-cpu-baseline --no-config4 --no-variants --no-sparse > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o b -- python $R/bench.py --no-cpu-baseline --no-config4 --no-variants --no-sparse --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o b -- python $R/bench.py --no-cpu-baseline --no-config4 --no-variants --no-sparse --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o b -- python $R/bench.py --no-cpu-baseline --no-config4 --no-variants --no-sparse --steps 2 --warmup 1 > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $OUT/pmc_lds -o b -- python $R/bench.py --no-cpu-baseline --no-config4 --no-variants --no-sparse --steps 2 --warmup 1 > $OUT/pmc_lds.log 2>&1
cut -c1-400 $OUT/bench_n1.json
ls $OUT
