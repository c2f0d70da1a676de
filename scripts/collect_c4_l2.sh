#!/bin/bash
# usage: scripts/collect_c4_l2.sh <tag>   (run on the GPU box via gpurun) -> gpurun_out/<tag>/...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r02}/c4_l2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/a -o b -- python $R/bench.py --workload c4 --no-cpu-baseline --steps 1 --warmup 1 > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/b -o b -- python $R/bench.py --workload c4 --no-cpu-baseline --steps 1 --warmup 1 > $OUT/b.log 2>&1
tail -3 $OUT/a.log; ls $OUT/a $OUT/b
