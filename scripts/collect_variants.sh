#!/bin/bash
# Run on the GPU box (via gpurun): one bench line per variant of the workload -> gpurun_out/<tag>/variants/*.json
# usage: scripts/collect_variants.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG/variants
mkdir -p $OUT
cd $R
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-config4 --no-variants --no-sparse --steps 4 --warmup 1 "$@" 2>/dev/null | tail -1 > $OUT/$name.json; cut -c1-160 $OUT/$name.json; }
run distinct --distinct
run L2 --L 2
run L3 --L 3
run L4 --L 4
# five to eight controls: the register tiles with streamed operators (kh_tile64x.h)
run L5 --L 5 --steps 3
run L6 --L 6 --steps 3
run L8 --L 8 --steps 3
run K64 --K 64
run K128 --K 128
run K512 --K 512 --steps 3
run K512_distinct --K 512 --distinct --steps 3
# more objectives than co-resident workgroups: an ensemble proper (one drift, scaled control operators: kh_ens.h) ...
run K1024 --K 1024 --steps 2
run K2048 --K 2048 --steps 2
run K4096 --K 4096 --steps 2
# ... and per-objective drifts (the streaming register-tile kernel, kh_tile64s.h), SURVEY.md 8d's second variant
run K1024_distinct --K 1024 --distinct --steps 2
run K2048_distinct --K 2048 --distinct --steps 2
run K1024_L2 --K 1024 --L 2 --steps 2
run c4_liouville_N400 --workload c4 --steps 3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats -o b -- python bench.py --workload c4 --no-cpu-baseline --steps 2 --warmup 1 > $OUT/c4_stats.log 2>&1
ls $OUT
