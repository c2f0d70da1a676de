#!/bin/bash
# Run on the GPU box (via gpurun): PMC passes (SQ_*, FETCH_SIZE, WRITE_SIZE) of the streaming update kernel, K = 1024
# -> gpurun_out/r04/tile_pmc/K1024_<set>/ (summarised into profiles/<tag>/pmc_tile.json by summarize_profiles.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04/tile_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"
for set in sq fetch write; do
  case $set in sq) C="$SQ";; fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; esac
  for v in ""; do
    name=K1024
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${name}_$set -o b -- python $R/scripts/perf_sweeps.py 1024 64 501 1 > $OUT/${name}_$set.log 2>&1
  done
done
grep -h "kh_stream" $OUT/K1024_fetch/b_counter_collection.csv | head -2
tail -2 $OUT/K1024_sq.log
