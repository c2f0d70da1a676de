#!/bin/bash
# Config 4: where do the memory-side bytes of the cooperative sweeps come from?  (run on the GPU box)
#   bash scripts/collect_c4_ab.sh <tag>
# A/B passes of FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc runs) for: the default placement (one column group per
# XCD, ring of 4 blocks in its L2), KH_COOP_XCD=0 (column groups spread over the XCDs, ring of 32 blocks across them),
# and the timing of both.  -> gpurun_out/<tag>/c4_ab/<variant>/<counter>/..., timings.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r06}/c4_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for variant in xcd1 xcd0; do
  if [ $variant = xcd0 ]; then export KH_COOP_XCD=0; else unset KH_COOP_XCD; fi
  for rep in 1 2; do
    echo -n "$variant | " >> $OUT/timings.txt
    python $R/scripts/perf_c4.py 2>&1 | grep -v amdgpu.ids | head -1 >> $OUT/timings.txt
  done
  mkdir -p $OUT/$variant
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/$variant/$ctr -o b -- python $R/scripts/perf_c4.py > $OUT/$variant/$ctr.log 2>&1
  done
done
unset KH_COOP_XCD
python $R/scripts/slim_counters.py $OUT
cat $OUT/timings.txt
