#!/bin/bash
# Config 4 traffic accounting, second part: experiment builds that skip the per-interval table reads (wrong results,
# right counters): bash scripts/collect_c4_libs.sh <tag> lib1.so lib2.so ...  -> gpurun_out/<tag>/c4_ab/<lib>/...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG/c4_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  mkdir -p $OUT/$name
  export KH_LIB=$R/$lib
  for rep in 1 2; do
    echo -n "$name | " >> $OUT/timings.txt
    python $R/scripts/perf_c4.py 2>&1 | grep -v amdgpu.ids | head -1 >> $OUT/timings.txt
  done
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/$name/$ctr -o b -- python $R/scripts/perf_c4.py > $OUT/$name/$ctr.log 2>&1
  done
done
unset KH_LIB
python $R/scripts/slim_counters.py $OUT
cat $OUT/timings.txt
