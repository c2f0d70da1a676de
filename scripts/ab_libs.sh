#!/bin/bash
# A/B of library builds on one box, alternating, same session (run on the GPU box):
#   bash scripts/ab_libs.sh "<perf_sweeps.py arguments>" <rounds> lib1.so lib2.so ...
# prints one perf_sweeps.py line per library and round (minimum of three sweeps each).
ARGS=${1:-"256 64 4001 1"}; ROUNDS=${2:-3}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    echo -n "$(basename $lib) | "
    KH_LIB=$R/$lib python $R/scripts/perf_sweeps.py $ARGS 2>&1 | grep -v amdgpu.ids | head -1
  done
done
