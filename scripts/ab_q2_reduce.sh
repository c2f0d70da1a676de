#!/bin/bash
# A/B on the GPU box: q2 kernel variants selected by -D flags (each argument is one set of flags, "" = default)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude krotov_amd/csrc/krotov_hip.hip -o krotov_amd/libkrotov_hip.so"
for rep in 1 2; do
  for flags in "$@"; do
    $B $flags && echo "[$flags]" && timeout 120 python scripts/perf_sweeps.py 256 64 4001 1 2>&1 | grep backward
  done
done
$B
