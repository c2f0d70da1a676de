#!/bin/bash
# Per-interval breakdown of the q2 update sweep (KH_TIMING build) for the environment/flag variants given as
# arguments, e.g.  scripts/ab_q2_timing.sh "KH_NO_ADJ=1" ""
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude krotov_amd/csrc/krotov_hip.hip -DKH_TIMING -o gpurun_out/libkrotov_hip_timing.so
for v in "$@"; do echo "[$v]"; env $v timeout 120 python scripts/timing_update.py 2>&1 | tail -1; done
rm -f gpurun_out/libkrotov_hip_timing.so
