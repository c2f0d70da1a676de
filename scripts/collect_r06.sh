#!/bin/bash
# Round 6's measurement campaign, in two parts so that each fits one gpurun call (run on the GPU box):
#   bash scripts/collect_r06.sh a   headline bench line + rocprofv3 kernel stats + PMC passes (collect_profiles.sh), variants
#   bash scripts/collect_r06.sh b   PMC passes of the side kernels (collect_tile_pmc.sh)
#   bash scripts/collect_r06.sh c   PMC passes of config 4, timing-build breakdowns, micro-benchmarks, the cliffs
# then here: python scripts/summarize_profiles.py r06   -> profiles/r06/, profiles/pmc_*latest.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd $R
case ${1:-a} in
a)
  bash scripts/collect_profiles.sh r06
  bash scripts/collect_variants.sh r06
  ;;
b)
  bash scripts/collect_tile_pmc.sh r06
  ;;
c)
  bash scripts/collect_c4_pmc.sh r06
  hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather.hip -o /tmp/ubench_gather 2>/dev/null && /tmp/ubench_gather > $OUT/ubench_gather.txt 2>&1
  bash scripts/exp_ens.sh "512 1024 2048 4096" 1001 2>&1 | grep -v amdgpu.ids > $OUT/exp_ens.txt
  # the cliffs outside the register families: per-objective operators beyond N = 128, more than four controls
  for a in "256 160 101 1" "64 256 101 1" "256 64 501 5" "256 64 501 6" "256 64 501 7" "256 64 501 8" "256 100 201 6" "256 96 501 2" "256 128 201 1"; do
    timeout 600 python scripts/perf_sweeps.py $a 2>&1 | grep -v amdgpu.ids >> $OUT/cliffs.txt
  done
  # config 4: where a round of the cooperative kernels goes (timing build)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKH_TIMING -Iinclude krotov_amd/csrc/krotov_hip.hip -o gpurun_out/libkrotov_hip_timing.so 2>/dev/null
  python scripts/timing_coop.py 2>&1 | grep -v amdgpu.ids > $OUT/config4_timing.txt
  python scripts/timing_stream.py 1024 1 --distinct 2>&1 | grep -v amdgpu.ids > $OUT/stream_timing.txt
  rm -f gpurun_out/libkrotov_hip_timing.so
  # the retried update sweep next to a busy stream, with its timing line; more controls than the kernels take
  python -m pytest tests/test_hip_parity.py -q -s -k "busy_stream or more_controls" 2>&1 | grep -E "update sweep:|passed|failed" > $OUT/busy_stream.txt
  ;;
esac
# gpurun copies at most 64 MiB back: per-dispatch counter tables -> per-kernel averages
python scripts/slim_counters.py $OUT
du -sh $OUT
ls $OUT
