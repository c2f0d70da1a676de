#!/bin/bash
# Ensemble kernel (kh_ens.h) against the families it replaces: update / backward sweep per interval at engine level.
# usage: bash scripts/exp_ens.sh ["K list"] [nt]
KS=${1:-"512 1024 2048 4096"}; NT=${2:-1001}
for K in $KS; do
  for E in 1 0; do
    KH_ENS=$E timeout 300 python scripts/perf_sweeps.py $K 64 $NT 1 2>&1 | tail -2 | sed "s/^/KH_ENS=$E /"
  done
done
