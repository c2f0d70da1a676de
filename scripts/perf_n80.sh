#!/bin/bash
# K objectives of N = 80 / 96 / 128 with per-objective operators: the register-generator kernels against the generic ones
cd ${GRAFT_REPO_ROOT:-.}
for N in 80 96 128; do
  python scripts/perf_sweeps.py 256 $N 201 1 2>&1 | tail -2
  KH_KERNEL=generic python scripts/perf_sweeps.py 256 $N 201 1 2>&1 | tail -2
done
python scripts/perf_sweeps.py 256 80 201 2 2>&1 | tail -2
KH_KERNEL=generic python scripts/perf_sweeps.py 256 80 201 2 2>&1 | tail -2
