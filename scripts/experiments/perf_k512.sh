cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "families_agree or edge_cases or hermitian_packed or (sweeps_match_oracle and k300) or (sparse_operator and (n900 or n800))" 2>&1 | tail -15
echo "--- K=512 perf: q2h (packed store) / q2h with q2 store in turns / tile256"
python scripts/perf_sweeps.py 512 64 1001 1 2>&1 | tail -2
KH_Q2H_STORE=0 python scripts/perf_sweeps.py 512 64 1001 1 2>&1 | tail -2
KH_KERNEL=tile256 python scripts/perf_sweeps.py 512 64 1001 1 2>&1 | tail -2
