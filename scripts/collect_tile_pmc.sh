#!/bin/bash
# Run on the GPU box (via gpurun): PMC passes (separate runs per counter set, as MI355X_MICROARCH.md prescribes) for the
# kernels the headline passes of collect_profiles.sh do not launch: several controls, K = 512, the ensemble kernel and
# the streaming kernel beyond the co-resident limit (shared / per-objective drifts), N = 96, sparse operators.
# usage: scripts/collect_tile_pmc.sh <tag> ["case list"]  -> gpurun_out/<tag>/tile_pmc/<case>_<set>/
# (summarised into profiles/<tag>/pmc_tile.json by scripts/summarize_profiles.py; the case names are bench.py's leg names)
set -u
TAG=${1:-r05}
CASES=${2:-"L2 L4 L8 K512 K1024 K1024_distinct K2048_distinct N96 sparse sparse_n1600"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG/tile_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"
LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64"
run() {  # name, perf script + arguments
    name=$1; shift
    echo "$*" > $OUT/${name}.cmd
    timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $OUT/${name}_sq -o b -- python "$@" > $OUT/${name}_sq.log 2>&1
    timeout 300 rocprofv3 --pmc $LDS --output-format csv -d $OUT/${name}_lds -o b -- python "$@" > $OUT/${name}_lds.log 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${name}_fetch -o b -- python "$@" > $OUT/${name}_fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${name}_write -o b -- python "$@" > $OUT/${name}_write.log 2>&1
    tail -2 $OUT/${name}_sq.log
}
for c in $CASES; do
  case $c in
    L2) run L2 $R/scripts/perf_sweeps.py 256 64 1001 2;;
    L4) run L4 $R/scripts/perf_sweeps.py 256 64 1001 4;;
    L8) run L8 $R/scripts/perf_sweeps.py 256 64 501 8;;
    K512) run K512 $R/scripts/perf_sweeps.py 512 64 1001 1;;
    K1024) run K1024 $R/scripts/perf_sweeps.py 1024 64 501 1;;
    K1024_distinct) run K1024_distinct $R/scripts/perf_sweeps.py 1024 64 501 1 distinct;;
    K2048_distinct) run K2048_distinct $R/scripts/perf_sweeps.py 2048 64 251 1 distinct;;
    N96) run N96 $R/scripts/perf_sweeps.py 256 96 501 1;;
    sparse) run sparse $R/scripts/perf_sparse.py 25 501 16;;
    sparse_n1600) run sparse_n1600 $R/scripts/perf_sparse.py 40 101 3 csr;;
  esac
done
ls $OUT
