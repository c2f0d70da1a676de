#!/bin/bash
# Run on the GPU box (via gpurun): PMC passes for the one-term-per-phase register-tile kernels (several controls, K = 512)
# and the sparse kernels -- the kernels the headline passes of collect_profiles.sh do not launch.
# usage: scripts/collect_tile_pmc.sh <tag>  -> gpurun_out/<tag>/tile_pmc/<case>_<set>/
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG/tile_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"
LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64"
run() {  # name, perf script + arguments
    name=$1; shift
    timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $OUT/${name}_sq -o b -- python "$@" > $OUT/${name}_sq.log 2>&1
    timeout 300 rocprofv3 --pmc $LDS --output-format csv -d $OUT/${name}_lds -o b -- python "$@" > $OUT/${name}_lds.log 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${name}_fetch -o b -- python "$@" > $OUT/${name}_fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${name}_write -o b -- python "$@" > $OUT/${name}_write.log 2>&1
    tail -2 $OUT/${name}_sq.log
}
run L2 $R/scripts/perf_sweeps.py 256 64 1001 2
run L4 $R/scripts/perf_sweeps.py 256 64 1001 4
run K512 $R/scripts/perf_sweeps.py 512 64 1001 1
run sparse $R/scripts/perf_sparse.py 25 501 16
ls $OUT
