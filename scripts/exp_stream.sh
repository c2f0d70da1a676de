#!/bin/bash
# Streaming update kernel (kh_tile64s.h): time per interval against the workgroups in flight, with / without the
# Hermitian half fetch, and (one control) two workgroups per CU instead of the tile prefetch (run on the GPU box).
# usage: bash scripts/exp_stream.sh [K] [L] ["G list"] ["HERM list"] ["TWO list"]
K=${1:-1024}; L=${2:-1}; GS=${3:-"256 128"}; HS=${4:-"1 0"}; TS=${5:-"0"}
for T in $TS; do for H in $HS; do for G in $GS; do
  KH_STREAM_TWO=$T KH_STREAM_HERM=$H KH_STREAM_G=$G timeout 200 python bench.py $EXTRA --K $K --L $L --nt 1001 --steps 3 --warmup 1 --no-cpu-baseline --no-config4 --no-variants --no-sparse 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('K $K L $L G $G herm $H two $T', d['config'].get('kernel'), 'update %.2f ms  backward %.2f ms' % (d['kernels']['update_sweep_ms'], d['kernels']['backward_sweep_ms']))"
done; done; done
