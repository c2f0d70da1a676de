#!/bin/bash
# Update-sweep time of every experiment build build/libkh_x_*.so (KH_MM_X_* switches of kh_tile64mm.h: results of
# all but "base" are wrong, only their timing is of interest).  Run on the GPU box from the repo root.
for lib in build/libkh_x_*.so; do
  printf "%-34s" "$lib"
  KH_TIMING_LIB=$PWD/$lib KH_TIMEOUT_MS=3000 timeout 120 python scripts/timing_update.py 2>&1 | grep "update" | sed 's/.*update/update/; s/;.*//'
done
