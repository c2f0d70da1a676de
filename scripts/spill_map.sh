#!/bin/bash
# Where a kernel's scratch (spill) traffic sits relative to its barriers / sleeps (dev tool, no GPU needed).
# usage: scripts/spill_map.sh <mangled-name-prefix> [extra hipcc flags]
R=$(cd "$(dirname "$0")/.." && pwd)
K=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -S --cuda-device-only $R/krotov_amd/csrc/krotov_hip.hip -o /tmp/_sm.s "$@" 2>/dev/null
S=$(grep -n "^$K" /tmp/_sm.s | head -1 | cut -d: -f1)
E=$(awk -v s=$S 'NR>s && /^.Lfunc_end/{print NR; exit}' /tmp/_sm.s)
sed -n "${S},${E}p" /tmp/_sm.s > /tmp/_sm_kernel.s
grep -n "scratch_\|s_barrier\|s_sleep\|v_mfma" /tmp/_sm_kernel.s | awk -F: '{print $1": "$2}' | cut -c1-60 | \
  awk '{ if ($2 ~ /scratch_store/) st++; else if ($2 ~ /scratch_load/) ld++; else { if (st||ld) print "   [" st " stores, " ld " loads]"; st=0; ld=0; print } }'
echo "kernel ISA in /tmp/_sm_kernel.s"
