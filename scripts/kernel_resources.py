#!/usr/bin/env python3
"""Compile the library for gfx950 and print VGPRs / spills / occupancy per kernel (dev tool, no GPU needed).
usage: python scripts/kernel_resources.py [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-I' + os.path.join(ROOT, 'include'),
       os.path.join(ROOT, 'krotov_amd', 'csrc', 'krotov_hip.hip'), '-o', '/tmp/_kr.so',
       '-Rpass-analysis=kernel-resource-usage'] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip().split('(')[0]
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    if 'error' in line:
        print(line)
for name, r in rows.items():
    print('%-66s VGPR %3d AGPR %3d spill %3d scratch %4d B/lane  SGPR spill %3d occ %d' % (
        name[:66], r.get('VGPRs', -1), r.get('AGPRs', -1), r.get('VGPRs Spill', -1), r.get('ScratchSize [bytes/lane]', -1),
        r.get('SGPRs Spill', -1), r.get('Occupancy [waves/SIMD]', -1)))
