#!/usr/bin/env python3
"""Run on the GPU box at the end of a collection: every rocprofv3 b_counter_collection.csv under the given directory
(per-dispatch rows: tens of MiB for sweeps with many small launches) -> b_counter_summary.json next to it (kernel name
-> counter -> {avg_per_launch, launches}; values summed over the dispatch's rows first, as rocprofv3 writes one row per
dimension instance), and the CSV removed if it is larger than 256 KiB -- gpurun only copies 64 MiB back.
usage: python scripts/slim_counters.py gpurun_out/r05"""
import collections
import csv
import json
import os
import sys


def summarise(path):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        key = (r.get('Dispatch_Id') or r.get('Correlation_Id'), r['Counter_Name'])
        per_dispatch[key] += float(r['Counter_Value'])
        names[key[0]] = r['Kernel_Name']
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (disp, counter), v in per_dispatch.items():
        agg[names[disp]][counter].append(v)
    return {k: {c: {'avg_per_launch': sum(v) / len(v), 'launches': len(v)} for c, v in cs.items()} for k, cs in agg.items()}


if __name__ == '__main__':
    root = sys.argv[1]
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith('_counter_collection.csv'):
                p = os.path.join(dirpath, f)
                json.dump(summarise(p), open(os.path.join(dirpath, 'b_counter_summary.json'), 'w'), indent=0)
                if os.path.getsize(p) > 256 * 1024:
                    os.unlink(p)
            elif f.endswith(('_agent_info.csv',)) or (f.endswith('.csv') and 'trace' in f and os.path.getsize(os.path.join(dirpath, f)) > 1 << 20):
                os.unlink(os.path.join(dirpath, f))
