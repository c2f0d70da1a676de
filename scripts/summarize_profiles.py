#!/usr/bin/env python3
"""Turn gpurun_out/<tag> (scripts/collect_profiles.sh) into the committed summaries under profiles/<tag>/
and refresh profiles/pmc_latest.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_counters(dirpath):
    """{kernel name with template arguments: {counter: (average per launch, launches)}} of one rocprofv3 --pmc pass:
    from its b_counter_collection.csv, or from the b_counter_summary.json scripts/slim_counters.py left in its place."""
    out = collections.defaultdict(dict)
    js = os.path.join(dirpath, 'b_counter_summary.json')
    path = os.path.join(dirpath, 'b_counter_collection.csv')
    if os.path.exists(js):
        for kern, ctrs in json.load(open(js)).items():
            for c, v in ctrs.items():
                out[kern.split('(')[0].replace('void ', '')][c] = (v['avg_per_launch'], v['launches'])
    elif os.path.exists(path):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
        for kern, ctrs in agg.items():
            for c, v in ctrs.items():
                out[kern][c] = (sum(v) / len(v), len(v))
    return out

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles', tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, 'stats', 'b_kernel_stats.csv'), os.path.join(dst, 'kernel_stats.csv'))
shutil.copy(os.path.join(src, 'bench_n1.json'), os.path.join(dst, 'bench_n1.json'))
bench = json.loads(open(os.path.join(src, 'bench_n1.json')).read().strip())
summary = {}
for sub in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_lds'):
    for kern_full, ctrs in load_counters(os.path.join(src, sub)).items():
        kern = kern_full.split('<')[0]
        if not kern.startswith('kh_'):
            continue
        for c, (avg, n) in ctrs.items():
            prev = summary.setdefault(kern, {}).get(c)
            if prev is not None:  # (several instantiations of one kernel in the run: launch-weighted)
                avg = (prev['avg_per_launch'] * prev['launches'] + avg * n) / (prev['launches'] + n)
                n += prev['launches']
            summary[kern][c] = {'avg_per_launch': avg, 'launches': n}
json.dump(summary, open(os.path.join(dst, 'pmc_summary.json'), 'w'), indent=1, sort_keys=True)
cfg = bench['config']
sys.path.insert(0, ROOT)
import bench as _bench  # (build_id(): kh_version + hash of the kernel sources -- the record is only used by the same build)

# ONE convention for the memory-side counters, here and in bench.py (pmc_traffic / pmc_traffic_leg), DESIGN.md and commit
# messages: the JSON files hold the RAW counter values (KB, averaged per launch); bytes = (2 x FETCH_SIZE + WRITE_SIZE) x
# 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-byte requests of wide coalesced reads at 64
# bytes), WRITE_SIZE as reported.
CONVENTION = ('raw rocprofv3 counter values, KB per launch; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 '
              '(FETCH_SIZE doubled per MI355X_MICROARCH.md, gfx950)')
latest = {
    'convention': CONVENTION,
    'source': 'profiles/%s/pmc_summary.json' % tag,
    'build': bench.get('roofline', {}).get('build') or _bench.build_id(),
    'config': {'K': cfg['objectives'], 'N': cfg['N'], 'nt': cfg['time_steps'] + 1, 'L': cfg['controls']},
    'kernels': {},
}
for kern, c in summary.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        latest['kernels'][kern] = {'FETCH_SIZE_KB': c['FETCH_SIZE']['avg_per_launch'],
                                   'WRITE_SIZE_KB': c['WRITE_SIZE']['avg_per_launch']}
json.dump(latest, open(os.path.join(ROOT, 'profiles', 'pmc_latest.json'), 'w'), indent=1, sort_keys=True)
print(json.dumps(latest, indent=1))
for kern, c in summary.items():
    if 'SQ_WAVE_CYCLES' in c:
        wc = c['SQ_WAVE_CYCLES']['avg_per_launch']
        print(kern, {k: round(100 * v['avg_per_launch'] / wc, 1) for k, v in c.items() if k.startswith('SQ_') and k != 'SQ_WAVE_CYCLES'})

# variant bench lines (scripts/collect_variants.sh) -> profiles/<tag>/variants.json (+ config-4 kernel stats)
vdir = os.path.join(src, 'variants')
if os.path.isdir(vdir):
    variants = {}
    for name in sorted(os.listdir(vdir)):
        if name.endswith('.json'):
            try:
                variants[name[:-5]] = json.loads(open(os.path.join(vdir, name)).read().strip())
            except ValueError:
                variants[name[:-5]] = None
    json.dump(variants, open(os.path.join(dst, 'variants.json'), 'w'), indent=1)
    c4 = os.path.join(vdir, 'c4_stats', 'b_kernel_stats.csv')
    if os.path.exists(c4):
        shutil.copy(c4, os.path.join(dst, 'kernel_stats_config4.csv'))
    for k, v in variants.items():
        if v:
            print('%-20s %8.2f M props/s  %7.2f ms/iteration  kernel %s  (%s %.1f%% of fp64 peak)' % (
                k, v['value'] / 1e6, v['ms_per_step'], v['config']['kernel'], v['roofline']['kernel'],
                100 * v['roofline']['frac']))

# config-4 PMC passes (scripts/collect_c4_pmc.sh <tag>) -> profiles/<tag>/pmc_config4.json
cdir = os.path.join(src, 'c4_pmc')
if os.path.isdir(cdir):
    c4sum = {}
    for sub in sorted(os.listdir(cdir)):
        if not os.path.isdir(os.path.join(cdir, sub)):
            continue
        for kern_full, ctrs in load_counters(os.path.join(cdir, sub)).items():
            kern = kern_full.split('<')[0]
            if kern.startswith('kh_coop'):
                for c, (avg, n) in ctrs.items():
                    c4sum.setdefault(kern, {})[c] = {'avg_per_launch': avg, 'launches': n}
    c4sum['_build'] = _bench.build_id()
    c4sum['_convention'] = CONVENTION
    c4sum['_config'] = {'workload': 'c4', 'intervals_per_launch': 1000}
    json.dump(c4sum, open(os.path.join(dst, 'pmc_config4.json'), 'w'), indent=1, sort_keys=True)
    shutil.copy(os.path.join(dst, 'pmc_config4.json'), os.path.join(ROOT, 'profiles', 'pmc_config4_latest.json'))
    c4sum = {k: v for k, v in c4sum.items() if not k.startswith('_')}
    for kern, c in c4sum.items():
        print(kern, {k: '%.3g' % v['avg_per_launch'] for k, v in c.items()})

# PMC passes of the kernels the headline does not launch (scripts/collect_tile_pmc.sh <tag>) -> profiles/<tag>/pmc_tile.json
tdir = os.path.join(src, 'tile_pmc')
if os.path.isdir(tdir):
    tsum = {'_build': _bench.build_id(), '_convention': CONVENTION}
    for sub in sorted(os.listdir(tdir)):
        if not os.path.isdir(os.path.join(tdir, sub)):
            continue
        case = sub.rsplit('_', 1)[0]
        cmd_path = os.path.join(tdir, case + '.cmd')
        if os.path.exists(cmd_path) and case not in tsum:
            # what was profiled (scripts/collect_tile_pmc.sh writes the command line): the workload a leg's traffic is
            # scaled from (bench.py pmc_traffic_leg: bytes per INTERVAL x the leg's intervals)
            words = open(cmd_path).read().split()
            script, a = os.path.basename(words[0]), words[1:]
            if script == 'perf_sweeps.py':
                cfg_case = {'K': int(a[0]), 'N': int(a[1]), 'nt': int(a[2]), 'L': int(a[3]), 'distinct': 'distinct' in a[4:]}
            else:  # perf_sparse.py d nt K
                cfg_case = {'d': int(a[0]), 'N': int(a[0]) ** 2, 'nt': int(a[1]), 'K': int(a[2]), 'L': 1}
            tsum.setdefault(case, {})['_config'] = dict(cfg_case, command=' '.join([script] + a))
        for kern, ctrs in load_counters(os.path.join(tdir, sub)).items():
            if kern.startswith(('kh_tile', 'kh_tx_', 'kh_q2_sweep', 'kh_q2_forward', 'kh_ell', 'kh_gen', 'kh_ens_forward', 'kh_ens2_forward', 'kh_stream', 'kh_tn_sweep', 'kh_tn_forward')):
                for c, (avg, n) in ctrs.items():
                    tsum.setdefault(case, {}).setdefault(kern, {})[c] = {'avg_per_launch': avg, 'launches': n}
    json.dump(tsum, open(os.path.join(dst, 'pmc_tile.json'), 'w'), indent=1, sort_keys=True)
    shutil.copy(os.path.join(dst, 'pmc_tile.json'), os.path.join(ROOT, 'profiles', 'pmc_tile_latest.json'))  # (bench.py: the legs' traffic)
    for case, kerns in tsum.items():
        if case.startswith('_'):
            continue
        for kern, c in kerns.items():
            if kern.startswith('_'):
                continue
            if 'SQ_WAVE_CYCLES' in c:
                wc = c['SQ_WAVE_CYCLES']['avg_per_launch']
                print(case, kern, {k: round(100 * v['avg_per_launch'] / wc, 1) for k, v in c.items() if k.startswith('SQ_') and k not in ('SQ_WAVE_CYCLES',)})

# other artefacts of a round's collection, copied as they are
for name in ('ubench_gather.txt', 'config4_timing.txt', 'kernel_resources.txt', 'exp_ens.txt', 'cliffs.txt', 'busy_stream.txt',
             'stream_timing.txt', 'bench_wall.txt'):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
