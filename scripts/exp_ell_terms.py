import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ['KH_PROFILE'] = '1'
import bench
for cap in ('4', '6'):
    os.environ['KH_ELL_CAP'] = cap
    r = bench.sparse_leg()
    nt = 1999
    print('cap', cap, 'bw %.2f us/interval  up %.2f us/interval  terms(up) %.1f' % (
        r['kernels']['backward_sweep_ms'] * 1e3 / nt, r['kernels']['update_sweep_ms'] * 1e3 / nt, r['terms_per_step_update_sweep']))
