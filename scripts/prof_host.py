import sys, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import krotov_amd
from krotov_amd import configs
K = int(sys.argv[1])
spec = configs.config_c5(K=K, N=64, nt=4001, L=1)
objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd)
t = {}
def hook(**kw):
    torch.cuda.synchronize()
    t[kw['iteration']] = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
res = krotov_amd.optimize_pulses(objectives, pulse_options, spec.tlist, propagator=krotov_amd.propagators.expm,
                                 chi_constructor=krotov_amd.functionals.chis_re, info_hook=hook, iter_stop=6)
pr.disable()
print('K', K, 'ms/iter', [round(1e3*(t[i+1]-t[i]),2) for i in range(1,6)])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[-2600:])
