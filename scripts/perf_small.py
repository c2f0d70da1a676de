#!/usr/bin/env python3
"""Iteration time of the small BASELINE configurations (1: TLS, 2: X gate Hilbert/Liouville, 3: iSWAP)
through optimize_pulses on the device path (dev tool, GPU only)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import krotov_amd
from krotov_amd import configs
from krotov_amd.engine import LAST_ENGINE

for name, spec in (('c1 TLS', configs.config_c1()), ('c2 X gate (Hilbert)', configs.config_c2_hilbert()),
                   ('c2 X gate (Liouville, 3 states)', configs.config_c2_liouville()), ('c3 iSWAP', configs.config_c3())):
    objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd)
    prop = krotov_amd.propagators.HipExpm(liouville=True) if spec.is_super else krotov_amd.propagators.expm
    stamps = []

    def hook(**kw):
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())

    os.environ['KH_PROFILE'] = '1'  # per-sweep HIP-event timing
    krotov_amd.optimize_pulses(objectives, pulse_options, spec.tlist, propagator=prop,
                               chi_constructor=getattr(krotov_amd.functionals, 'chis_' + spec.chi),
                               info_hook=hook, iter_stop=8)
    per = sorted(b - a for a, b in zip(stamps[2:], stamps[3:]))
    eng = LAST_ENGINE()
    t = eng.kernel_times_ms()
    print('%-34s K=%d N=%d nt=%d  kernel %-13s  %.2f ms per iteration (kernels: backward %.2f + update %.2f ms = '
          '%.2f + %.2f us per interval)' % (
              name, spec.K, spec.N, len(spec.tlist), eng.kernel, 1e3 * per[len(per) // 2], min(t['backward']),
              min(t['update']), 1e3 * min(t['backward']) / (len(spec.tlist) - 1),
              1e3 * min(t['update']) / (len(spec.tlist) - 1)))
