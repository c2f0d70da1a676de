#!/usr/bin/env python3
"""Benchmark of the Krotov hot path on MI355X (the driver's contract).

    python bench.py --gpus N --steps K --warmup W

A "step" is one Krotov iteration of the BASELINE.json headline configuration
(robustness ensemble: 256 objectives IN TOTAL, Hilbert dimension 64, 4000 time
steps, L=1 control, complex128) through ``krotov_amd.optimize_pulses`` with
``propagator=krotov_amd.propagators.expm``: chi construction -> backward sweep
storing chi(t_n) -> forward sweep with sequential pulse update -> tau, exactly
the bracket the reference times per iteration (optimize.py:396 -> 510).  Inputs
are synthetic (``krotov_amd.configs.config_c5``) and resident in HBM before the
timed region.  One JSON line is printed by rank 0.

For N > 1 there is one rank per GPU.  Either a launcher provides them
(``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or -- plain
``python bench.py --gpus N`` with no WORLD_SIZE -- this file starts the N ranks
itself (``torch.distributed.run`` on 127.0.0.1 with a free port) and exits with
their return code.  The headline line keeps the per-GPU work fixed (``"scaling":
"weak"``, what the driver's contract asks of a path that shards over independent
units): 256 objectives PER GPU -- at N = 1 BASELINE config 5 itself --, the L update
sums crossing the GPUs once per time interval inside the persistent kernels
(peer-mapped windows over xGMI).  The same JSON line carries
* ``"strong"``: BASELINE config 5 to the letter on N GPUs, 256 objectives IN TOTAL
  sharded over the ranks, 256/N per GPU (``--scaling strong`` makes that one the
  headline instead; the sweeps are bound by the latency of a time step, not by the
  number of objectives per GPU, so this one does not get faster with N: DESIGN.md 4),
* ``"rccl"``: the strong-scaling job again with the north star's transport, one
  RCCL all-reduce of the L sums per time interval (``KH_P2P=0``; HIP-graph replay
  of the interval loop), a few iterations,
* ``"config4"`` (N = 2, 4, 8): three iterations of BASELINE config 4 -- quoted on 2 and 4
  GPUs -- with its 16 density matrices sharded over the ranks (cooperative matrix-core
  kernels, update sums through the peer windows),
* ``"n_ranks_seen"``: ``torch.distributed.get_world_size()``.
Each of these side measurements runs under a watchdog (``--rccl-leg-timeout``): if one
hangs or fails, the headline line is printed with an ``"error"`` entry in its place.
At N = 1 the line carries instead, measured after the headline and not part of
``value``: ``"config4"`` (three iterations of BASELINE config 4, 16 density
matrices under one 400-dim Liouvillian: the cooperative fp64 matrix-core
kernels; ``--no-config4`` skips it, ``--workload c4`` makes it the line itself)
and the two variants SURVEY.md 8d asks for, ``"L4"`` (four controls) and
``"distinct"`` (256 distinct random drifts), three iterations each, ``"L8"`` (eight controls)
(``--no-variants`` skips them).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector FMA peak == fp64 MFMA peak (256 CU x 128 flop/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0
TAYLOR_DEGREE = 14       # degree credited by the roofline formula (theta <= 0.5, 2^-53)


def algorithmic_flops(K, N, nt, L):
    """F_prop = 8 N^2 m, F_upd = 8 N^2 + 8 N per (k, l, step)."""
    f_prop = 8.0 * N * N * TAYLOR_DEGREE
    f_upd = 8.0 * N * N + 8.0 * N
    backward = K * (nt - 1) * f_prop
    update = K * (nt - 1) * (f_prop + L * f_upd)
    return backward, update


def algorithmic_bytes(K, N, nt, L):
    """HBM bytes per launch: chi written once (backward) / read once (update),
    operators read once per sweep, pulses and shapes."""
    chi = 16.0 * K * N * (nt - 1)
    ops = 16.0 * K * (1 + L) * N * N
    return chi + ops + 8.0 * L * (nt - 1), chi + ops + 3 * 8.0 * L * (nt - 1)


def build_id():
    """What the kernels were built from: kh_version() + the first 16 hex digits of the SHA-256 over the kernel sources
    (krotov_amd/csrc/*, sorted by name).  A git hash would do the same job but the GPU box has no .git."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, 'krotov_amd', 'csrc', '*'))):
        with open(path, 'rb') as fh:
            h.update(os.path.basename(path).encode() + b'\0' + fh.read())
    try:
        from krotov_amd import _lib

        version = _lib.load().kh_version().decode().split(' (')[0]
    except Exception:
        version = 'krotov_hip ?'
    return '%s %s' % (version, h.hexdigest()[:16])


def pmc_traffic(kernel, K_loc, args):
    """(HBM bytes per launch of `kernel`, where they come from) from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json, written by scripts/collect_profiles.sh + summarize_profiles.py: separate FETCH_SIZE and
    WRITE_SIZE passes, KB units, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  The record is
    keyed on the workload AND on the build (build_id()): counters of another build of the kernels do not ride along --
    (None, why) then."""
    path = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    try:
        rec = json.load(open(path))
    except Exception:
        return None, 'no profiles/pmc_latest.json'
    cfg = rec.get('config', {})
    if (cfg.get('K'), cfg.get('N'), cfg.get('nt'), cfg.get('L')) != (K_loc, args.N, args.nt, args.L):
        return None, 'profiles/pmc_latest.json holds another workload'
    if rec.get('build') != build_id():
        return None, 'profiles/pmc_latest.json is from another build of the kernels (%s; this one: %s): not used' % (
            rec.get('build'), build_id())
    k = rec.get('kernels', {}).get(kernel)
    if not k:
        return None, 'profiles/pmc_latest.json has no record of %s' % kernel
    return ((2.0 * k['FETCH_SIZE_KB'] + k['WRITE_SIZE_KB']) * 1024.0,
            '%s: rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE) of an earlier run of this command on THIS build '
            '(%s), not measured in this run' % (rec.get('source', 'profiles/pmc_latest.json'), rec.get('build')))


def pmc_traffic_leg(case, kernel, intervals):
    """The same for a side measurement: (HBM bytes per launch, source) of `kernel` (name without template arguments) from
    the committed PMC passes of the case `case` (profiles/pmc_tile_latest.json: scripts/collect_tile_pmc.sh +
    summarize_profiles.py; 'config4': profiles/pmc_config4_latest.json), under the same rules as the headline's -- one
    convention for the counters, keyed on the build -- and scaled from the profiled run's intervals per launch to this
    leg's (every kernel here streams per interval; the operators read once per sweep are < 2 % of that)."""
    name = 'pmc_config4_latest.json' if case == 'config4' else 'pmc_tile_latest.json'
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', name)))
    except Exception:
        return None, 'no profiles/%s' % name
    if rec.get('_build') != build_id():
        return None, 'profiles/%s is from another build of the kernels (%s; this one: %s): not used' % (name, rec.get('_build'), build_id())
    kerns = rec if case == 'config4' else rec.get(case)
    if not kerns:
        return None, 'profiles/%s has no case %s' % (name, case)
    cfg = kerns.get('_config', {})
    ran = cfg.get('intervals_per_launch') or (cfg.get('nt', 0) - 1)
    best = None
    for kname, c in kerns.items():
        if kname.split('<')[0] == kernel and 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            b = (2.0 * c['FETCH_SIZE']['avg_per_launch'] + c['WRITE_SIZE']['avg_per_launch']) * 1024.0
            best = b if best is None else max(best, b)
    if best is None or ran <= 0:
        return None, 'profiles/%s, case %s: no FETCH_SIZE / WRITE_SIZE record of %s' % (name, case, kernel)
    return best * intervals / ran, ('profiles/%s case %s (%s): rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE) on THIS build, %d '
                                    'intervals per launch there, scaled to %d' % (name, case, cfg.get('command', 'bench.py --workload c4'), ran, intervals))


def cpu_baseline(args):
    """Oracle in reference-structured mode on a bounded sample of the workload."""
    from krotov_amd import configs
    from oracle import cpu_baseline as cb

    cores = len(os.sched_getaffinity(0))
    # calibrate: time of one dense expm @ state at this N on this host (a first short run warms the worker's
    # imports and caches up; the second, longer one is the measurement)
    cb.timed_iteration(configs.config_c5(K=1, N=args.N, nt=21, L=args.L), processes=1)
    r = cb.timed_iteration(configs.config_c5(K=1, N=args.N, nt=401, L=args.L), processes=1)
    t_prop = r['seconds'] / r['props']
    # how many worker processes?  The reference would use every core (parallel_map); its per-interval
    # synchronisation makes that slower, not faster, beyond some count on a large box -- short pilots pick the count
    # that gives THIS baseline its best throughput (all visible cores, half of them (SMT), 64)
    if args.cpu_procs > 0:
        candidates = [min(cores, args.cpu_procs, args.K)]
    else:
        candidates = sorted({max(1, min(c, args.K, cores)) for c in (cores, cores // 2, 64)}, reverse=True)
    pilots = {}
    for P in candidates:
        K_p = min(args.K, 2 * P)
        rp = cb.timed_iteration(configs.config_c5(K=K_p, N=args.N, nt=121, L=args.L, distinct=args.distinct), processes=P)
        pilots[P] = rp['props'] / rp['seconds']
    P = max(pilots, key=pilots.get)
    K_s = min(args.K, 2 * P)
    per_proc = (K_s + P - 1) // P
    t_in_sample = P / pilots[P]  # seconds per propagation per process, synchronisation included
    nt_s = int(min(args.nt - 1, max(30, args.cpu_seconds / (2 * per_proc * t_in_sample))))
    spec = configs.config_c5(K=K_s, N=args.N, nt=nt_s + 1, L=args.L, distinct=args.distinct)
    r = cb.timed_iteration(spec, processes=P)
    return {
        'value': r['props'] / r['seconds'],
        'unit': 'state*timestep props/s',
        'cores': r['processes'],
        'kind': 'port',
        'sample': 'one Krotov iteration of the same ensemble restricted to K=%d objectives x %d intervals '
                  '(dt unchanged), NumPy oracle in reference-structured mode: dense expm per objective per '
                  'step (SciPy if present), 1 BLAS thread per process, %d processes; %.1f s' % (
                      K_s, nt_s, r['processes'], r['seconds']),
        'backward_seconds': r['backward_seconds'],
        'update_seconds': r['update_seconds'],
        'backward_props_per_s': 0.5 * r['props'] / r['backward_seconds'],
        'update_props_per_s': 0.5 * r['props'] / r['update_seconds'],
        'seconds_per_prop_single_core': t_prop,
        'seconds_per_prop_in_the_sample': r['seconds'] * r['processes'] / r['props'],
        'host_cores_visible': cores,
        'pilot_props_per_s_by_process_count': {str(k): v for k, v in pilots.items()},
        'cpu_model': _cpu_model(),
        # the REAL reference (krotov.optimize_pulses, serial_map, numpy mode) on this very workload at full size: wall
        # time recorded by the committed fixture generator when it produced tests/golden/ref_c5_full.npz -- iteration 0
        # (one forward propagation) + one Krotov iteration = 3 sweeps of K x (nt-1) propagations, single process, in the
        # BUILD container (not on this box): the true baseline next to the generous one above
        'reference_loop': _reference_loop_record(args),
        'expm': 'scipy.linalg.expm %s' % _scipy_version() if _scipy_version() else "oracle's own Pade-13 (NumPy)",
        'note': 'per-core cost in the sample vs alone: what P processes sharing the memory system and one '
                'synchronisation per time interval (the reference\'s parallel_map_fw_prop_step structure) cost on '
                'top of the arithmetic (backward_props_per_s: the sweep with no per-interval synchronisation; '
                'update_props_per_s: the one with it); scripts/cpu_probe.py prints the pieces on the box',
    }


def _reference_loop_record(args):
    if (args.K, args.N, args.nt, args.L) != (256, 64, 4001, 1) or args.distinct:
        return None
    try:
        import numpy as np

        secs = float(np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_c5_full.npz'))['seconds'])
    except Exception:
        return None
    props = 256 * 4000 * 3
    return {'props_per_s': props / secs, 'seconds': secs, 'sweeps': 3, 'processes': 1,
            'cpu_model': 'Intel(R) Xeon(R) Processor @ 2.10GHz (build container)',
            'source': 'tests/golden/make_reference_goldens.py c5full -> ref_c5_full.npz["seconds"]'}


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def _scipy_version():
    try:
        import scipy

        return scipy.__version__
    except Exception:
        return None


def sparse_leg(steps=3, warmup=1):
    """The sparse-operator row of the path on the reference's own workload for it: notebook 06
    (docs/notebooks/06_example_3states.ipynb: two coupled 5-level transmons in Liouville space, 625-dim Liouvillian with
    4.8-6.4 entries per row, 3 weighted density matrices, 2 controls, 2000 grid points, propagator=
    DensityMatrixODEPropagator), operators and controls from tests/golden/dump_3states.npz, through
    ``optimize_pulses``; plus the engine-level sweep times of the 16-density-matrix ladder scripts/perf_sparse.py has
    tracked since round 1.  Roofline: a term of the series is one gather of 16 bytes and one complex multiply-add per
    matrix entry -- bound by the LDS gather (128 B/clk per CU); credited per propagation: m_alg = 14 terms x nnz."""
    import numpy as np
    import scipy.sparse as sp
    import torch

    import krotov_amd
    from krotov_amd import configs, engine as _engine_mod

    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'dump_3states.npz'))
    N = int(g['N'])
    Lops = [sp.csr_matrix((g['L%d_data' % i], g['L%d_indices' % i], g['L%d_indptr' % i]), shape=(N, N)) for i in range(3)]
    ctrls = [np.array(c, dtype=np.float64) for c in g['controls_it3']]
    T = g['tlist'][-1]
    S = lambda t: krotov_amd.shapes.flattop(t, 0.0, T, float(g['t_rise']))  # noqa: E731
    objs = []
    for k in range(3):
        obj = krotov_amd.Objective(initial_state=g['rho0'][k], target=g['rho_tgt'][k],
                                   H=[Lops[0], [Lops[1], ctrls[0]], [Lops[2], ctrls[1]]])
        obj.weight = float(g['weights'][k])
        objs.append(obj)
    opts = {id(c): dict(lambda_a=float(g['lambda_a']), update_shape=S) for c in ctrls}
    marks, n_iter = {}, warmup + steps

    def hook(**kw):
        if kw['iteration'] == warmup:
            _engine_mod.LAST_ENGINE().kernel_times_ms(reset=True)
            torch.cuda.synchronize()
            marks['t0'] = time.perf_counter()
        elif kw['iteration'] == n_iter:
            torch.cuda.synchronize()
            marks['t1'] = time.perf_counter()

    krotov_amd.optimize_pulses(objs, opts, g['tlist'], propagator=krotov_amd.propagators.DensityMatrixODEPropagator(),
                               chi_constructor=krotov_amd.functionals.chis_re, info_hook=hook, iter_stop=n_iter)
    eng = _engine_mod.LAST_ENGINE()
    times = eng.kernel_times_ms(reset=True)
    stats = eng.stats()
    K, nt = 3, len(g['tlist'])
    elapsed = marks['t1'] - marks['t0']
    nnz_union = int((abs(Lops[0]) + abs(Lops[1]) + abs(Lops[2])).nnz)
    t_up = float(np.mean(times['update'][-steps:])) * 1e-3
    t_bw = float(np.mean(times['backward'][-steps:])) * 1e-3
    terms_per_step = stats['matvecs'] / (K * (nt - 1))  # (incl. the L control products of the update sums)
    # LDS read rate of one CU per MI355X_MICROARCH.md ("64 dwords (256 B) wide per clock") x 2.4 GHz, over the K CUs this
    # job can occupy; next to it the whole chip's (256 CUs) and the gather rate measured with the kernels' own access
    # pattern (scripts/ubench_gather.hip -> profiles/<round>/ubench_gather.txt, committed)
    lds_peak_cu = 256.0 * 2.4  # GB/s
    lds_peak = K * lds_peak_cu
    credited_bytes = K * (nt - 1) * TAYLOR_DEGREE * nnz_union * 16.0
    measured_gather = None
    try:
        import glob

        for line in open(sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'ubench_gather.txt')))[-1]):
            if line.startswith('band') and 'E= 8 gather+fma' in line and '256 workgroups' in line:
                measured_gather = float(line.split('workgroups')[1].split('GB/s')[0])
    except Exception:
        pass
    rec = {
        'workload': "reference notebook 06 (3states): K=3 density matrices, N=625 sparse Liouvillian (%d entries on the union "
                    "pattern, %.1f per row), L=2, %d time steps, chis_re with weights; propagator=DensityMatrixODEPropagator"
                    % (nnz_union, nnz_union / N, nt - 1),
        'kernel': eng.kernel, 'steps': steps, 'ms_per_step': elapsed / steps * 1e3,
        'value': K * (nt - 1) * 2 * steps / elapsed, 'unit': 'props/s',
        'us_per_propagation': elapsed / steps / (K * (nt - 1) * 2) * 1e6,
        'kernels': {'backward_sweep_ms': t_bw * 1e3, 'update_sweep_ms': t_up * 1e3},
        'terms_per_step_update_sweep': terms_per_step,
        'us_per_term': t_up / ((nt - 1) * terms_per_step) * 1e6,
        'roofline': {'bound': 'lds-gather', 'kernel': 'kh_ell_forward_update',
                     'achieved': credited_bytes / t_up / 1e9, 'peak': lds_peak, 'unit': 'GB/s',
                     'frac': credited_bytes / t_up / 1e9 / lds_peak,
                     'chip_peak': 256 * lds_peak_cu, 'chip_frac': credited_bytes / t_up / 1e9 / (256 * lds_peak_cu),
                     'measured_gather_peak_per_cu': measured_gather,
                     'frac_of_measured_gather': None if not measured_gather else credited_bytes / t_up / 1e9 / (K * measured_gather),
                     'note': 'credited: 14 terms x nnz x 16 B of LDS gather per propagation (SURVEY.md 8d: m = 14), over the '
                             'LDS read rate of the K = 3 CUs the job can use (256 B/clk x 2.4 GHz each, MI355X_MICROARCH.md); '
                             'chip_frac: over all 256 CUs -- 253 idle by construction, the chain of terms is serial and one '
                             'objective cannot be spread over CUs without a cross-CU exchange per term; '
                             'measured_gather_peak_per_cu: scripts/ubench_gather.hip, the kernels\' access pattern with '
                             'nothing but the gathers and their FMAs'},
        'reference_recorded': 'the reference ran this optimisation at 23 s per iteration (12 h 53 min for 2000, notebook cell 55)',
    }
    # the 16-density-matrix ladder of scripts/perf_sparse.py (N = 625, 5.8 entries per row, 500 intervals), engine level
    spec = configs.config_sparse_lindblad(d=25, nt=501, K=16)
    tl = spec.tlist
    pulses = np.array([[spec.controls[0](t + 0.5 * (tl[1] - tl[0]), None) for t in tl[:-1]]])
    e2 = _engine_mod.HipKrotovEngine(configs.sparse_ops(spec), np.diff(tl), is_super=True)
    e2.profile = True
    chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
    for _ in range(3):
        chi = e2.backward(chi_T, pulses)
        e2.forward_update(chi, np.full(16, 1.0 / 32), spec.init, pulses, np.ones((1, 500)), np.full(1, 2.0))
    e2.check()
    t2 = e2.kernel_times_ms()
    rec['ladder_16x625'] = {'kernel': e2.kernel, 'backward_sweep_ms': min(t2['backward']), 'update_sweep_ms': min(t2['update']),
                            'us_per_propagation': (min(t2['backward']) + min(t2['update'])) * 1e3 / (16 * 500 * 2),
                            'round_3': '202 / 221 ms per sweep, 25 us per propagation (generic CSR kernels)'}
    e2.close()
    tr, src = pmc_traffic_leg('sparse', 'kh_ell_forward_update', 500)
    rec['ladder_16x625']['traffic_update_sweep'] = tr
    rec['ladder_16x625']['traffic_source'] = src
    # beyond 1024 rows: a d = 40 ladder (N = 1600, 5.9 entries per row), four rows per lane of 512 threads -- and the
    # generic CSR kernels the same problem ran on before
    spec = configs.config_sparse_lindblad(d=40, nt=101, K=3)
    tl = spec.tlist
    pulses = np.array([[spec.controls[0](t + 0.5 * (tl[1] - tl[0]), None) for t in tl[:-1]]])
    chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
    lad = {}
    for label, env in (('ell', None), ('generic', 'generic')):
        saved = os.environ.get('KH_KERNEL')
        if env is not None:
            os.environ['KH_KERNEL'] = env
        try:
            e3 = _engine_mod.HipKrotovEngine(configs.sparse_ops(spec), np.diff(tl), is_super=True)
        finally:
            if env is not None:
                if saved is None:
                    os.environ.pop('KH_KERNEL', None)
                else:
                    os.environ['KH_KERNEL'] = saved
        e3.profile = True
        for _ in range(2):
            chi = e3.backward(chi_T, pulses)
            e3.forward_update(chi, np.full(3, 1.0 / 6), spec.init, pulses, np.ones((1, 100)), np.full(1, 2.0))
        e3.check()
        t3 = e3.kernel_times_ms()
        lad[label] = {'kernel': e3.kernel, 'backward_sweep_ms': min(t3['backward']), 'update_sweep_ms': min(t3['update']),
                      'us_per_propagation': (min(t3['backward']) + min(t3['update'])) * 1e3 / (3 * 100 * 2)}
        e3.close()
    lad['speedup'] = lad['generic']['us_per_propagation'] / lad['ell']['us_per_propagation']
    rec['ladder_3x1600'] = lad
    # ... and beyond what registers (N <= 2048) and the generic kernels' LDS vectors (N <= 2540) hold: d = 64, N = 4096 on the
    # streamed form of the sparse kernels (rows read from the pools per term)
    spec = configs.config_sparse_lindblad(d=64, nt=51, K=3)
    tl = spec.tlist
    pulses = np.array([[spec.controls[0](t + 0.5 * (tl[1] - tl[0]), None) for t in tl[:-1]]])
    chi_T = spec.target / np.linalg.norm(spec.target, axis=1)[:, None]
    e4 = _engine_mod.HipKrotovEngine(configs.sparse_ops(spec), np.diff(tl), is_super=True)
    e4.profile = True
    for _ in range(2):
        chi = e4.backward(chi_T, pulses)
        e4.forward_update(chi, np.full(3, 1.0 / 6), spec.init, pulses, np.ones((1, 50)), np.full(1, 2.0))
    e4.check()
    t4 = e4.kernel_times_ms()
    rec['ladder_3x4096'] = {'kernel': e4.kernel, 'backward_sweep_ms': min(t4['backward']), 'update_sweep_ms': min(t4['update']),
                            'us_per_propagation': (min(t4['backward']) + min(t4['update'])) * 1e3 / (3 * 50 * 2),
                            'terms_per_step': e4.stats()['matvecs'] / (3 * 50)}
    e4.close()
    return rec


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: run this very command line under
    ``torch.distributed.run`` (one rank per GPU, rendezvous on 127.0.0.1, a free port) and return its exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--K', type=int, default=256, help='objectives (per GPU: weak scaling; in total: strong scaling)')
    ap.add_argument('--N', type=int, default=64)
    ap.add_argument('--nt', type=int, default=4001)
    ap.add_argument('--L', type=int, default=1)
    ap.add_argument('--distinct', action='store_true', help='every objective gets its own random drift')
    ap.add_argument('--workload', choices=['c5', 'c4'], default='c5',
                    help="c5 (default): BASELINE config 5, the configuration the metric is quoted on; c4: config 4 "
                         "(transmon Liouvillian, N=400, 16 density matrices sharing one operator list), a "
                         "single-GPU variant line for profiles/, not the headline")
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help='weak (default): --K objectives per GPU; strong: --K objectives in total over all GPUs = BASELINE '
                         'config 5 to the letter.  The other one is measured too and reported under "strong" / "weak" of '
                         'the same line (at one GPU they are the same job)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-config4', action='store_true',
                    help='skip the short run of BASELINE config 4 (the one matrix-core workload) that the default '
                         'single-GPU line carries under "config4"')
    ap.add_argument('--no-variants', action='store_true',
                    help='skip the short L=4 and distinct-drift runs the default single-GPU line carries under "L4" / "distinct"')
    ap.add_argument('--no-sparse', action='store_true',
                    help='skip the sparse-operator leg the default single-GPU line carries under "sparse" (the reference\'s '
                         'notebook 06 through DensityMatrixODEPropagator)')
    ap.add_argument('--no-rccl-leg', action='store_true',
                    help='multi-rank runs: skip the extra measurement with one RCCL all-reduce per time interval ("rccl")')
    ap.add_argument('--rccl-leg-timeout', type=int, default=240, help='seconds the "rccl" side measurement may take')
    ap.add_argument('--force-dist', action='store_true',
                    help='run the multi-GPU code path (stepwise sweep + RCCL all-reduce per interval) even on 1 rank')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--cpu-procs', type=int, default=0, help='worker processes of the CPU baseline (0: every visible core)')
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # no launcher: start one rank per GPU ourselves and hand their verdict on (rank 0 prints the JSON line)
        sys.exit(self_launch(args.gpus))
    if world != args.gpus and rank == 0:
        print('bench.py: --gpus %d but WORLD_SIZE=%d; measuring on %d rank(s)' % (args.gpus, world, world),
              file=sys.stderr)
    n_dev = max(1, torch.cuda.device_count())
    local_rank = local_rank % n_dev  # (testing: several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    group = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')

        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # RCCL refuses two ranks on one device: with fewer GPUs than ranks (a test set-up) the host-side
        # collectives go through gloo; the in-kernel peer windows do not care
        backend = os.environ.get('KH_DIST_BACKEND', 'nccl' if n_dev >= world else 'gloo')
        if n_dev < world:
            # ... and the cooperative kernels must not pin their column groups to XCDs: every rank would claim the SAME
            # XCDs (group y -> XCD y) of the one GPU, more workgroups than those XCDs have CUs, and ranks that wait for
            # each other's sums inside their kernels would never all be resident (one rank per GPU: no such conflict)
            os.environ.setdefault('KH_COOP_XCD', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    os.environ['KH_PROFILE'] = '1'
    import krotov_amd
    from krotov_amd import configs, engine as _engine_mod

    def measure(scaling):
        K_total = args.K * world if scaling == 'weak' else args.K
        propagator = krotov_amd.propagators.expm
        if args.workload == 'c4':
            spec = configs.config_c4()
            K_total, args.K, args.N, args.nt, args.L = spec.K, spec.K, spec.N, len(spec.tlist), spec.L
            args.no_cpu_baseline = True
            propagator = krotov_amd.propagators.HipExpm(liouville=True)
        else:
            spec = configs.config_c5(K=K_total, N=args.N, nt=args.nt, L=args.L, distinct=args.distinct)
        objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd)

        n_iter = args.warmup + args.steps
        marks = {}

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()

        stamps = []

        def hook(**kw):
            it = kw['iteration']
            if os.environ.get('KH_BENCH_DEBUG'):
                torch.cuda.synchronize()
                stamps.append((it, time.perf_counter()))
            if it == args.warmup:
                eng = _engine_mod.LAST_ENGINE()
                if eng is not None:
                    eng.kernel_times_ms(reset=True)
                gc.collect()  # set-up garbage (K objectives, nested lists) is collected before, not inside, the timed steps
                barrier()
                marks['t0'] = time.perf_counter()
            elif it == n_iter:
                barrier()
                marks['t1'] = time.perf_counter()
            return None

        res = krotov_amd.optimize_pulses(
            objectives, pulse_options, spec.tlist,
            propagator=propagator,
            chi_constructor=krotov_amd.functionals.chis_re,
            info_hook=hook, iter_stop=n_iter, process_group=group,
        )
        elapsed = marks['t1'] - marks['t0']
        if stamps and rank == 0:
            print('per-iteration ms:', [round(1e3 * (b[1] - a[1]), 2) for a, b in zip(stamps, stamps[1:])], file=sys.stderr)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        eng = _engine_mod.LAST_ENGINE()
        times = eng.kernel_times_ms(reset=True)
        stats = eng.stats()
        kernel_names = {
            'tile64q2/512': ('kh_q2_forward_update', 'kh_q2_sweep_store'),
            'mini16/wave': ('kh_mini_forward_update', 'kh_mini_sweep_store'),
            'mini4/wave': ('kh_quad_forward_update', 'kh_quad_sweep_store'),
            'generic': ('kh_gen_forward_update', 'kh_gen_sweep_store'),
            'coop16/mfma': ('kh_coop_forward_update', 'kh_coop_sweep_store'),
            'tile128/512': ('kh_tn_forward_update', 'kh_tn_sweep_store'),
            'tile64x/512': ('kh_tx_forward_update', 'kh_tx_sweep_store'),  # five to eight controls (kh_tile64x.h)
            # more objectives than the GPU keeps co-resident: the streaming update kernel (kh_tile64s.h); the plain sweeps
            # take the objectives in turns
            'tile64/stream': ('kh_stream_forward_update', 'kh_q2_sweep_store' if args.L == 1 else 'kh_tile_sweep_store'),
            # an ensemble proper (one drift, scaled control operators) beyond the co-resident limit: the update sweep on the
            # matrix cores, several objectives per workgroup (kh_ens.h); plain sweeps as above
            'ens64/mfma': ('kh_ens_forward_update', 'kh_q2_sweep_store'),
        }.get(eng.kernel, ('kh_tile_forward_update', 'kh_tile_sweep_store'))
        if eng.kernel == 'ens64/mfma':
            # two objectives' columns per wave and one control: the A^2-chain form of the ensemble kernel (kh_ens.h:
            # kh_ens2_forward_update); which one ran is read off the library's launch record
            from krotov_amd import _lib as _khlib
            if any(name.startswith('kh_ens2_forward_update') for name in _khlib.kernel_instantiations(launched_only=True)):
                kernel_names = ('kh_ens2_forward_update', kernel_names[1])
        if group is not None and not getattr(eng, '_p2p_used', False) and eng.kernel != 'generic':
            # per-interval launches (RCCL path) run the two-tile kernel, see krotov_hip.hip:launch_update
            kernel_names = ('kh_tile_forward_update', kernel_names[1])

        diag = None
        if world > 1:
            # what every rank actually did (the first run on a real multi-GPU node must explain itself):
            # the transport, why if it is not the peer windows, the set-up self-test's round trip, and where workgroup 0
            # waited per interval -- inside its GPU or for the other GPUs (kh_p2p_stats)
            used = getattr(eng, '_p2p_used', False)
            fell = getattr(eng, '_p2p_fell_back', False)
            try:
                ps = eng.p2p_stats()
            except Exception as exc:
                ps = {'error': repr(exc)[:120]}
            mine = {
                'rank': rank, 'device': torch.cuda.current_device(), 'gpu': torch.cuda.get_device_name(),
                'transport': ('per-interval all-reduce after a fallback' if fell else
                              'peer windows' if used else 'all-reduce per interval'),
                'why': getattr(eng, 'p2p_why', None),
                'objectives': eng.K, 'kernel': eng.kernel,
                'update_sweep_ms': float(np.mean(times['update'][-args.steps:])) if times.get('update') else None,
                'backward_sweep_ms': float(np.mean(times['backward'][-args.steps:])) if times.get('backward') else None,
                'p2p': ps,
            }
            if os.environ.get('KH_P2P', '1') == '0' or fell or not used:
                # the transport of this measurement is one all-reduce of the L sums per interval: what one costs by itself
                x = torch.zeros(args.L, dtype=torch.float64, device='cuda')
                for _ in range(20):
                    torch.distributed.all_reduce(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    torch.distributed.all_reduce(x)
                torch.cuda.synchronize()
                mine['allreduce_us'] = (time.perf_counter() - t0) / 200 * 1e6
            try:
                gathered = [None] * world
                torch.distributed.all_gather_object(gathered, mine)
                diag = gathered
            except Exception as exc:  # (diagnostics must not cost the measurement: the line then carries this rank's only)
                mine['gather_error'] = repr(exc)[:160]
                diag = [mine]

        if rank == 0:
            K_loc = eng.K
            props = K_total * (args.nt - 1) * 2 * args.steps
            f_bw, f_up = algorithmic_flops(K_loc, args.N, args.nt, args.L)
            b_bw, b_up = algorithmic_bytes(K_loc, args.N, args.nt, args.L)
            t_bw = float(np.mean(times['backward'][-args.steps:])) * 1e-3
            t_up = float(np.mean(times['update'][-args.steps:])) * 1e-3
            ms_per_step = elapsed / args.steps * 1e3
            dominant = 'update' if t_up >= t_bw else 'backward'
            f_dom, t_dom = (f_up, t_up) if dominant == 'update' else (f_bw, t_bw)
            dom_kernel = kernel_names[0] if dominant == 'update' else kernel_names[1]
            traffic, traffic_source = pmc_traffic(dom_kernel, K_loc, args)
            # flops the dominant kernel really executed (its own count of matrix-vector products; the update sweep's is the
            # one kh_last_stats holds after an iteration): the A^2 chain and the shorter series issue fewer than credited
            executed = stats['matvecs'] * 8.0 * args.N * args.N if dominant == 'update' else None
            mfma_kernel = eng.kernel == 'coop16/mfma' or (eng.kernel == 'ens64/mfma' and dominant == 'update')
            out = {
                'metric': 'state*timestep propagations/s (Krotov iterations/s in iterations_per_sec), ' +
                          ('16-objective N=400 Liouvillian (variant)' if args.workload == 'c4' else
                           '256-objective N=64 ensemble' +
                           (' per GPU (weak scaling: %d objectives in total)' % K_total
                            if world > 1 and scaling == 'weak' else '')),
                'value': props / elapsed,
                'unit': 'props/s',
                'iterations_per_sec': args.steps / elapsed,
                'n_gpus': world,
                'steps': args.steps,
                'warmup': args.warmup,
                'ms_per_step': ms_per_step,
                'higher_is_better': True,
                'scaling': scaling,
                'vs_baseline': None,
                'dtype': 'f64',
                'data': 'synthetic',
                'config': {
                    'workload': ('BASELINE config 4 (variant line): transmon X-gate in Liouville space, %d density-matrix '
                                 'objectives sharing one %d-dim Liouvillian x %d time steps, L=%d control, chis_re, '
                                 'complex128; one step = one Krotov iteration' % (K_total, args.N, args.nt - 1, args.L))
                    if args.workload == 'c4' else
                                'BASELINE config 5: robustness ensemble, %d objectives%s x N=%d x %d time steps, '
                                'L=%d control, chis_re, complex128; one step = one Krotov iteration '
                                '(backward sweep + forward/update sweep)' % (
                                    K_total, ' (%d per GPU)' % eng.K if world > 1 else '', args.N, args.nt - 1, args.L),
                    'objectives': K_total, 'objectives_total': K_total, 'objectives_per_gpu': eng.K,
                    'N': args.N, 'time_steps': args.nt - 1, 'controls': args.L,
                    'distinct_drifts': bool(args.distinct),
                    'parallelism': 'objectives sharded over %d GPU(s); per time step the L update sums cross the '
                                   'GPUs %s' % (world, 'inside the persistent kernel (peer-mapped windows over xGMI)'
                                                if getattr(eng, '_p2p_used', False) else
                                                ('by one %s all-reduce per time step' % ('RCCL' if torch.distributed.get_backend(group) == 'nccl' else torch.distributed.get_backend(group) + ' (test set-up: fewer GPUs than ranks)') if group is not None else '(single GPU: in-kernel exchange)')),
                    'kernel': eng.kernel,
                },
                'roofline': {
                    # what the kernel is bound by, said truthfully: the register-tile kernels do their matrix-vector
                    # products with fp64 VECTOR FMAs (no MFMA in the propagation loop); only the cooperative kernels of
                    # config 4 are MFMA kernels.  The contract's vocabulary ("hbm" | "mfma") is kept in `contract_bound`:
                    # the fp64 vector peak and the dense fp64 MFMA peak are one pipe and one number on gfx950.
                    'bound': 'mfma' if mfma_kernel else 'fp64-valu',
                    'contract_bound': 'mfma',
                    'pipe': 'fp64 MFMA' if mfma_kernel else 'fp64 vector FMA (same pipe and same peak as the fp64 MFMA: scripts/ubench_hybrid.hip)',
                    'kernel': dom_kernel,
                    'achieved': f_dom / t_dom / 1e12,
                    'peak': FP64_PEAK_TFLOPS,
                    'unit': 'TFLOP/s',
                    'frac': f_dom / t_dom / 1e12 / FP64_PEAK_TFLOPS,
                    # the same fraction on EXECUTED flops (credited: the algorithmic formula of SURVEY.md 8d, degree 14)
                    'executed': None if executed is None else executed / t_dom / 1e12,
                    'executed_frac': None if executed is None else executed / t_dom / 1e12 / FP64_PEAK_TFLOPS,
                    'traffic': traffic,
                    'traffic_source': traffic_source,
                    'build': build_id(),
                    'note': ('fp64 MFMA (v_mfma_f64_16x16x4): the objectives share the operators, so a Taylor term is a '
                             'dense (N x N)(N x K) product; latency-bound by one cross-workgroup exchange per term. '
                             'Credited flops as below with m = 14 per propagation (SURVEY.md 8d), whatever was issued. '
                             if eng.kernel == 'coop16/mfma' else
                             'fp64 vector-FMA bound (complex matrix-vector products cannot use MFMA tiles); the fp64 '
                             'vector peak equals the fp64 MFMA peak on MI355X (78.6 TFLOP/s). ') +
                            'Algorithmic (credited) flops: K*(nt-1)*(8 N^2 * 14 [+ L*(8 N^2 + 8 N) for the update sweep]), '
                            'SURVEY.md 8d, whatever was issued: the q2 kernels evaluate the same degree-14 polynomial '
                            'with 8 matrix-vector products per step (A^2 chain + one A product by linearity); see '
                            'kernels.matvecs_issued_last_update_sweep for the executed count.',
                    'launch_ms': t_dom * 1e3,
                },
                'kernels': {
                    'backward_sweep_ms': t_bw * 1e3,
                    'update_sweep_ms': t_up * 1e3,
                    'backward_tflops': f_bw / t_bw / 1e12,
                    'update_tflops': f_up / t_up / 1e12,
                    'backward_hbm_gbs': b_bw / t_bw / 1e9,
                    'update_hbm_gbs': b_up / t_up / 1e9,
                    'hbm_frac_of_8TBs': max(b_bw / t_bw, b_up / t_up) / 1e9 / HBM_PEAK_GBS,
                    'matvecs_issued_last_update_sweep': stats['matvecs'],
                    'matvecs_credited_per_sweep': K_loc * (args.nt - 1) * (TAYLOR_DEGREE + args.L),
                    'update_executed_tflops': stats['matvecs'] * 8.0 * args.N * args.N / t_up / 1e12,
                },
                'final_J_T_re': float(1 - np.mean(np.array(res.tau_vals[-1]).real)),
            }
            if diag is not None:
                out['ranks'] = diag
                if args.workload == 'c5' and args.N == 64 and args.L == 1 and args.nt > 1000:
                    # DESIGN.md 4 / docs/HISTORY.md 4, written down before any second GPU was available: the first SCALE
                    # record confirms or refutes it by itself
                    out['predicted'] = {
                        'source': 'DESIGN.md 4 (docs/HISTORY.md 4): one GPU 4.8-4.9 us per interval of the update sweep; '
                                  'peer windows +1.0 us (cross-GPU stage, measured with ranks sharing one GPU) '
                                  '+0.5-1.0 us (xGMI store -> poll, assumed); all-reduce per interval 19 us step + 10-20 us collective',
                        'update_us_per_interval': [6.4, 6.9] if getattr(eng, '_p2p_used', False) else [30.0, 40.0],
                        'ms_per_step': [38.0, 40.0] if getattr(eng, '_p2p_used', False) else [130.0, 170.0],
                        'measured_update_us_per_interval': t_up * 1e6 / (args.nt - 1),
                        'measured_ms_per_step': ms_per_step,
                    }
            return out
        return None

    def leg(scaling='strong', env=None, pmc_case=None, **overrides):
        """A short extra measurement in this process with some arguments (and environment switches) changed;
        everything is put back afterwards.  Returns the reduced record that goes into the headline line."""
        saved = {k: getattr(args, k) for k in ('workload', 'K', 'N', 'nt', 'L', 'distinct', 'steps', 'warmup',
                                                'no_cpu_baseline')}
        saved_env = {k: os.environ.get(k) for k in (env or {})}
        try:
            for k, v in overrides.items():
                setattr(args, k, v)
            os.environ.update(env or {})
            line = measure(scaling)
            if line is None:
                return None
            rec = {
                'workload': line['config']['workload'], 'ms_per_step': line['ms_per_step'], 'value': line['value'],
                'unit': line['unit'], 'steps': line['steps'], 'kernel': line['config']['kernel'],
                'parallelism': line['config']['parallelism'],
                'roofline': {k: line['roofline'][k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'executed_frac')},
                'kernels': {k: line['kernels'][k] for k in ('backward_sweep_ms', 'update_sweep_ms')
                            if k in line['kernels']}}
            for k in ('ranks', 'predicted'):
                if k in line:
                    rec[k] = line[k]
            if pmc_case is not None:
                rec['roofline']['traffic'], rec['roofline']['traffic_source'] = pmc_traffic_leg(
                    pmc_case, line['roofline']['kernel'], line['config']['time_steps'])
                if rec['roofline']['traffic']:
                    # the same kernel against the OTHER roof: counter bytes per launch over its duration (the streaming
                    # kernel of K1024_distinct moves 136 MB per interval: this is the fraction that says how close it is)
                    gbs = rec['roofline']['traffic'] / (line['roofline']['launch_ms'] * 1e-3) / 1e9
                    rec['roofline']['hbm_gbs'] = gbs
                    rec['roofline']['hbm_frac'] = gbs / HBM_PEAK_GBS
            return rec
        except Exception as exc:  # (the headline line must not depend on a side measurement)
            if world > 1:
                raise  # ... but ranks must not part ways
            return {'error': repr(exc)[:200]}
        finally:
            for k, v in saved.items():
                setattr(args, k, v)
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    other = 'weak' if args.scaling == 'strong' else 'strong'
    out = measure(args.scaling)
    if rank == 0:
        out['n_ranks_seen'] = torch.distributed.get_world_size() if group is not None else 1
    else:
        out = {}

    def guarded(name, seconds, fn):
        """A side measurement of a sharded run under a watchdog: if it does not come back within `seconds` (a collective
        or an in-kernel wait that hangs cannot be interrupted from Python), every rank ends the process with the headline
        line -- and what the earlier side measurements gave -- printed and rc 0."""
        import threading

        def give_up(why=None):
            # (rc stays 0: the headline measurement above is complete and valid, and a non-zero exit would throw it away
            # with the side measurement; the line says so itself -- "degraded": true plus the leg's "error")
            if rank == 0:
                out[name] = {'error': why or 'no result within %d s' % seconds}
                out['degraded'] = True
                out['degraded_why'] = {k: v['error'] for k, v in out.items() if isinstance(v, dict) and 'error' in v}
                print(json.dumps(out), flush=True)
            os._exit(0)

        watchdog = threading.Timer(seconds, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            return fn()
        except Exception as exc:  # (the other ranks may be waiting in a collective: nobody goes on; theirs end the same way)
            give_up(repr(exc)[:200])
        finally:
            watchdog.cancel()

    def name_the_baseline_job(strong_line):
        # BASELINE.json's config 5 is 256 objectives IN TOTAL: on N > 1 GPUs that is the strong-scaling job, whichever of
        # the two is the headline.  A reader of `value` alone at N = 8 (weak: 2 048 objectives) finds the figure for the
        # job BASELINE names right next to it, under a name that says so.
        if rank == 0 and strong_line is not None and args.workload == 'c5':
            out['value_baseline_config5'] = strong_line['value']
            out['ms_per_step_baseline_config5'] = strong_line['ms_per_step']
            out['baseline_config5_objectives_total'] = strong_line['config']['objectives_total']

    if world == 1 or args.scaling == 'strong':
        name_the_baseline_job(out if rank == 0 else None)
    if world > 1 and args.workload == 'c5':
        # the same job with the other partitioning of the objectives (see the module docstring) -- like every side
        # measurement of a sharded run under the watchdog: the headline line above must get out whatever happens here
        second = guarded(other, args.rccl_leg_timeout, lambda: measure(other))
        if other == 'strong':
            name_the_baseline_job(second)
        if rank == 0 and second is not None:
            out[other] = {k: second[k] for k in ('value', 'unit', 'iterations_per_sec', 'ms_per_step', 'scaling')}
            out[other]['objectives'] = second['config']['objectives']
            out[other]['objectives_total'] = second['config']['objectives_total']
            out[other]['kernels'] = {k: second['kernels'][k] for k in ('backward_sweep_ms', 'update_sweep_ms')}
            for k in ('ranks', 'predicted'):
                if k in second:
                    out[other][k] = second[k]
    rccl = None
    if (world > 1 or args.force_dist) and args.workload == 'c5' and not args.no_rccl_leg:
        # the north star's transport: one RCCL all-reduce of the L update sums per time interval (kh_update_begin /
        # step_dev / end with HIP-graph replay) instead of the peer windows inside the persistent kernel.
        rccl = guarded('rccl', args.rccl_leg_timeout,
                       lambda: leg('strong', env={'KH_P2P': '0'}, steps=min(args.steps, 3), warmup=1))
        if rank == 0 and rccl is not None:
            out['rccl'] = rccl
            rccl = None
    if world > 1 and 16 % world == 0 and args.workload == 'c5' and not args.no_config4:
        # BASELINE config 4 is quoted on 2 and 4 GPUs: its 16 density matrices sharded over the ranks (16 / N each: the
        # cooperative matrix-core kernels with fewer column groups per GPU, the update sums across the GPUs through the
        # peer windows), three iterations.  Not part of `value`.
        c4 = guarded('config4', args.rccl_leg_timeout, lambda: leg('strong', workload='c4', steps=3, warmup=1))
        if rank == 0 and c4 is not None:
            out['config4'] = c4
    if rank == 0:
        if rccl is not None:
            out['rccl'] = rccl
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
            out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
            if out['cpu_baseline'].get('reference_loop'):
                out['speedup_vs_reference_loop_single_process'] = out['value'] / out['cpu_baseline']['reference_loop']['props_per_s']
    headline_shape = args.workload == 'c5' and args.K == 256 and args.N == 64 and args.L == 1 and not args.distinct
    if world == 1 and group is None and headline_shape:
        if not args.no_config4:
            # BASELINE config 4 (16 density matrices under one 400-dim Liouvillian: the cooperative fp64 MFMA kernels),
            # three iterations after the headline measurement, in this process (a second process on the GPU while this
            # one holds its context measured 2x slower): so that the default line -- the one the driver records -- also
            # carries a number for the one workload whose propagator is a dense product.  Not part of `value`.
            out['config4'] = leg(workload='c4', steps=3, warmup=1, pmc_case='config4')
        if not args.no_variants:
            # "L=1 (also report L=4)" and "a second variant with K distinct random H0_k"
            out['L4'] = leg(L=4, steps=3, warmup=1, pmc_case='L4')
            out['distinct'] = leg(distinct=True, steps=3, warmup=1)
            # five to eight controls: the register-tile kernels with streamed operators (kh_tile64x.h)
            out['L8'] = leg(L=8, steps=2, warmup=1, pmc_case='L8')
            # per-objective operators beyond the N <= 64 register tiles (kh_tilen.h: the generator in registers up to N = 128)
            out['N96'] = leg(N=96, steps=3, warmup=1, pmc_case='N96')
            # an ensemble that does not fit the GPU's co-resident workgroups: one drift and scaled control operators run
            # the update sweep on the matrix cores, several objectives per workgroup (kh_ens.h); per-objective drifts
            # stream their operators (kh_tile64s.h) -- SURVEY.md 8d asks for both
            out['K1024'] = leg(K=1024, steps=2, warmup=1, pmc_case='K1024')
            out['K1024_distinct'] = leg(K=1024, distinct=True, steps=2, warmup=1, pmc_case='K1024_distinct')
        if not args.no_sparse:
            try:
                out['sparse'] = sparse_leg()
            except Exception as exc:  # (the headline line must not depend on a side measurement)
                out['sparse'] = {'error': repr(exc)[:200]}
    if rank == 0:
        failed = {k: v['error'] for k, v in out.items() if isinstance(v, dict) and 'error' in v}
        if failed:
            out['degraded'] = True  # a side measurement was replaced by its error (the headline itself is complete)
            out['degraded_why'] = failed
        print(json.dumps(out), flush=True)
    if group is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
