"""Reference-structured CPU baseline of one Krotov iteration (TEST/BENCH
INFRASTRUCTURE -- only ``bench.py``'s ``cpu_baseline`` leg and tests use it).

Structure mirrors how the reference runs this path on host cores (BASELINE.md
section 3): per objective and per time step one dense ``expm(A dt) @ state``
(reference propagators.py:100-117) with BLAS pinned to one thread
(propagators.py:116), objectives distributed over ``P`` worker processes for
the backward sweep (parallelization.py:233-299), and long-lived per-chunk
workers with one synchronisation per time step for the forward/update sweep
(parallelization.py:433-495; here a shared-memory barrier instead of the
reference's queues, i.e. a *faster* exchange than the reference's).

It carries none of the reference's ``Qobj`` overhead, so it is faster than the
true reference: speed-ups quoted against it are conservative.
"""
import multiprocessing as mp
import os
import time

import numpy as np

from . import krotov_oracle as ko


def _worker(rank, P, ks, ops, init, chi_T, chi_norms, tlist, pulses, shapes, lambdas, shared_part, barrier,
            out_q, use_scipy):
    """Backward sweep for objectives ``ks``, then the forward/update sweep with
    a per-interval exchange of the partial sums through shared memory."""
    try:  # OpenBLAS re-creates its thread pool after fork(): pin it again (propagators.py:116)
        import threadpoolctl

        _pin = threadpoolctl.threadpool_limits(limits=1, user_api='blas')  # noqa: F841 (kept alive)
    except Exception:
        _pin = None
    nt = len(tlist)
    L = len(pulses)
    adj = [[None if o is None else o.conj().T for o in ops[k]] for k in ks]
    barrier.wait()
    t0 = time.perf_counter()
    # ---- backward sweep, storing chi(t_n) (optimize.py:849-886)
    store = np.empty((len(ks), nt, init.shape[1]), dtype=np.complex128)
    for i, k in enumerate(ks):
        state = chi_T[k].copy()
        store[i, nt - 1] = state
        for n in range(nt - 2, -1, -1):
            dt = tlist[n + 1] - tlist[n]
            state = ko.step(adj[i], [p[n] for p in pulses], dt, state, False, True, use_scipy)
            store[i, n] = state
    barrier.wait()
    t1 = time.perf_counter()
    # ---- forward sweep with sequential update (optimize.py:444-508)
    part = np.frombuffer(shared_part, dtype=np.float64).reshape(2, P, L)
    opt = [np.array(p, copy=True) for p in pulses]
    fw = [init[k].copy() for k in ks]
    for n in range(nt - 1):
        dt = tlist[n + 1] - tlist[n]
        for l in range(L):
            acc = 0.0
            for i, k in enumerate(ks):
                op = ops[k][1 + l]
                if op is not None:
                    acc += chi_norms[k] * np.vdot(store[i, n], op @ fw[i]).imag
            part[n & 1, rank, l] = acc
        barrier.wait()  # the one cross-objective exchange per interval (optimize.py:470)
        for l in range(L):
            d1 = 0.0
            for r in range(P):
                d1 += part[n & 1, r, l]
            opt[l][n] += shapes[l][n] / lambdas[l] * d1
        eps = [p[n] for p in opt]
        for i, k in enumerate(ks):
            fw[i] = ko.step(ops[k], eps, dt, fw[i], False, False, use_scipy)
    barrier.wait()
    t2 = time.perf_counter()
    out_q.put((rank, t0, t1, t2, np.array(opt) if rank == 0 else None, np.array(fw)))


def timed_iteration(spec, processes=None, use_scipy=True):
    """One Krotov iteration (chis_re) of ``spec`` on ``processes`` host cores.

    Returns dict(seconds, backward_seconds, update_seconds, props, processes,
    opt_pulses, fw_T).  ``props`` = state*timestep propagations performed
    (K * (nt-1) * 2).
    """
    K = spec.K
    if processes is None:
        processes = min(K, len(os.sched_getaffinity(0)))
    P = max(1, min(processes, K))
    ops = [[spec.H0[k]] + [spec.Hc[k][l] for l in range(spec.L)] for k in range(K)]
    _, gp, S = ko.initialize_controls(spec.controls, [spec.update_shape] * spec.L, spec.tlist)
    lambdas = [spec.lambda_a] * spec.L
    chi_T = (1.0 / (2 * K)) * spec.target
    chi_norms = np.linalg.norm(chi_T, axis=1)
    chi_T = chi_T / chi_norms[:, None]
    ctx = mp.get_context('fork')
    shared = ctx.RawArray('d', 2 * P * spec.L)
    barrier = ctx.Barrier(P)
    out_q = ctx.Queue()
    chunks = np.array_split(np.arange(K), P)
    procs = [
        ctx.Process(target=_worker, args=(r, P, list(chunks[r]), ops, spec.init, chi_T, chi_norms, spec.tlist,
                                          gp, S, lambdas, shared, barrier, out_q, use_scipy))
        for r in range(P)
    ]
    for p in procs:
        p.start()
    res = [out_q.get() for _ in range(P)]
    for p in procs:
        p.join()
    res.sort(key=lambda r: r[0])
    t0 = min(r[1] for r in res)
    t1 = max(r[2] for r in res)
    t2 = max(r[3] for r in res)
    return dict(
        seconds=t2 - t0,
        backward_seconds=t1 - t0,
        update_seconds=t2 - t1,
        props=K * (len(spec.tlist) - 1) * 2,
        processes=P,
        opt_pulses=res[0][4],
        fw_T=np.concatenate([r[5] for r in res], axis=0),
    )
