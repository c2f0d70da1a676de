"""``optimize_pulses``: Krotov's method behind the reference's plugin surface.

Same signature, keyword semantics, error behaviour and ``Result`` as
``krotov.optimize_pulses`` (reference src/krotov/optimize.py:33-590).  Two
execution paths share all bookkeeping:

* **device path** -- taken when ``propagator`` is this package's
  :func:`krotov_amd.propagators.expm` / :class:`~krotov_amd.propagators.HipExpm`
  and ``mu``, ``overlap``, ``storage`` are the defaults (first- or second-order
  update).  Per
  iteration the three objective-parallel dispatches of the reference
  (optimize.py:302-313, 413-425, 449-501) become three calls into
  ``libkrotov_hip.so``: all objectives are batched in one launch per sweep,
  the co-states never leave HBM, and the K*L*(nt-1) Python ``mu``/``overlap``
  calls of optimize.py:454-470 disappear into the kernel.  Objectives can be
  sharded over the GPUs of a node with ``process_group=`` (one rank per GPU,
  one all-reduce of the L update sums per interval).

* **plugin path** -- any other ``propagator`` (a user's own callable, in
  whatever arithmetic it implements) runs through a host loop with the
  reference's call structure, so custom ``propagator`` / ``mu`` / ``overlap`` /
  ``norm`` / ``parallel_map`` / ``storage`` plugins keep working.  This path
  does no arithmetic of its own.

There is no CPU fallback for the device path: without a GPU or without the
built library, requesting it raises.
"""
import copy
import inspect
import logging
import os
import time
from functools import partial

import numpy as np

from . import functionals as _functionals
from ._ingest import obj_type, state_array, state_to_vector, to_dense, to_sparse, vector_to_state
from .conversions import (
    control_onto_interval,
    discretize,
    extract_controls,
    extract_controls_mapping,
    plug_in_pulse_values,
    pulse_onto_tlist,
    pulse_options_dict_to_list,
)
from ._lib import KrotovHipError as _KrotovHipError
from ._lib import KH_ERR_TIMEOUT as _KH_ERR_TIMEOUT, KH_ERR_UNSUPPORTED as _KH_ERR_UNSUPPORTED

_KH_MAX_CONTROLS = 32  # KH_GEN_MAX_L of krotov_amd/csrc/kh_common.h (the register-resident kernel families: 8)
_FULL_GRID_PROBE_EVERY = 16  # update sweeps in a row on a reduced grid before the full one is tried again
from .info_hooks import chain
from .mu import derivative_wrt_pulse
from .parallelization import serial_map
from .propagators import HipExpm, Propagator, expm
from .result import Result
from .second_order import _overlap
from .shapes import one_shape, zero_shape
from .sharding import gather_rows, shard_range

__all__ = ['optimize_pulses']


# ---------------------------------------------------------------------------
# control initialisation (reference optimize.py:593-704)
# ---------------------------------------------------------------------------


def _shape_callable(val):
    if callable(val):
        return val
    if val == 1:
        return one_shape
    if val == 0:
        return zero_shape
    raise ValueError("update_shape must be a callable")


def _checked_shape(arr):
    """Shapes must lie in [0, 1] up to the rounding of the un-averaging; then
    clip (reference optimize.py:605-620)."""
    lo, hi = np.min(arr), np.max(arr)
    if lo < -0.01 or hi > 1.01:
        raise ValueError(
            "Update shapes ('update_shape' in pulse options-dict) must have "
            "values in the range [0, 1], not [%s, %s]" % (lo, hi)
        )
    return np.clip(arr, a_min=0.0, a_max=1.0)


def _initialize_krotov_controls(objectives, pulse_options, tlist):
    guess_controls = extract_controls(objectives)
    pulses_mapping = extract_controls_mapping(objectives, guess_controls)
    options_list = pulse_options_dict_to_list(pulse_options, guess_controls)
    try:
        guess_controls = [
            discretize(c, tlist, args=(options_list[i].get('args', None),), via_midpoints=True)
            for i, c in enumerate(guess_controls)
        ]
    except TypeError as exc:
        raise ValueError(
            "Cannot discretize controls: %s. Note that "
            "all controls must be real-valued. Complex controls must be "
            "split into an independent real and imaginary part in the "
            "objectives before passing them to the optimization" % exc
        )
    guess_pulses = [control_onto_interval(c) for c in guess_controls]
    try:
        lambda_vals = np.array([float(o['lambda_a']) for o in options_list])
    except KeyError:
        raise ValueError("Each value in pulse_options must be a dict that contains the key 'lambda_a'.")
    shape_arrays = []
    for o in options_list:
        try:
            S = discretize(_shape_callable(o['update_shape']), tlist, args=(), via_midpoints=True)
        except KeyError:
            raise ValueError("Each value in pulse_options must be a dict that contains the key 'update_shape'.")
        except TypeError as exc:
            raise ValueError(
                "Update shapes ('update_shape' in pulse options-dict) must be real-valued: %s" % exc
            )
        shape_arrays.append(_checked_shape(control_onto_interval(S)))
    return guess_controls, guess_pulses, pulses_mapping, lambda_vals, shape_arrays


def _restore_from_previous_result(result, objectives, tlist, store_all_pulses):
    """Guess controls/pulses of a continued optimisation (reference
    optimize.py:707-774), with the same compatibility checks."""
    if not isinstance(result, Result):
        raise ValueError("Continuation is only possible from a Result object")
    if len(objectives) != len(result.objectives):
        raise ValueError("When continuing from a previous Result, the number of objectives must be the same")
    for a, b in zip(objectives, result.objectives):
        if a != b:
            raise ValueError("When continuing from a previous Result, the objectives must remain unchanged")
    if store_all_pulses and len(result.all_pulses) == 0:
        raise ValueError(
            "The store_all_pulses parameter cannot be changed when continuing from a previous Result. "
            "Pass it as False."
        )
    if not store_all_pulses and len(result.all_pulses) > 0:
        raise ValueError(
            "The store_all_pulses parameter cannot be changed when continuing from a previous Result. "
            "Pass it as True."
        )
    same_grid = len(tlist) == len(result.tlist) and np.max(np.abs(np.array(tlist) - np.array(result.tlist))) <= 1e-5
    if not same_grid:
        raise ValueError("When continuing from a previous Result, the controls must be defined on the same time grid")
    nt = len(tlist)
    guess_controls = []
    for control in result.optimized_controls:
        if len(control) == nt - 1:  # dumped mid-optimisation: these are pulses
            guess_controls.append(pulse_onto_tlist(control))
        elif len(control) == nt:
            guess_controls.append(control)
        else:
            raise ValueError("Invalid Result: optimized_controls and tlist are incongruent")
    return guess_controls, [control_onto_interval(c) for c in guess_controls]


def _check_propagators_interface(propagators, logger):
    """Warn about propagators whose signature is neither ``expm``'s nor
    ``Propagator.__call__``'s (reference optimize.py:623-638)."""
    ok = (inspect.getfullargspec(expm), inspect.getfullargspec(Propagator.__call__))
    for p in propagators:
        try:
            spec = inspect.getfullargspec(p)
        except TypeError:
            spec = None
        if spec not in ok:
            logger.warning("The propagator %s does not have the expected interface.", p)


# ---------------------------------------------------------------------------
# plugin path: the reference's call structure around user callables
# ---------------------------------------------------------------------------


def _forward_propagation(i_objective, objectives, pulses, pulses_mapping, tlist, propagators, storage,
                         store_all=True):
    """Task for ``parallel_map[0]`` (reference optimize.py:806-846)."""
    obj = objectives[i_objective]
    state = obj.initial_state
    mapping = pulses_mapping[i_objective]
    out = None
    if store_all:
        out = storage(len(tlist))
        out[0] = state
    for n in range(len(tlist) - 1):
        H = plug_in_pulse_values(obj.H, pulses, mapping[0], n)
        c_ops = [plug_in_pulse_values(c, pulses, mapping[ic + 1], n) for ic, c in enumerate(obj.c_ops)]
        state = propagators[i_objective](H, state, tlist[n + 1] - tlist[n], c_ops, initialize=(n == 0))
        if store_all:
            out[n + 1] = state
    return out if store_all else state


def _backward_propagation(i_state, chi_states, adjoint_objectives, pulses, pulses_mapping, tlist, propagators,
                          storage):
    """Task for ``parallel_map[1]`` (reference optimize.py:849-886)."""
    state = chi_states[i_state]
    obj = adjoint_objectives[i_state]
    mapping = pulses_mapping[i_state]
    nt = len(tlist)
    out = storage(nt)
    out[-1] = state
    for n in range(nt - 2, -1, -1):
        H = plug_in_pulse_values(obj.H, pulses, mapping[0], n, conjugate=True)
        c_ops = [plug_in_pulse_values(c, pulses, mapping[ic + 1], n) for ic, c in enumerate(obj.c_ops)]
        state = propagators[i_state](
            H, state, tlist[n + 1] - tlist[n], c_ops, backwards=True, initialize=(n == nt - 2)
        )
        out[n] = state
    return out


def _forward_propagation_step(i_state, states, objectives, pulses, pulses_mapping, tlist, time_index, propagators):
    """Task for ``parallel_map[2]`` (reference optimize.py:889-911)."""
    obj = objectives[i_state]
    mapping = pulses_mapping[i_state]
    H = plug_in_pulse_values(obj.H, pulses, mapping[0], time_index)
    c_ops = [plug_in_pulse_values(c, pulses, mapping[ic + 1], time_index) for ic, c in enumerate(obj.c_ops)]
    dt = tlist[time_index + 1] - tlist[time_index]
    return propagators[i_state](H, states[i_state], dt, c_ops, initialize=(time_index == 0))


class _PluginBackend:
    """Host loop over user callables, call-for-call like the reference."""

    device = False

    def __init__(self, objectives, adjoint_objectives, pulses_mapping, tlist, propagators, storage, parallel_map,
                 mu, overlap):
        self.objectives = objectives
        self.adjoint_objectives = adjoint_objectives
        self.mapping = pulses_mapping
        self.tlist = tlist
        self.propagators = propagators
        self.storage = storage
        self.pmap = parallel_map
        self.mu = mu
        self.overlap = overlap

    def initial_forward(self, pulses):
        K = len(self.objectives)
        forward_states = self.pmap[0](
            _forward_propagation, list(range(K)),
            (self.objectives, pulses, self.mapping, self.tlist, self.propagators, self.storage),
        )
        return [s[-1] for s in forward_states], forward_states

    def tau_vals(self, fw_states_T):
        return np.array([self.overlap(obj.target, s) for s, obj in zip(fw_states_T, self.objectives)])

    def iterate(self, chi_states, chi_norms, guess_pulses, lambda_vals, shape_arrays, sigma=None,
                forward_states0=None):
        objectives, tlist = self.objectives, self.tlist
        K, nt = len(objectives), len(tlist)
        backward_states = self.pmap[1](
            _backward_propagation, list(range(K)),
            (chi_states, self.adjoint_objectives, guess_pulses, self.mapping, tlist, self.propagators, self.storage),
        )
        g_a = np.zeros(len(guess_pulses))
        optimized = copy.deepcopy(guess_pulses)
        fw_states = [obj.initial_state for obj in objectives]
        forward_states = moved = None
        if sigma is not None:  # second order: keep phi(t_n) and phi(t_n) - phi_prev(t_n) (optimize.py:429-442)
            forward_states = [self.storage(nt) for _ in range(K)]
            for k in range(K):
                forward_states[k][0] = objectives[k].initial_state
            moved = [fw_states[k] - forward_states0[k][0] for k in range(K)]  # zero at t=0
        for n in range(nt - 1):
            dt = tlist[n + 1] - tlist[n]
            if sigma is not None:
                half_sigma = 0.5 * sigma(tlist[n] + 0.5 * dt)
            for l in range(len(guess_pulses)):
                total = 0j
                for k in range(K):  # optimize.py:455-470
                    mu_op = self.mu(objectives, k, guess_pulses, self.mapping, l, n)
                    mu_phi = mu_op(fw_states[k])
                    update = self.overlap(backward_states[k][n], mu_phi)
                    update *= chi_norms[k]
                    if sigma is not None:
                        update += half_sigma * self.overlap(moved[k], mu_phi)
                    total += update
                step = shape_arrays[l][n] / lambda_vals[l]
                d1 = total.imag
                g_a[l] += step * abs(d1) ** 2 * dt
                optimized[l][n] += step * d1
            fw_states = self.pmap[2](
                _forward_propagation_step, list(range(K)),
                (fw_states, objectives, optimized, self.mapping, tlist, n, self.propagators),
            )
            if sigma is not None:
                moved = [fw_states[k] - forward_states0[k][n + 1] for k in range(K)]
                for k in range(K):
                    forward_states[k][n + 1] = fw_states[k]
        return backward_states, optimized, fw_states, g_a, forward_states


# ---------------------------------------------------------------------------
# device path
# ---------------------------------------------------------------------------


class _LazyStates:
    """List-like view of per-objective states held as one (K, N) array -- on the
    host, or still on the device and fetched on first access (``fetch``);
    elements are converted to the caller's state type on access."""

    def __init__(self, array, likes, fetch=None):
        self._host = array
        self._fetch = fetch
        self._likes = likes

    @property
    def _array(self):
        if self._host is None:
            self._host = self._fetch()
            self._fetch = None
        return self._host

    def __len__(self):
        return len(self._likes)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        return vector_to_state(self._array[k], self._likes[k])

    def __iter__(self):
        return (self[k] for k in range(len(self)))

    def __reduce__(self):
        # pickled (Result.dump from a check_convergence hook, copy.deepcopy) as the plain list of states
        # it stands for: the fetch closure holds device tensors and must not travel
        return (list, (list(self),))


class _DeviceTrajectory:
    """States of one objective on the time grid, one device row per access."""

    def __init__(self, owner, k):
        self._o, self._k = owner, k

    def __len__(self):
        return self._o._t.shape[1]

    def __getitem__(self, n):
        o, k = self._o, self._k
        nt = len(self)
        if isinstance(n, slice):
            return [self[i] for i in range(*n.indices(nt))]
        if not -nt <= n < nt:
            raise IndexError("time index %d out of range" % n)
        n %= nt
        like = o._likes[k]
        if n == nt - 1 and o._final is not None:
            return vector_to_state(o._final[k], like)
        if not o._k0 <= k < o._k0 + o._t.shape[0]:
            raise IndexError("objective %d lives on another rank; only its final state is available here" % k)
        return vector_to_state(o._t[k - o._k0, n].cpu().numpy(), like)

    def __iter__(self):
        return (self[n] for n in range(len(self)))


class _DeviceTrajectories:
    """``backward_states[k][n]`` / ``forward_states[k][n]`` for ``info_hook`` and
    ``sigma.refresh``: (K_loc, nt, N) in HBM, fetched on access.  ``final``: the
    states at T of all objectives of all ranks on the host, if known."""

    def __init__(self, tensor, likes, k0=0, final=None):
        self._t, self._likes, self._k0, self._final = tensor, likes, k0, final

    def __len__(self):
        return len(self._likes) if self._final is not None else self._t.shape[0]

    def __getitem__(self, k):
        if self._final is None:
            k += self._k0  # local numbering (backward states of this rank)
        if not 0 <= k < len(self._likes):
            raise IndexError("objective index out of range")
        return _DeviceTrajectory(self, k)

    def __iter__(self):
        return (self[k] for k in range(len(self)))


def _use_device_path(propagator, mu, overlap, sigma, storage, objectives):
    if isinstance(propagator, list):
        if not propagator or any(not (p is expm or isinstance(p, HipExpm)) for p in propagator):
            return False
    elif not (propagator is expm or isinstance(propagator, HipExpm)):
        return False
    if mu is not None and mu is not derivative_wrt_pulse:
        return False
    if overlap is not None and overlap is not _overlap:
        return False
    if storage != 'array':
        return False
    return all(len(obj.c_ops) == 0 for obj in objectives)


class _HipBackend:
    """All objectives of this rank resident on one GPU."""

    device = True

    def __init__(self, objectives, pulses_mapping, tlist, n_controls, propagator, process_group=None):
        import torch

        from .engine import HipKrotovEngine

        self.torch = torch
        self.group = process_group
        self.objectives = objectives
        K_total = len(objectives)
        self.rank, self.world = 0, 1
        if process_group is not None:
            import torch.distributed as dist

            self.dist = dist
            self.rank = dist.get_rank(process_group)
            self.world = dist.get_world_size(process_group)
        # contiguous shard of objectives for this rank
        self.k0, self.k1 = shard_range(K_total, self.world, self.rank)
        if K_total < self.world:
            # the same error on EVERY rank, before any collective (a lone raising rank would leave the
            # others hanging in the first all-gather)
            raise ValueError("%d objectives cannot be sharded over %d ranks" % (K_total, self.world))
        self.K_total = K_total
        L = n_controls
        dense = {}
        props = propagator if isinstance(propagator, list) else [propagator]
        # operators stay in CSR form on the device when the propagator asks for it
        # (DensityMatrixODEPropagator / HipExpm(sparse=True): large sparse Liouvillians)
        self.sparse = any(isinstance(p, HipExpm) and getattr(p, 'sparse', False) for p in props)
        convert = to_sparse if self.sparse else to_dense

        def dense_of(op):
            key = id(op)
            if key not in dense:
                dense[key] = (convert(op), op)
            return dense[key][0]

        sums = {}

        def summed(terms):
            """Dense sum of several operators, cached on their identities so
            objectives sharing the same nested list share one device copy."""
            if len(terms) == 1:
                return dense_of(terms[0])
            key = tuple(id(t) for t in terms)
            if key not in sums:
                total = dense_of(terms[0]).copy()
                for t in terms[1:]:
                    total = total + dense_of(t)
                sums[key] = (total, terms)
            return sums[key][0]

        liouville = None
        for p in props:
            if isinstance(p, HipExpm) and p.liouville is not None:
                liouville = bool(p.liouville)
        ops, first_op = [], None
        for k in range(self.k0, self.k1):
            obj = objectives[k]
            H = obj.H if isinstance(obj.H, list) else [obj.H]
            drift = [t for t in H if not isinstance(t, list)]
            if len(drift) == 0:
                raise ValueError("objective %d has no drift term in H" % k)
            if first_op is None:
                first_op = drift[0]
            row = [summed(drift)]
            for l in range(L):
                where = pulses_mapping[k][0][l]
                row.append(summed([H[i][0] for i in where]) if len(where) else None)
            ops.append(row)
        N = ops[0][0].shape[0]
        if liouville is None:
            if obj_type(first_op) is not None:
                liouville = obj_type(first_op) == 'super'
            else:
                s0 = state_array(objectives[self.k0].initial_state)
                liouville = bool(s0.ndim == 2 and s0.shape[0] == s0.shape[1] and s0.shape[0] > 1 and s0.size == N)
        self.is_super = liouville
        self.N, self.L = N, L
        tlist = np.asarray(tlist, dtype=np.float64)
        self.engine = HipKrotovEngine(ops, np.diff(tlist), is_super=self.is_super)
        self.nt = len(tlist)
        self.tlist_host = tlist
        self.likes = [obj.initial_state for obj in objectives]
        init = [state_to_vector(obj.initial_state, N, self.is_super) for obj in objectives[self.k0:self.k1]]
        if any(v is None for v in init):
            raise ValueError("initial states do not match the operator dimension %d" % N)
        self.init_host = np.array(init)
        self.init = self.engine.dev(self.init_host, torch.complex128)
        tg = [state_to_vector(obj.target, N, self.is_super) for obj in objectives]
        self.targets_host = None if any(v is None for v in tg) else np.array(tg)
        self.targets = (
            None if self.targets_host is None
            else self.engine.dev(self.targets_host[self.k0:self.k1], torch.complex128)
        )
        w = [getattr(obj, 'weight', None) for obj in objectives]
        self.weights = None if all(x is None for x in w) else np.array([1.0 if x is None else x for x in w])
        self.chi_store = None
        self.fw_T_dev = None
        self.fw_prev = self.fw_next = None  # second order: phi(t_n) of the last / the running iteration
        self.last_chi = None
        # device-side exchange over peer-mapped windows (xGMI) when every rank can set it up;
        # otherwise (or after a failed sweep) one RCCL all-reduce per interval
        self._single_launch_ok = True  # (one GPU) until the single-launch update sweep has failed for good
        self._single_launch_failures = 0
        self._update_grid = None  # workgroups of the single-launch update sweep after a retry (None: the engine's choice)
        self._reduced_in_a_row = 0
        self._last_good_grid = None  # the reduced grid that got through the last time the full one timed out
        self.p2p = False
        self.engine.p2p_why = "KH_P2P=0" if self.world > 1 else None
        if self.world > 1 and os.environ.get('KH_P2P', '1') != '0':
            self.p2p = self.engine.enable_p2p(self.group)
            logging.getLogger('krotov').info(
                "cross-GPU exchange: %s", "peer windows" if self.p2p else "RCCL all-reduce per interval")

    # -- helpers -----------------------------------------------------------
    def _cached_upload(self, key, host, dtype=None):
        """Device copy of a small host array that is usually the same from one iteration to the next (every upload from
        pageable memory blocks the host for ~30 us with the GPU idle)."""
        cache = self.__dict__.setdefault('_uploads', {})
        hit = cache.get(key)
        if hit is not None and hit[0].shape == host.shape and np.array_equal(hit[0], host):
            return hit[1]
        dev = self.engine.dev(host, dtype if dtype is not None else self.torch.float64)
        cache[key] = (host.copy(), dev)
        return dev

    def _pulses(self, pulses, cached=False):
        host = np.array(pulses, dtype=np.float64).reshape(self.L, self.nt - 1)
        if cached:  # (the guess of an iteration is normally the optimized pulse of the one before: still on the device)
            return self._cached_upload('pulses', host)
        return self.engine.dev(host, self.torch.float64)

    def _gather_rows(self, local):
        """(K_loc, ...) host array on every rank -> (K_total, ...) on every rank."""
        return gather_rows(local, self.K_total, self.world, self.group, self.engine.device)

    def initial_forward(self, pulses, store=False):
        forward_states = None
        if store:
            self.fw_T_dev, self.fw_prev = self.engine.forward(self._pulses(pulses), self.init, store=True)
        else:
            self.fw_T_dev = self.engine.forward(self._pulses(pulses), self.init)
        fw_states_T = self._final_states(self.fw_T_dev)
        if store:
            forward_states = _DeviceTrajectories(self.fw_prev, self.likes, self.k0, final=fw_states_T._array)
        return fw_states_T, forward_states

    def tau_vals(self, fw_states_T):
        if self.targets is None:
            return np.array([None] * self.K_total)
        tau = self.engine.tau(self.targets, self.fw_T_dev).cpu().numpy()
        return self._gather_rows(tau)

    def _final_states(self, psi_T):
        """phi_k(T) of all objectives for the caller: gathered now when sharded (a
        collective), else left on the device until somebody looks at them."""
        if self.world > 1:
            return _LazyStates(self._gather_rows(psi_T.cpu().numpy()), self.likes)
        return _LazyStates(None, self.likes, fetch=lambda: psi_T.cpu().numpy())

    def chi_host(self):
        """Normalised chi_k(T) and their norms of the last iteration, all objectives, on
        the host (only the second-order ``sigma.refresh`` needs them)."""
        chi, norms = self.last_chi
        return self._gather_rows(chi.cpu().numpy()), self._gather_rows(norms.cpu().numpy())

    def iterate(self, chi_T, chi_norms, guess_pulses, lambda_vals, shape_arrays, sigma=None, chi_coef=None):
        """chi_T: (K_total, N) normalised co-states (host); chi_norms (K_total,) -- or
        ``chi_coef = (c, d)``, (K_total,) each: chi_k(T) = c_k target_k + d_k phi_k(T) is
        then formed and normalised on the device (kh_chi_boundary)."""
        t = self.torch
        eng = self.engine
        guess = self._pulses(guess_pulses, cached=True)
        # every upload of the iteration BEFORE the first sweep is launched: a host-to-device copy from pageable memory
        # is ordered behind the kernels already in the stream and blocks the host until it is done -- issued between the
        # two sweeps it kept the update sweep's launch back until the backward sweep had finished (the GPU idled for
        # two copies and a launch per iteration).  Shapes and step widths rarely change: uploaded when they do.
        shapes = self._cached_upload('shapes', np.array(shape_arrays, dtype=np.float64).reshape(self.L, self.nt - 1))
        lambdas = self._cached_upload('lambdas', np.asarray(lambda_vals, dtype=np.float64))
        if sigma is not None:
            # sigma at the interval mid-points (optimize.py:451-452); phi under the guess pulses is
            # the trajectory the previous sweep stored, the running one goes to the other buffer
            tl = np.asarray(self.tlist_host)
            sig = np.array([sigma(tl[n] + 0.5 * (tl[n + 1] - tl[n])) for n in range(self.nt - 1)], dtype=np.float64)
            if self.fw_next is None:
                self.fw_next = t.empty_like(self.fw_prev)
            eng.set_second_order(self.fw_prev, self.fw_next, sig)
        if chi_coef is not None:
            c, d = chi_coef
            psi = self.fw_T_dev if self.fw_T_dev is not None else self.init  # (d == 0 without phi(T))
            c_dev = self._cached_upload('chi_c', np.ascontiguousarray(c[self.k0:self.k1], dtype=np.complex128), t.complex128)
            d_dev = self._cached_upload('chi_d', np.ascontiguousarray(d[self.k0:self.k1], dtype=np.complex128), t.complex128)
            chi_loc, norms_loc = eng.chi_boundary(self.targets, psi, c_dev, d_dev)
        else:
            chi_loc = eng.dev(chi_T[self.k0:self.k1], t.complex128)
            norms_loc = eng.dev(np.asarray(chi_norms, dtype=np.float64)[self.k0:self.k1], t.float64)
        self.last_chi = (chi_loc, norms_loc)
        self.chi_store = eng.backward(chi_loc, guess, out=self.chi_store)
        done = False
        if self.group is None:
            if self._single_launch_ok:
                # the single-launch sweep needs all its workgroups resident at once; if the GPU could not give it that
                # (CUs held by another stream or process: its in-kernel exchange times out; or the device cannot hold
                # the grid at all), redo it on HALF the workgroups, each walking through twice the objectives (about
                # twice the time: kh_set_update_workgroups), then on an eighth, and only then interval by interval -- one
                # launch each, nothing waits inside a kernel, ten times the time -- like the sharded path does.  A
                # reduced grid that got through stays.
                # Anything else than a timeout / "cannot be resident" is a real error.
                grid = self._update_grid  # (None: the engine's own choice)
                if grid is not None and self._reduced_in_a_row >= _FULL_GRID_PROBE_EVERY:
                    # a co-tenant may have gone away: one sweep on the full grid again (at worst one more timeout)
                    eng.set_update_workgroups(0)
                    grid, self._reduced_in_a_row = None, 0
                rungs = 0
                while not done:
                    try:
                        opt, psi_T, g_a = eng.forward_update(self.chi_store, norms_loc, self.init, guess, shapes, lambdas)
                        eng.check()
                        done = True
                    except _KrotovHipError as exc:
                        if exc.code not in (_KH_ERR_TIMEOUT, _KH_ERR_UNSUPPORTED):
                            raise
                        try:
                            # every rung costs a timed-out sweep (KH_TIMEOUT_MS, 1 s by default): two of them at most --
                            # half the workgroups (what gets through next to a stream holding half of the device), then
                            # an eighth -- before the form that cannot time out
                            rungs += 1
                            if rungs > 2:
                                raise _KrotovHipError("two reduced grids timed out", _KH_ERR_UNSUPPORTED)
                            full = eng.set_update_workgroups(0)
                            if grid is None and self._last_good_grid is not None and self._last_good_grid < full:
                                nxt = self._last_good_grid  # (what got through the last time this happened)
                            else:
                                nxt = (grid if grid is not None else full) // (2 if rungs == 1 else 4)
                            if nxt < 1:
                                raise _KrotovHipError("no smaller grid", _KH_ERR_UNSUPPORTED)
                            grid = eng.set_update_workgroups(nxt)
                            logging.getLogger('krotov').warning(
                                "single-launch update sweep failed (%s); repeating it on %d workgroups", exc, grid)
                        except _KrotovHipError as exc2:
                            if exc2.code != _KH_ERR_UNSUPPORTED:
                                raise
                            # back to the engine's own grid, in the engine AND in what this object remembers of it
                            grid = None
                            eng.set_update_workgroups(0)
                            self._update_grid, self._reduced_in_a_row, self._last_good_grid = None, 0, None
                            self._single_launch_failures += 1
                            # a co-tenant may go away, but not after three sweeps in a row
                            if exc.code == _KH_ERR_UNSUPPORTED or self._single_launch_failures >= 3:
                                self._single_launch_ok = False
                            logging.getLogger('krotov').warning(
                                "single-launch update sweep failed (%s); repeating it with one launch per interval%s", exc,
                                "" if self._single_launch_ok else " (and staying with that form)")
                            break
                if done:
                    self._single_launch_failures = 0
                    if grid is not None:
                        # a reduced grid got through: it is KEPT (trying the full one first would cost a timed-out sweep
                        # per iteration for as long as the co-tenant stays); the full grid gets its chance again after
                        # _FULL_GRID_PROBE_EVERY sweeps in a row on the reduced one
                        self._reduced_in_a_row += 1
                        self._last_good_grid = grid
                    else:
                        self._reduced_in_a_row = 0
                    self._update_grid = grid
            if not done:
                opt, psi_T, g_a = eng.forward_update_sharded(
                    self.chi_store, norms_loc, self.init, guess, shapes, lambdas, lambda x: None, graph_chunk=0)
                done = True
        elif self.p2p:
            # one persistent launch per rank; the per-GPU sums cross the node inside the kernel
            failed = 0
            if not self.__dict__.get('_p2p_synced', False):
                # the ranks' kernels wait for each other INSIDE the sweep (bounded by KH_TIMEOUT_MS): before the first
                # one, line the ranks up -- set-up time (imports, engine creation, operator norms) differs by more than
                # that bound between ranks; afterwards they stay within a sweep's jitter of each other
                t.cuda.synchronize(eng.device)
                self.dist.barrier(group=self.group)
                self._p2p_synced = True
            try:
                opt, psi_T, g_a = eng.forward_update(self.chi_store, norms_loc, self.init, guess, shapes, lambdas)
                eng.check()
            except Exception as exc:  # exchange timeout: every rank falls back together
                logging.getLogger('krotov').warning("cross-GPU exchange failed (%s)", exc)
                eng.p2p_why = "peer windows dropped after a failed sweep on this rank: %s" % exc
                failed = 1
            flag = t.tensor([failed], dtype=t.int32, device=eng.device)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX, group=self.group)
            if int(flag.item()) == 0:
                done = True
                eng._p2p_used = True
            else:
                self.p2p = False
                eng.disable_p2p()
                eng._p2p_fell_back = True  # (stays with the per-interval transport from here on)
                if not failed:
                    eng.p2p_why = "peer windows dropped: the sweep failed on another rank"
        if not done:
            def all_reduce(x):
                self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM, group=self.group)

            # HIP-graph replay of the interval loop needs a capturable collective (RCCL)
            chunk = None if self.dist.get_backend(self.group) == 'nccl' else 0
            opt, psi_T, g_a = eng.forward_update_sharded(
                self.chi_store, norms_loc, self.init, guess, shapes, lambdas, all_reduce, graph_chunk=chunk)
        self.fw_T_dev = psi_T
        eng.check()
        fw_states_T = self._final_states(psi_T)
        opt_host = opt.cpu().numpy()
        self.__dict__.setdefault('_uploads', {})['pulses'] = (opt_host.copy(), opt)  # (the next iteration's guess, if unchanged)
        optimized = [opt_host[l].copy() for l in range(self.L)]
        backward_states = _DeviceTrajectories(self.chi_store, self.likes, self.k0)
        forward_states = None
        if sigma is not None:
            forward_states = _DeviceTrajectories(self.fw_next, self.likes, self.k0, final=fw_states_T._array)
        return backward_states, optimized, fw_states_T, g_a.cpu().numpy(), forward_states

    def advance_second_order(self):
        """The stored trajectory of the finished iteration becomes ``forward_states0``
        of the next one (optimize.py:577)."""
        self.fw_prev, self.fw_next = self.fw_next, self.fw_prev


# ---------------------------------------------------------------------------
# the driver
# ---------------------------------------------------------------------------


def optimize_pulses(
    objectives,
    pulse_options,
    tlist,
    *,
    propagator,
    chi_constructor,
    mu=None,
    sigma=None,
    iter_start=0,
    iter_stop=5000,
    check_convergence=None,
    info_hook=None,
    modify_params_after_iter=None,
    storage='array',
    parallel_map=None,
    store_all_pulses=False,
    continue_from=None,
    skip_initial_forward_propagation=False,
    norm=None,
    overlap=None,
    limit_thread_pool=None,
    process_group=None,
):
    """Optimise all controls in ``objectives`` with Krotov's method.

    Arguments, callbacks, returned :class:`~krotov_amd.result.Result` and raised
    ``ValueError`` s are those of ``krotov.optimize_pulses`` (reference
    optimize.py:33-228).  Differences:

    * ``propagator=krotov_amd.propagators.expm`` runs on the GPU (see module
      docstring); ``parallel_map`` is then unused -- objectives are batched in
      the kernels.
    * ``process_group`` (extension): a ``torch.distributed`` group with one
      rank per GPU; every rank passes the full objective list and gets the
      full result, objectives are sharded contiguously over the ranks.
    * ``sigma`` (second order): on the device path the trajectories phi(t_n) of
      the last two iterations stay in HBM; ``forward_states`` / ``forward_states0``
      handed to ``info_hook`` and ``sigma.refresh`` fetch single states on
      access (with ``process_group``: all time points of this rank's
      objectives, the final time of every objective).
    * ``limit_thread_pool`` is accepted and ignored (no BLAS on the hot path).
    """
    logger = logging.getLogger('krotov')
    logger.info("Initializing optimization with Krotov's method")
    second_order = sigma is not None
    if second_order and skip_initial_forward_propagation:
        raise ValueError(
            "skip_initial_forward_propagation is incompatible with second order Krotov (sigma is not None)"
        )
    device_path = _use_device_path(propagator, mu, overlap, sigma, storage, objectives)
    if process_group is not None and not device_path:
        raise ValueError("process_group requires the device path (propagator=krotov_amd.propagators.expm)")
    if mu is None:
        mu = derivative_wrt_pulse
    default_norm = norm is None
    if norm is None:
        def norm(state):
            return state.norm() if hasattr(state, 'norm') else float(np.linalg.norm(np.asarray(state)))
    if overlap is None:
        overlap = _overlap
    if modify_params_after_iter is not None:
        info_hook = modify_params_after_iter if info_hook is None else chain(modify_params_after_iter, info_hook)
    if isinstance(propagator, list):
        propagators = propagator
        assert len(propagators) == len(objectives)
    else:
        propagators = [copy.deepcopy(propagator) for _ in objectives]
    _check_propagators_interface(propagators, logger)

    adjoint_objectives = [obj.adjoint() for obj in objectives]
    if storage == 'array':
        storage = partial(np.empty, dtype=object)
    if parallel_map is None:
        parallel_map = serial_map
    if not isinstance(parallel_map, (tuple, list)):
        parallel_map = (parallel_map, parallel_map, parallel_map)

    (guess_controls, guess_pulses, pulses_mapping, lambda_vals, shape_arrays) = _initialize_krotov_controls(
        objectives, pulse_options, tlist
    )
    if continue_from is not None:
        guess_controls, guess_pulses = _restore_from_previous_result(
            continue_from, objectives, tlist, store_all_pulses
        )
    g_a_integrals = np.zeros(len(guess_pulses))
    if device_path and len(guess_pulses) > _KH_MAX_CONTROLS and process_group is None:
        # the sweep kernels are compiled for at most 32 controls (KH_GEN_MAX_L: the generic kernels; the register-resident
        # families take 8); the reference takes any number (optimize.py:33-55, conversions.py:140-254).  More than that
        # runs the reference's own structure: the host loop around single-interval propagations (each of them on the
        # GPU) -- correct, and slow
        logger.warning("%d controls: the device sweeps take at most %d; running the host loop around single-step "
                       "propagations on the GPU", len(guess_pulses), _KH_MAX_CONTROLS)
        device_path = False

    if continue_from is None:
        result = Result()
        result.start_local_time = time.localtime()
    else:
        result = copy.deepcopy(continue_from)

    if device_path:
        backend = _HipBackend(objectives, pulses_mapping, tlist, len(guess_pulses), propagator, process_group)
    else:
        backend = _PluginBackend(
            objectives, adjoint_objectives, pulses_mapping, tlist, propagators, storage, parallel_map, mu, overlap
        )

    # ---- iteration 0: forward propagation under the guess (optimize.py:295-322)
    tic = time.time()
    if skip_initial_forward_propagation:
        if continue_from is not None:
            fw_states_T = list(continue_from.states)
        else:
            logger.warning(
                "You should not use `skip_initial_forward_propagation` unless you are also passing `continue_from`"
            )
            fw_states_T = [None for _ in objectives]
        if device_path:
            backend.fw_T_dev = None
        tau_vals = np.array([overlap(obj.target, s) for s, obj in zip(fw_states_T, objectives)])
    else:
        if device_path:
            fw_states_T, forward_states = backend.initial_forward(guess_pulses, store=second_order)
        else:
            fw_states_T, forward_states = backend.initial_forward(guess_pulses)
        tau_vals = backend.tau_vals(fw_states_T)
    toc = time.time()
    # the stored trajectories are only needed by the second-order update (optimize.py:324-329);
    # in iteration 0 the states under "optimized" and guess pulses coincide
    forward_states0 = forward_states = forward_states if second_order else None

    info = None
    optimized_pulses = copy.deepcopy(guess_pulses)
    static_args = dict(
        objectives=objectives, adjoint_objectives=adjoint_objectives, lambda_vals=lambda_vals,
        shape_arrays=shape_arrays, tlist=tlist, propagator=propagator, chi_constructor=chi_constructor,
        mu=mu, sigma=sigma, iter_start=iter_start, iter_stop=iter_stop,
    )
    if info_hook is not None:
        info = info_hook(
            backward_states=None, forward_states=forward_states, forward_states0=forward_states0,
            guess_pulses=guess_pulses,
            optimized_pulses=optimized_pulses, g_a_integrals=g_a_integrals, fw_states_T=fw_states_T,
            tau_vals=tau_vals, start_time=tic, stop_time=toc, iteration=0, info_vals=[], shared_data={},
            **static_args,
        )

    result.tlist = tlist
    result.objectives = objectives
    result.guess_controls = guess_controls
    result.optimized_controls = optimized_pulses
    result.controls_mapping = pulses_mapping
    if continue_from is None:
        if info is not None:
            result.info_vals.append(info)
        result.iters.append(0)
        result.iter_seconds.append(int(toc - tic))
        if not np.all(tau_vals == None):  # noqa: E711
            result.tau_vals.append(tau_vals)
        if store_all_pulses:
            result.all_pulses.append(guess_pulses)
    else:
        iter_start = continue_from.iters[-1]
        logger.info("Continuing from previous result, with iteration %d", iter_start + 1)
    result.states = fw_states_T

    # ---- main loop (optimize.py:392-581)
    for krotov_iteration in range(iter_start + 1, iter_stop + 1):
        logger.info("Started Krotov iteration %d", krotov_iteration)
        tic = time.time()

        chi_T = chi_coef = None
        if device_path and default_norm and backend.targets is not None:
            # built-in functionals: chi_k(T) is formed and normalised on the device from K scalars
            # (no K-long Python loop, phi_k(T) and chi_k(T) stay in HBM)
            if backend.fw_T_dev is not None or chi_constructor is _functionals.chis_re:
                chi_coef = _functionals.chi_coefficients(
                    chi_constructor, backend.weights, tau_vals, len(objectives))
        if chi_coef is not None:
            chi_states = chi_norms = None
        else:
            chi_states = chi_constructor(fw_states_T=fw_states_T, objectives=objectives, tau_vals=tau_vals)
            chi_norms = [norm(chi) for chi in chi_states]
            chi_states = [chi / nrm for chi, nrm in zip(chi_states, chi_norms)]
            if device_path:
                vecs = [state_to_vector(c, backend.N, backend.is_super) for c in chi_states]
                if any(v is None for v in vecs):
                    raise ValueError("chi_constructor returned states that do not match the state dimension")
                chi_T = np.array(vecs)

        g_a_integrals[:] = 0.0
        if device_path:
            backward_states, optimized_pulses, fw_states_T, g_a, forward_states = backend.iterate(
                chi_T, chi_norms, guess_pulses, lambda_vals, shape_arrays, sigma=sigma, chi_coef=chi_coef
            )
        else:
            backward_states, optimized_pulses, fw_states_T, g_a, forward_states = backend.iterate(
                chi_states, chi_norms, guess_pulses, lambda_vals, shape_arrays, sigma=sigma,
                forward_states0=forward_states0,
            )
        g_a_integrals[:] = g_a
        tau_vals = backend.tau_vals(fw_states_T)
        toc = time.time()

        if info_hook is not None:
            info = info_hook(
                backward_states=backward_states, forward_states=forward_states, forward_states0=forward_states0,
                fw_states_T=fw_states_T, guess_pulses=guess_pulses, optimized_pulses=optimized_pulses,
                g_a_integrals=g_a_integrals, tau_vals=tau_vals, start_time=tic, stop_time=toc,
                info_vals=result.info_vals, shared_data={}, iteration=krotov_iteration, **static_args,
            )
        result.iters.append(krotov_iteration)
        result.iter_seconds.append(int(toc - tic))
        if info is not None:
            result.info_vals.append(info)
        if not np.all(tau_vals == None):  # noqa: E711
            result.tau_vals.append(tau_vals)
        result.optimized_controls = optimized_pulses
        if store_all_pulses:
            result.all_pulses.append(copy.deepcopy(optimized_pulses))
        result.states = fw_states_T
        logger.info("Finished Krotov iteration %d", krotov_iteration)

        msg = None
        if check_convergence is not None:
            msg = check_convergence(result)
        if krotov_iteration >= static_args['iter_stop']:  # a hook may have changed it
            iter_stop = static_args['iter_stop']
            result.message = "Reached %d iterations" % iter_stop
            break
        if bool(msg) is True:
            result.message = "Reached convergence"
            if isinstance(msg, str):
                result.message += ": " + msg
            break
        guess_pulses = optimized_pulses
        if second_order:
            if chi_states is None:  # co-states formed on the device, in the caller's state type
                chi_T, chi_norms = backend.chi_host()
                chi_states = _LazyStates(chi_T, backend.likes)
            sigma.refresh(
                forward_states=forward_states, forward_states0=forward_states0, chi_states=chi_states,
                chi_norms=chi_norms, optimized_pulses=optimized_pulses, guess_pulses=guess_pulses,
                objectives=objectives, result=result,
            )
            forward_states0 = forward_states
            if device_path:
                backend.advance_second_order()
    else:
        result.message = "Reached %d iterations" % max(iter_start, iter_stop)

    # ---- finalize (optimize.py:583-590)
    result.end_local_time = time.localtime()
    result.optimized_controls = [pulse_onto_tlist(np.asarray(p)) for p in optimized_pulses]
    if isinstance(result.states, _LazyStates):
        result.states = list(result.states)
    return result
