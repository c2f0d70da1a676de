"""Propagator plugins for :func:`krotov_amd.optimize_pulses`.

Signature contract (reference src/krotov/propagators.py:13-47, 79, 125-159)::

    propagator(H, state, dt, c_ops=None, backwards=False, initialize=False) -> state

``H`` is a nested list with *scalar* control values.  :func:`expm` evaluates
``exp(f (H0 + sum_l eps_l H_l) dt) |state>`` with ``f = -i`` (Hilbert space),
``+i`` for ``backwards``, ``1`` for Liouvillians -- on the GPU, through the
same kernels as the batched sweeps (one objective, one interval).  There is no
CPU implementation in this package.

Passing :func:`expm` (or a :class:`HipExpm`) to ``optimize_pulses`` selects the
device-resident fast path: the whole backward sweep and the whole
forward/update sweep of an iteration are one kernel launch each, instead of
one Python call per objective per interval.
"""
from abc import ABC, abstractmethod

import numpy as np

from ._ingest import obj_type, state_to_vector, to_dense, vector_to_state

__all__ = ['expm', 'Propagator', 'HipExpm', 'DensityMatrixODEPropagator']


class Propagator(ABC):
    """Base class of stateful propagators (reference propagators.py:125-159).

    A propagator called with ``initialize=False`` may assume its input state is
    the result of its previous call.
    """

    @abstractmethod
    def __call__(self, H, state, dt, c_ops=None, backwards=False, initialize=False):
        pass


def _fold(H):
    """Sum a scalar-valued nested list into one dense operator; also report
    whether it is a super-operator (``.type == 'super'``)."""
    assert isinstance(H, list) and len(H) > 0
    total = None
    is_super = None
    for part in H:
        if isinstance(part, list):
            op, coeff = part[0], part[1]
        else:
            op, coeff = part, 1.0
        if is_super is None:
            is_super = obj_type(op) == 'super'
        term = complex(coeff) * to_dense(op)
        total = term if total is None else total + term
    return total, bool(is_super)


def _single_step(H, state, dt, c_ops, backwards, liouville):
    from .engine import HipKrotovEngine  # needs a GPU; raises otherwise

    if c_ops is None:
        c_ops = []
    if len(c_ops) > 0:
        raise NotImplementedError("Liouville exponentiation not implemented")
    A, is_super = _fold(H)
    if liouville is not None:
        is_super = bool(liouville)
    N = A.shape[0]
    st = obj_type(state)
    if st is not None and obj_type(H[0][0] if isinstance(H[0], list) else H[0]) is not None:
        ok = (st == 'oper' and is_super) or (st in ('ket', 'bra') and not is_super)
        if not ok:
            raise NotImplementedError(
                "Cannot handle argument types A:%s, state:%s" % ('super' if is_super else 'oper', st)
            )
    else:
        arr = np.asarray(state.full() if hasattr(state, 'full') else state)
        if liouville is None and arr.ndim == 2 and arr.shape[0] == arr.shape[1] and arr.size == N and arr.shape[0] > 1:
            is_super = True  # a density matrix handed to an N = d*d operator
    vec = state_to_vector(state, N, is_super)
    if vec is None:
        raise NotImplementedError("state of shape %s does not fit operator dimension %d" % (np.shape(state), N))
    # backwards in Hilbert space: exp(+i A dt) = exp(-i (-A) dt)
    op = -A if (backwards and not is_super) else A
    eng = HipKrotovEngine([[op]], [float(dt)], is_super=is_super)
    try:
        out = eng.forward(np.zeros((0, 1)), vec[None, :])
        res = out.cpu().numpy()[0]
    finally:
        eng.close()
    return vector_to_state(res, state)


def expm(H, state, dt, c_ops=None, backwards=False, initialize=False):
    """One interval of exact (machine-precision) time evolution on the GPU.

    Drop-in for ``krotov.propagators.expm`` (reference propagators.py:79-122):
    same arguments, ``c_ops`` unsupported (NotImplementedError), ``initialize``
    ignored.  For plain arrays the operators are taken to be Hamiltonians
    unless the state is a square density matrix; use :class:`HipExpm` with
    ``liouville=True`` for pre-vectorised Liouville-space problems.
    """
    return _single_step(H, state, dt, c_ops, backwards, None)


class HipExpm(Propagator):
    """:func:`expm` as a :class:`Propagator` object with an explicit
    ``liouville`` switch (None = infer from ``.type`` / state shape).  With
    ``sparse=True`` ``optimize_pulses`` keeps the operators in CSR form on the
    device (large operators with a few entries per row); results are the same
    to round-off."""

    def __init__(self, liouville=None, sparse=False):
        self.liouville = liouville
        self.sparse = bool(sparse)

    def __call__(self, H, state, dt, c_ops=None, backwards=False, initialize=False):
        return _single_step(H, state, dt, c_ops, backwards, self.liouville)


class DensityMatrixODEPropagator(HipExpm):
    """Drop-in for ``krotov.propagators.DensityMatrixODEPropagator`` (reference
    propagators.py:162-327): density matrices under a sparse Liouvillian
    ``d/dt vec(rho) = L vec(rho)`` (``H`` is the Liouvillian in nested-list form,
    ``c_ops`` empty).  The reference integrates with SciPy's ``zvode`` to
    ``rtol``/``atol``; here every interval is the exponential action of the
    sparse generator to machine precision (CSR matrix-vector products inside the
    same sweep kernels), so the integrator options are accepted for
    compatibility and have no effect, and the object holds no state
    (``reentrant`` is moot: one instance serves any number of objectives)."""

    def __init__(self, method='adams', order=12, atol=1e-8, rtol=1e-6, nsteps=1000, first_step=0, min_step=0,
                 max_step=0, reentrant=False):
        super().__init__(liouville=True, sparse=True)
        self.method, self.order, self.atol, self.rtol = method, order, atol, rtol
        self.nsteps, self.first_step, self.min_step, self.max_step = nsteps, first_step, min_step, max_step
        self.reentrant = reentrant
