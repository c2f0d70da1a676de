"""Second-order update support and the inner product used by the optimisation.

Mirrors ``krotov.second_order`` (reference src/krotov/second_order.py):
:class:`Sigma` (9-66), ``_overlap`` (69-83), :func:`numerical_estimate_A`
(86-164).  In the kernels the second-order term is folded into the co-state:
``Im <chi + sigma/(2 ||chi||) (phi - phi_prev) | mu | phi>`` (krotov_hip.h,
``kh_set_second_order``).
"""
from abc import ABC, abstractmethod

import numpy as np

__all__ = ['Sigma', 'numerical_estimate_A']


class Sigma(ABC):
    """sigma(t) of the second-order update equation.

    Subclass it for the problem at hand: ``__call__(t)`` evaluates sigma at
    time ``t`` (it is sampled at the mid-point of every time interval), and
    ``refresh(...)`` -- called after every iteration but the last -- updates
    whatever sigma depends on parametrically (typically the constant A, see
    :func:`numerical_estimate_A`).  Pass an instance as ``sigma=`` to
    :func:`krotov_amd.optimize_pulses`.
    """

    @abstractmethod
    def __call__(self, t):  # pragma: nocover
        raise NotImplementedError()

    @abstractmethod
    def refresh(self, forward_states, forward_states0, chi_states, chi_norms, optimized_pulses, guess_pulses,
                objectives, result):  # pragma: nocover
        """Arguments as in the reference (second_order.py:31-66):
        ``forward_states[k][n]`` / ``forward_states0[k][n]`` are the states of
        objective k at ``tlist[n]`` under the optimized / guess pulses of the
        finished iteration, ``chi_states`` the normalised boundary co-states
        and ``chi_norms`` their original norms."""
        raise NotImplementedError()


def numerical_estimate_A(forward_states, forward_states0, chi_states, chi_norms, Delta_J_T):
    """New value of the second-order constant A:

        A = (sum_k 2 Re <chi_k(T)|dphi_k(T)> + Delta_J_T) / sum_k <dphi_k(T)|dphi_k(T)>

    with ``dphi_k(T) = forward_states[k][-1] - forward_states0[k][-1]`` and the
    un-normalised ``chi_k(T) = chi_norms[k] * chi_states[k]``; 0 when the states
    did not move (reference second_order.py:86-164).
    """
    K = len(forward_states0)
    moved = [forward_states[k][-1] - forward_states0[k][-1] for k in range(K)]
    denom = sum(_overlap(d, d).real for d in moved)
    if not denom > 1.0e-30:
        return 0
    numer = sum((2 * chi_norms[k] * _overlap(chi_states[k], moved[k])).real for k in range(K)) + Delta_J_T
    return numer / denom


def _overlap(a, b):
    """<a|b> for kets, tr(a^dag b) for operators; None if ``a``/``b`` are not
    compatible quantum objects (e.g. a target 'PE')."""
    ta, tb = getattr(a, 'type', None), getattr(b, 'type', None)
    if ta is not None and tb is not None:  # Qobj-like
        try:
            if ta == tb == 'oper':
                if getattr(a, 'isherm', False):
                    return complex((a * b).tr())
                return complex((a.dag() * b).tr())
            return a.overlap(b)
        except AttributeError:
            return None
    if isinstance(a, str) or isinstance(b, str) or a is None or b is None:
        return None
    try:
        va = np.asarray(a.full() if hasattr(a, 'full') else a, dtype=np.complex128)
        vb = np.asarray(b.full() if hasattr(b, 'full') else b, dtype=np.complex128)
    except (TypeError, ValueError):
        return None
    if va.size != vb.size:
        return None
    # vdot conjugates its first argument; for matrices this is tr(a^dag b)
    return complex(np.vdot(va.reshape(-1), vb.reshape(-1)))
