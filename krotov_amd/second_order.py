"""Inner product used by the optimisation (first-order method).

Mirrors ``krotov.second_order._overlap`` (reference src/krotov/second_order.py:
69-83).  The second-order ``Sigma`` machinery of the reference is outside the
hot path this package accelerates.
"""
import numpy as np

__all__ = ['_overlap', 'Sigma']


class Sigma:
    """Placeholder base class of second-order update functions (reference
    second_order.py:9-66).  ``optimize_pulses(sigma=...)`` is not supported by
    this package yet and raises NotImplementedError."""

    def __call__(self, t):
        raise NotImplementedError()

    def refresh(self, **kwargs):
        raise NotImplementedError()


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
