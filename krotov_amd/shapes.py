"""Update-shape / guess-pulse helper functions.

Same public names and argument meaning as the reference's ``krotov.shapes``
(reference src/krotov/shapes.py:20-174) so that user scripts written against it
keep working: ``qutip_callback``, ``zero_shape``, ``one_shape``, ``flattop``,
``box``, ``blackman``.  Host-side, evaluated once per control at problem setup.
"""
import functools

import numpy as np

__all__ = ['qutip_callback', 'zero_shape', 'one_shape', 'flattop', 'box', 'blackman']


def qutip_callback(func, **kwargs):
    """Wrap ``func(t, **params)`` as a QuTiP-style control ``f(t, args)``.

    Parameters given in ``kwargs`` are frozen; the remaining ones are taken at
    call time from the ``args`` dict (reference shapes.py:20-38).
    """
    frozen = functools.partial(func, **kwargs)

    def control(t, args):
        return frozen(t, **(args or {}))

    return control


def zero_shape(t):
    """S(t) = 0."""
    return 0


def one_shape(t):
    """S(t) = 1."""
    return 1


def box(t, t_start, t_stop):
    """1 on ``[t_start, t_stop]``, 0 outside (reference shapes.py:120-137)."""
    return 1.0 if (t_start <= t <= t_stop) else 0.0


def _blackman_value(t, t_start, t_stop, a, inside):
    # expression order kept identical to the reference's formula
    # (shapes.py:160-174) so that sampled guess pulses agree bitwise
    T = t_stop - t_start
    return (
        0.5
        * inside
        * (
            1.0
            - a
            - np.cos(2.0 * np.pi * (t - t_start) / T)
            + a * np.cos(4.0 * np.pi * (t - t_start) / T)
        )
    )


def blackman(t, t_start, t_stop, a=0.16):
    """Blackman window between ``t_start`` and ``t_stop`` (reference
    shapes.py:140-174); accepts a scalar or an array ``t``."""
    if np.ndim(t) == 0:
        return _blackman_value(t, t_start, t_stop, a, box(t, t_start, t_stop))
    t = np.asarray(t, dtype=np.float64)
    inside = ((t >= t_start) & (t <= t_stop)).astype(np.float64)
    return _blackman_value(t, t_start, t_stop, a, inside)


def _ramp_up(kind, t, t_start, t_rise):
    if kind == 'blackman':  # first half of a Blackman window of width 2*t_rise
        return blackman(t, t_start, t_start + 2 * t_rise)
    return np.sin(np.pi * (t - t_start) / (2.0 * t_rise)) ** 2


def _ramp_down(kind, t, t_stop, t_fall):
    if kind == 'blackman':  # second half of a Blackman window of width 2*t_fall
        return blackman(t, t_stop - 2 * t_fall, t_stop)
    return np.sin(np.pi * (t - t_stop) / (2.0 * t_fall)) ** 2


def flattop(t, t_start, t_stop, t_rise, t_fall=None, func='blackman'):
    """Flat-top shape: 0 outside ``[t_start, t_stop]``, ramps of duration
    ``t_rise`` / ``t_fall`` (half-Blackman or sine-squared), 1 in between
    (reference shapes.py:51-117)."""
    if func not in ('blackman', 'sinsq'):
        raise ValueError("Invalid func: %s" % func)
    if t_fall is None:
        t_fall = t_rise
    if t < t_start or t > t_stop:
        return 0.0
    if t <= t_start + t_rise:
        return _ramp_up(func, t, t_start, t_rise)
    if t >= t_stop - t_fall:
        return _ramp_down(func, t, t_stop, t_fall)
    return 1.0
