"""Controls <-> pulses and nested-list bookkeeping (host side).

Same public names and semantics as the reference's ``krotov.conversions``
(reference src/krotov/conversions.py) because these numbers define what every
sweep consumes: "controls" live on the points of ``tlist``, "pulses" on its
``nt-1`` intervals.  Controls are identified by object identity.
"""
import logging

import numpy as np

__all__ = [
    'control_onto_interval',
    'discretize',
    'extract_controls',
    'extract_controls_mapping',
    'plug_in_pulse_values',
    'pulse_onto_tlist',
    'pulse_options_dict_to_list',
]

def _real_scalar(v):
    """``float(v)``; complex-typed values are rejected like the reference does
    (conversions.py:103, 121-128: ``float()`` of a complex raises TypeError, and
    numpy's ComplexWarning is promoted to an error)."""
    if isinstance(v, complex) or np.iscomplexobj(v):
        raise TypeError("can't convert complex to float")
    return float(v)


def _tlist_midpoints(tlist):
    """Mid-points of the intervals of ``tlist`` (reference conversions.py:35-40)."""
    tlist = np.asarray(tlist, dtype=np.float64)
    return 0.5 * (tlist[1:] + tlist[:-1])


def pulse_onto_tlist(pulse):
    """Interval values -> grid-point values: end points are kept, interior
    points are the mean of the two neighbouring intervals
    (reference conversions.py:368-390)."""
    pulse = np.asarray(pulse)
    control = np.empty(len(pulse) + 1, dtype=pulse.dtype.type)
    control[0] = pulse[0]
    control[-1] = pulse[-1]
    if len(pulse) > 1:
        control[1:-1] = 0.5 * (pulse[:-1] + pulse[1:])
    return control


def control_onto_interval(control):
    """Grid-point values -> interval values, the inverse of
    :func:`pulse_onto_tlist`: ``p[0]=c[0]``, ``p[i]=2c[i]-p[i-1]``,
    ``p[-1]=c[-1]`` (reference conversions.py:333-365)."""
    if not isinstance(control, np.ndarray):
        raise ValueError("Not implemented: control type %s" % control.__class__.__name__)
    assert control.ndim == 1
    n = len(control) - 1
    pulse = np.zeros(n, dtype=control.dtype.type)
    pulse[0] = control[0]
    prev = pulse[0]
    for i in range(1, n):  # a recurrence: evaluated in order for bitwise parity
        prev = 2.0 * control[i] - prev
        pulse[i] = prev
    pulse[-1] = control[-1]
    return pulse


def discretize(control, tlist, args=(None,), kwargs=None, via_midpoints=False):
    """Values of ``control`` on ``tlist`` (reference conversions.py:61-137).

    A callable is evaluated as ``control(t, *args, **kwargs)``; with
    ``via_midpoints`` it is sampled at the interval mid-points (first and last
    sample at the grid boundaries) and un-averaged with
    :func:`pulse_onto_tlist`.  An array must already have ``len(tlist)``
    entries.  Complex values raise.
    """
    if callable(control):
        kwargs = {} if kwargs is None else kwargs
        if via_midpoints:
            mid = (tlist + 0.5 * (tlist[1] - tlist[0]))[:-1]
            mid[0] = tlist[0]
            mid[-1] = tlist[-1]
            return pulse_onto_tlist(discretize(control, mid, args=args, kwargs=kwargs))
        return np.array([_real_scalar(control(t, *args, **kwargs)) for t in tlist], dtype=np.float64)
    if isinstance(control, (np.ndarray, list)):
        values = np.array([_real_scalar(v) for v in control], dtype=np.float64)
        if len(values) != len(tlist):
            raise ValueError("If control is an array, it must of the same length as tlist")
        return values
    raise TypeError("control must be either a callable func(t, args) or a numpy array")


def _index_of(obj, seq):
    """Position of ``obj`` in ``seq``; arrays compare by identity."""
    if isinstance(obj, np.ndarray):
        for i, other in enumerate(seq):
            if other is obj:
                return i
        return -1
    try:
        return seq.index(obj)
    except ValueError:
        return -1


def extract_controls(objectives):
    """Unique controls of all objectives' ``H``, in order of first occurrence
    (reference conversions.py:140-164)."""
    controls = []
    for objective in objectives:
        for term in objective.H:
            if isinstance(term, list):
                assert len(term) == 2
                if _index_of(term[1], controls) < 0:
                    controls.append(term[1])
    return controls


def _positions(nested, control):
    return [i for i, term in enumerate(nested) if isinstance(term, list) and term[1] is control]


def extract_controls_mapping(objectives, controls):
    """``mapping[i_obj][0 = H | 1 + i_c_op][i_control] -> [indices]``
    (reference conversions.py:179-254)."""
    mapping = []
    for objective in objectives:
        per_objective = [[_positions(objective.H, c) for c in controls]]
        for c_op in objective.c_ops:
            per_objective.append([_positions(c_op, c) for c in controls])
        mapping.append(per_objective)
    return mapping


def pulse_options_dict_to_list(pulse_options, controls):
    """Options dict per control, in the order of ``controls``; array controls
    are keyed by ``id(control)`` (reference conversions.py:257-285)."""
    if len(pulse_options) > len(controls):
        logging.getLogger('krotov').warning(
            "pulse_options contains extra elements that are not in `controls`"
        )
    out = []
    for control in controls:
        try:
            try:
                out.append(pulse_options[control])
            except TypeError:  # unhashable ndarray
                out.append(pulse_options[id(control)])
        except KeyError:
            raise ValueError("The control %s does not have any associated pulse options" % str(control))
    return out


def plug_in_pulse_values(H, pulses, mapping, time_index, conjugate=False):
    """Copy of the nested list ``H`` with every control replaced by its pulse
    value on interval ``time_index`` (reference conversions.py:288-330)."""
    if isinstance(H, list):
        H = [list(term) if isinstance(term, list) else term for term in H]
    for pulse, positions in zip(pulses, mapping):
        for i in positions:
            value = pulse[time_index]
            H[i][1] = np.conjugate(value) if conjugate else value
    return H
