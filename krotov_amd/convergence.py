"""``check_convergence`` routines for :func:`krotov_amd.optimize_pulses`.

Same names, arguments, return values and messages as ``krotov.convergence``
(reference src/krotov/convergence.py:84-419): a ``check_convergence(result)``
returns None to continue, or a message (anything true) to stop.  Host
bookkeeping around the accelerated path.

Where the reference takes a ``glom`` spec to pick a value out of the
:class:`~krotov_amd.result.Result`, this module takes

* a callable ``spec(result)``, or
* a path: a tuple/list whose elements are applied in turn -- a string is an
  attribute (or mapping key), an integer an index, a callable is called --
  e.g. ``('info_vals', -1)``, the default, is ``result.info_vals[-1]``.

(``glom.T[-1]``-style specs of existing scripts work when ``glom`` is installed:
anything else is handed to ``glom.glom``.)
"""
import logging

__all__ = [
    'Or', 'value_below', 'value_above', 'delta_below', 'check_monotonic_error', 'check_monotonic_fidelity',
    'dump_result',
]

_LOOKUP_ERRORS = (AttributeError, KeyError, IndexError, TypeError)


def _extract(result, spec, **kwargs):
    if callable(spec):
        return spec(result)
    if isinstance(spec, str):
        spec = (spec,)
    if isinstance(spec, (tuple, list)) and all(isinstance(p, (str, int)) or callable(p) for p in spec):
        value = result
        for part in spec:
            if callable(part):
                value = part(value)
            elif isinstance(part, int):
                value = value[part]
            elif isinstance(value, dict):
                value = value[part]
            else:
                value = getattr(value, part)
        return value
    import glom  # a genuine glom spec

    return glom.glom(result, spec, **kwargs)


def Or(*funcs):
    """The first true result among ``funcs(result)``, else None."""

    def check_convergence(result):
        for func in funcs:
            msg = func(result)
            if bool(msg) is True:
                return msg
        return None

    return check_convergence


def _threshold(limit, spec, name, below, **kwargs):
    # `limit` may be a string so that the message shows it as written ("1e-4", not 0.0001)
    label = str(spec) if name is None else name

    def check_convergence(result):
        value = _extract(result, spec, **kwargs)
        if below:
            return "%s < %s" % (label, limit) if value < float(limit) else None
        return "%s > %s" % (label, limit) if value > float(limit) else None

    return check_convergence


def value_below(limit, spec=('info_vals', -1), name=None, **kwargs):
    """Stop when the value picked by ``spec`` (default: the last ``info_vals``
    entry, e.g. J_T) drops below ``limit``; message ``"<name> < <limit>"``."""
    return _threshold(limit, spec, name, True, **kwargs)


def value_above(limit, spec=('info_vals', -1), name=None, **kwargs):
    """Stop when the value exceeds ``limit`` (for fidelities); message
    ``"<name> > <limit>"``."""
    return _threshold(limit, spec, name, False, **kwargs)


def delta_below(limit, spec1=('info_vals', -1), spec0=('info_vals', -2), absolute_value=True, name=None, **kwargs):
    """Stop when the change ``spec1 - spec0`` (default: between the last two
    ``info_vals``; its absolute value unless ``absolute_value=False``) is below
    ``limit``.  While only one of the two values exists (first iteration) the
    check passes; if neither can be read the lookup error is raised."""
    label = "Δ(%s,%s)" % (spec1, spec0) if name is None else name

    def check_convergence(result):
        values, failure = [], None
        for spec in (spec1, spec0):
            try:
                values.append(_extract(result, spec, **kwargs))
            except _LOOKUP_ERRORS as exc:
                values.append(None)
                failure = exc
        if (values[0] is None) != (values[1] is None):
            return None
        if failure is not None:
            raise failure
        delta = values[0] - values[1]
        if absolute_value:
            delta = abs(delta)
        return "%s < %s" % (label, limit) if delta < float(limit) else None

    return check_convergence


_error_decrease = delta_below(
    limit=0, spec1=('info_vals', -2), spec0=('info_vals', -1), absolute_value=False,
    name="Loss of monotonic convergence; error decrease",
)
_fidelity_increase = delta_below(
    limit=0, spec1=('info_vals', -1), spec0=('info_vals', -2), absolute_value=False,
    name="Loss of monotonic convergence; fidelity increase",
)


def check_monotonic_error(result):
    """Message ``'Loss of monotonic convergence; error decrease < 0'`` if the last
    ``info_vals`` entry (an error such as J_T) is larger than the one before."""
    return _error_decrease(result)


def check_monotonic_fidelity(result):
    """The same for ``info_vals`` that should grow (a fidelity)."""
    return _fidelity_increase(result)


def dump_result(filename, every=10, reference=False):
    """A ``check_convergence`` that never stops the optimisation but writes
    ``result.dump(filename.format(iter=...))`` every ``every`` iterations; if the
    file cannot be written the message ``"Could not store <file>: <error>"`` is
    returned, which ends the optimisation.  ``reference=True`` (extension): the
    files are written in the reference's own dump format (:meth:`.Result.dump`)."""
    every = int(every)
    if every <= 0:
        raise ValueError("every must be > 0")

    def _dump_result(result):
        iteration = result.iters[-1]
        if iteration % every == 0:
            outfile = filename.format(iter=iteration)
            logging.getLogger('krotov').info("Dumping result to %s", outfile)
            try:
                result.dump(outfile, reference=True) if reference else result.dump(outfile)
            except IOError as exc_info:
                return "Could not store %s: %s" % (outfile, exc_info)
        return None

    return _dump_result
