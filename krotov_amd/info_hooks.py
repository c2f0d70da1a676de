"""``info_hook`` helpers: :func:`chain` and the iteration table printer.

Mirrors ``krotov.info_hooks`` (reference src/krotov/info_hooks.py): ``chain``
(24-56) and ``print_table`` (352-621) with the same keyword arguments, column
layout and text, so that logs of existing scripts look the same (compared
against the reference's own ``tests/test_krotov/oct.log`` in
tests/test_host_helpers.py).  An ``info_hook`` receives the keyword arguments
listed in reference info_hooks.py:59-86; ``print_debug_information``
(59-291) writes the same report as the reference's, with state sizes and norms
taken from arrays instead of Qobj internals.  Host bookkeeping (SURVEY.md 8f,
rank 4).
"""
import sys
import time
import unicodedata

import numpy as np

__all__ = ['chain', 'print_table', 'print_debug_information']


def chain(*hooks):
    """Call ``hooks`` in order with the same keyword arguments; the results
    that are not None are returned as a tuple (None if there are none, the bare
    value if there is exactly one)."""

    def info_hook(**kwargs):
        results = []
        for hook in hooks:
            res = hook(**kwargs)
            if res is not None:
                results.append(res)
        if len(results) == 0:
            return None
        if len(results) == 1:
            return results[0]
        return tuple(results)

    return info_hook


def _width(text):
    """Number of character cells of ``text`` (combining marks take none)."""
    return sum(1 for ch in str(text) if unicodedata.combining(ch) == 0)


def _right(text, width):
    text = str(text)
    return ' ' * max(0, width - _width(text)) + text


class _PerPulseHeader:
    """Header of the per-pulse g_a column: ∫gₐ(ϵ₁)dt, ∫gₐ(ϵ₂)dt, ..."""

    _SUB = '₀₁₂₃₄₅₆₇₈₉'

    def format(self, l):
        return "∫gₐ(ϵ%s)dt" % ''.join(self._SUB[int(d)] for d in str(l))


# column order: iteration, J_T, g_a integral per pulse, sum of g_a integrals, J, Delta J_T, Delta J, seconds
_DEFAULTS = {
    True: dict(headers=["iter.", "J_T", _PerPulseHeader(), "∑∫gₐ(t)dt", "J", "ΔJ_T", "ΔJ", "secs"],
               single="∫gₐ(t)dt", min_widths=[5, 9, 12, 12, 11, 11, 11, 6]),
    False: dict(headers=["iter.", "J_T", "g_a_int_{l}", "g_a_int", "J", "Delta J_T", "Delta J", "secs"],
                single="g_a_int", min_widths=[5, 9, 11, 11, 11, 11, 11, 6]),
}
_SAMPLE = [10, 1e-15, 1e-15, 1e-15, 1e-15, -1e-15, -1e-15, 30]  # widest values a column has to hold


def print_table(*, J_T, show_g_a_int_per_pulse=False, J_T_prev=None, unicode=True,
                col_formats=('%d', '%.2e', '%.2e', '%.2e', '%.2e', '%.2e', '%.2e', '%d'), col_headers=None,
                out=sys.stdout):
    """``info_hook`` that writes one table row per iteration to ``out`` and
    returns the value of J_T (so it ends up in ``Result.info_vals``)::

        iter.      J_T    ∫gₐ(t)dt          J       ΔJ_T         ΔJ  secs
        0     1.00e+00    0.00e+00   1.00e+00        n/a        n/a     0
        1     7.65e-01    1.18e-01   8.82e-01  -2.35e-01  -1.18e-01     2

    Columns: iteration; ``J_T(**kwargs)``; with ``show_g_a_int_per_pulse`` and
    more than one pulse, the integral of g_a for each pulse; their sum; J = J_T +
    that sum; ΔJ_T with respect to ``J_T_prev(**kwargs)`` (default: the last
    ``info_vals`` entry); ΔJ = ΔJ_T + the sum; wall-clock seconds.  ``*`` / ``**``
    after the row flag a loss of monotonic convergence in ΔJ_T and/or ΔJ.
    ``col_formats`` / ``col_headers`` are 8-tuples (the third header must
    support ``.format(l=...)``); widths follow the headers and formats.
    """
    if J_T_prev is None:
        def J_T_prev(**kwargs):
            try:
                return kwargs['info_vals'][-1]
            except IndexError:
                return 0

    defaults = col_headers is None
    if defaults:
        headers = list(_DEFAULTS[bool(unicode)]['headers'])
        single = _DEFAULTS[bool(unicode)]['single']
        min_widths = list(_DEFAULTS[bool(unicode)]['min_widths'])
    else:
        headers, single = list(col_headers), None
        min_widths = [2, 4, 4, 4, 4, 4, 4, 3]  # room for "n/a"
    formats = list(col_formats)
    if len(formats) != 8 or len(headers) != 8:
        raise ValueError("col_formats, and col_headers must each have exactly 8 elements")
    try:
        per_pulse_sample = headers[2].format(l=10)
    except (AttributeError, NameError, TypeError, KeyError) as exc_info:
        raise ValueError(
            "The third label %r in col_headers must support '.format(l=l)' where l is an integer: %r"
            % (headers[2], exc_info)
        )
    try:
        samples = [fmt % val for fmt, val in zip(formats, _SAMPLE)]
    except TypeError:
        raise ValueError(
            "Invalid col_formats %r: Each element must specify a percent format string for a single value"
            % (col_formats,)
        )
    except ValueError as exc_info:
        raise ValueError("Invalid col_formats %r: %s" % (col_formats, exc_info))
    if show_g_a_int_per_pulse:  # the per-pulse columns are sized by their own header, not the default minimum
        min_widths[2] = max(_width(samples[2]) + 1, _width(per_pulse_sample) + 1)
    widths = []
    for i in range(8):
        label_w = _width(headers[i]) if isinstance(headers[i], str) else 0
        widths.append(max(min_widths[i], _width(samples[i]) + 1, label_w + 1))
    if defaults and formats[0] == '%d':
        widths[0] = 5  # the layout of the reference's published examples

    def info_hook(**kwargs):
        iteration = kwargs['iteration']
        g_a = kwargs['g_a_integrals']
        n_pulses = len(kwargs['guess_pulses'])
        w = list(widths)
        w[0] = max(w[0], len(str(kwargs['iter_stop'])) + 1)
        w[2] = max(w[2], _width(headers[2].format(l=n_pulses)) + 1)
        per_pulse = show_g_a_int_per_pulse and n_pulses > 1
        if iteration == 0:
            cells = [str(headers[0]).ljust(w[0]), _right(headers[1], w[1])]
            if per_pulse:
                cells += [_right(headers[2].format(l=l + 1), w[2]) for l in range(n_pulses)]
            cells.append(_right(single if (n_pulses == 1 and defaults) else headers[3], w[3]))
            cells += [_right(headers[i], w[i]) for i in (4, 5, 6, 7)]
            out.write(''.join(cells) + "\n")
        J_T_val = J_T(**kwargs)
        g_a_sum = np.sum(g_a)
        cells = [str(formats[0] % iteration).ljust(w[0]), _right(formats[1] % J_T_val, w[1])]
        if per_pulse:
            cells += [_right(formats[3] % g_a[i], w[2]) for i in range(n_pulses)]
        cells += [_right(formats[3] % g_a_sum, w[3]), _right(formats[4] % (J_T_val + g_a_sum), w[4])]
        flags = ''
        if iteration == 0:
            cells += [_right("n/a", w[5]), _right("n/a", w[6])]
        else:
            d_J_T = J_T_val - J_T_prev(**kwargs)
            d_J = d_J_T + g_a_sum
            cells += [_right(formats[5] % d_J_T, w[5]), _right(formats[6] % d_J, w[6])]
            flags = ('*' if d_J_T > 0 else '') + ('*' if d_J > 0 else '')
        secs = int(kwargs['stop_time'] - kwargs['start_time'])
        cells.append(" " + _right(formats[7] % secs, w[7] - 1))
        out.write(''.join(cells) + (" " + flags if flags else "") + "\n")
        out.flush()
        return J_T_val

    return info_hook



def _range_text(pulse):
    """'[min, max]' of a pulse (reference info_hooks.py:624-641: real and imaginary ranges if complex)."""
    pulse = np.asarray(pulse)
    if np.iscomplexobj(pulse):
        return '[(r:%.2f, i:%.2f), (r:%.2f, i:%.2f)]' % (
            pulse.real.min(), pulse.imag.min(), pulse.real.max(), pulse.imag.max())
    return '[%.2f, %.2f]' % (pulse.min(), pulse.max())


def _state_nbytes(state):
    """Memory of one state: the array's bytes (the reference estimates a Qobj's, info_hooks.py:12-21)."""
    return np.asarray(state).nbytes


def _storage_text(states, mb_per_slot):
    if states is None:
        return 'None'
    first = states[0]
    return '[%d * %s(%d)] (%.1f MB)' % (len(states), first.__class__.__name__, len(first), len(first) * mb_per_slot)


def print_debug_information(*, objectives, adjoint_objectives, backward_states, forward_states, forward_states0,
                            guess_pulses, optimized_pulses, g_a_integrals, lambda_vals, shape_arrays, fw_states_T,
                            tlist, tau_vals, start_time, stop_time, iteration, info_vals, shared_data, propagator,
                            chi_constructor, mu, sigma, iter_start, iter_stop, out=sys.stdout):
    """``info_hook`` with the full keyword signature (reference info_hooks.py:59-86) that writes a report of the
    iteration to ``out``: the set-up once (iteration 0), then duration, pulse ranges, the g_a integrals, lambda_a,
    what is stored of the propagated states, the norms of the final states and the overlaps tau.  Same lines
    and formats as the reference's (info_hooks.py:172-291); use :func:`functools.partial` to pass ``out``."""
    w = out.write
    w('Iteration %d\n' % iteration)
    if iteration == 0:
        w('    objectives:\n')
        for i, obj in enumerate(objectives):
            w('        %d:%s\n' % (i + 1, obj))
        w('    adjoint objectives:\n')
        for i, obj in enumerate(adjoint_objectives):
            w('        %d:%s\n' % (i + 1, obj))
        if isinstance(propagator, (list, tuple)):
            names = [getattr(f, '__name__', f.__class__.__name__) for f in propagator]
            w('    propagator: (%s)\n' % ', '.join(names))
        elif hasattr(propagator, '__name__'):
            w('    propagator: %s\n' % propagator.__name__)
        for label, fn in (('chi_constructor', chi_constructor), ('mu', mu)):
            if hasattr(fn, '__name__'):
                w('    %s: %s\n' % (label, fn.__name__))
        if sigma is not None:
            w('    sigma: %s\n' % sigma.__class__.__name__)
        w('    S(t) (ranges): %s\n' % ', '.join('[%f, %f]' % (np.min(S), np.max(S)) for S in shape_arrays))
        w('    iter_start: %s\n' % iter_start)
        w('    iter_stop: %s\n' % iter_stop)
    w('    duration: %.1f secs (started at %s)\n' % (
        stop_time - start_time, time.strftime('%Y-%m-%d %H:%M:%S', time.localtime(start_time))))
    w('    optimized pulses (ranges): %s\n' % ', '.join(_range_text(pulse) for pulse in optimized_pulses))
    w('    \u222bg\u2090(t)dt: %s\n' % ', '.join('%.2e' % v for v in g_a_integrals))
    w('    \u03bb\u2090: %s\n' % ', '.join('%.2e' % v for v in lambda_vals))
    mb_per_slot = 0.0  # (no final states, e.g. skip_initial_forward_propagation)
    if fw_states_T is not None:
        mb_per_slot = sum(_state_nbytes(state) for state in fw_states_T) / 1024**2
    w('    storage (bw, fw, fw0): %s, %s, %s\n' % (
        _storage_text(backward_states, mb_per_slot), _storage_text(forward_states, mb_per_slot),
        _storage_text(forward_states0, mb_per_slot)))
    if fw_states_T is not None:
        norms = [state.norm() if hasattr(state, 'norm') else np.linalg.norm(np.asarray(state)) for state in fw_states_T]
        w('    fw_states_T norm: %s\n' % ', '.join('%f' % v for v in norms))
    if tau_vals is not None and not any(z is None for z in tau_vals):
        w('    \u03c4: %s\n' % ', '.join('(%.2e:%.2f\u03c0)' % (abs(z), np.angle(z) / np.pi) for z in tau_vals))
    out.flush()
