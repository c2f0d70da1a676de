"""dH/d(eps) for equations of motion that are linear in the controls.

Same signature as ``krotov.mu.derivative_wrt_pulse`` (reference
src/krotov/mu.py:74-140).  On the device this operator is never built per
call: the engine uses the staged control operator itself, times ``i`` for
Liouvillians (csrc/kh_generic.h, KhUpdateArgs::mu_re/mu_im).
"""
import numpy as np

from ._ingest import obj_type, to_dense

__all__ = ['derivative_wrt_pulse']


def derivative_wrt_pulse(objectives, i_objective, pulses, pulses_mapping, i_pulse, time_index):
    """Callable applying dH/d(eps_{i_pulse}) of objective ``i_objective``.

    Hilbert space: sum of the operators the control multiplies; Liouville
    space: ``i`` times that sum (the abstract H is ``i L``); a zero map if the
    control does not occur; time-dependent collapse operators raise
    NotImplementedError -- as in the reference (mu.py:123-140).
    """
    objective = objectives[i_objective]
    where = pulses_mapping[i_objective][0][i_pulse]
    for i_c_op in range(len(objective.c_ops)):
        if len(pulses_mapping[i_objective][i_c_op + 1][i_pulse]) != 0:
            raise NotImplementedError("Time-dependent collapse operators not implemented")
    if len(where) == 0:
        return lambda state: 0 * state
    first = objective.H[where[0]][0]
    is_super = obj_type(first) == 'super'
    total = first
    for i in where[1:]:
        total = total + objective.H[i][0]
    if is_super:
        total = 1j * total
    if hasattr(total, 'full') or callable(total):
        return total  # Qobj-like: callable on states
    dense = to_dense(total)
    if is_super:
        return lambda state: dense @ np.asarray(state)

    def apply(state):
        # array-typed operators carry no `.type`: Liouville space is recognised the way the device path
        # does it -- a square d x d density matrix under an operator of dimension d^2 -- and then
        # mu = i dL/d(eps) acts on the column-stacked vec(rho) (reference mu.py:130-134, propagators.py:307)
        st = np.asarray(state)
        if st.ndim == 2 and st.shape[0] == st.shape[1] and st.size == dense.shape[0] and st.shape[0] > 1:
            return (1j * (dense @ st.ravel('F'))).reshape(st.shape, order='F')
        return dense @ st

    return apply
