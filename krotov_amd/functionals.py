"""Final-time functionals and the chi boundary conditions derived from them.

Same names, signatures and values as ``krotov.functionals`` (reference
src/krotov/functionals.py:82-437): a ``chi_constructor`` receives
``fw_states_T``, ``objectives``, ``tau_vals`` and returns one co-state per
objective.  They run once per iteration on K states (O(K N) work).

On the device path the optimiser does not call the four constructors below: each
is ``chi_k = c_k target_k + d_k phi_k(T)`` with per-objective scalars
(:func:`chi_coefficients`) and is formed and normalised in HBM by
``kh_chi_boundary``.  :func:`chi_stacked` is the same thing as a (K, N) array
expression on the host; the list forms are what user code calls.
"""
import logging

import numpy as np

from .second_order import _overlap

__all__ = [
    'f_tau',
    'F_ss',
    'J_T_ss',
    'chis_ss',
    'F_sm',
    'J_T_sm',
    'chis_sm',
    'F_re',
    'J_T_re',
    'chis_re',
    'J_T_hs',
    'chis_hs',
    'F_avg',
    'gate',
    'mapped_basis',
]


def _weight(obj):
    return getattr(obj, 'weight', None)


def _taus(fw_states_T, objectives, tau_vals):
    if tau_vals is None:
        return [_overlap(obj.target, psi) for psi, obj in zip(fw_states_T, objectives)]
    return tau_vals


def f_tau(fw_states_T, objectives, tau_vals=None, **kwargs):
    """(1/K) sum_k w_k tau_k  (reference functionals.py:82-141); a tau that is
    None (no forward propagation yet) is skipped with a warning."""
    total = 0j
    for obj, tau in zip(objectives, _taus(fw_states_T, objectives, tau_vals)):
        if tau is None:
            logging.getLogger('krotov').warning("τ is None in f_tau")
            continue
        w = _weight(obj)
        total += tau if w is None else w * tau
    return total / len(objectives)


def F_ss(fw_states_T, objectives, tau_vals=None, **kwargs):
    """(1/K) sum_k w_k |tau_k|^2  (reference functionals.py:116-162)."""
    abssq = [abs(t) ** 2 for t in _taus(fw_states_T, objectives, tau_vals)]
    F = f_tau(fw_states_T, objectives, abssq)
    assert abs(complex(F).imag) < 1e-10
    return complex(F).real


def J_T_ss(fw_states_T, objectives, tau_vals=None, **kwargs):
    return 1 - F_ss(fw_states_T, objectives, tau_vals)


def F_sm(fw_states_T, objectives, tau_vals=None, **kwargs):
    return abs(f_tau(fw_states_T, objectives, tau_vals)) ** 2


def J_T_sm(fw_states_T, objectives, tau_vals=None, **kwargs):
    return 1 - F_sm(fw_states_T, objectives, tau_vals)


def F_re(fw_states_T, objectives, tau_vals=None, **kwargs):
    return complex(f_tau(fw_states_T, objectives, tau_vals)).real


def J_T_re(fw_states_T, objectives, tau_vals=None, **kwargs):
    return 1 - F_re(fw_states_T, objectives, tau_vals)


def J_T_hs(fw_states_T, objectives, tau_vals=None, **kwargs):
    """Hilbert-Schmidt distance functional (reference functionals.py:320-386)."""
    taus = _taus(fw_states_T, objectives, tau_vals)
    total = 0.0
    for rho, obj, tau in zip(fw_states_T, objectives, taus):
        w = _weight(obj)
        nr = complex(_overlap(rho, rho)).real
        nt = complex(_overlap(obj.target, obj.target)).real
        term = nr + nt - 2 * complex(tau).real
        total += term if w is None else w * term
    return total / (2 * len(objectives))


def chis_ss(fw_states_T, objectives, tau_vals):
    """chi_k = (tau_k / K) w_k target_k  (reference functionals.py:177-197)."""
    K = len(objectives)
    out = []
    for obj, tau in zip(objectives, tau_vals):
        w = _weight(obj)
        out.append((tau / K) * obj.target if w is None else (tau / K) * w * obj.target)
    return out


def chis_sm(fw_states_T, objectives, tau_vals):
    """chi_k = w_k / K^2 (sum_j w_j tau_j) target_k  (reference functionals.py:225-253)."""
    s = 0
    for obj, tau in zip(objectives, tau_vals):
        w = _weight(obj)
        s += tau if w is None else w * tau
    c = 1.0 / len(objectives) ** 2
    out = []
    for obj in objectives:
        w = _weight(obj)
        out.append(c * obj.target * s if w is None else c * w * obj.target * s)
    return out


def chis_re(fw_states_T, objectives, tau_vals):
    """chi_k = w_k / (2K) target_k  (reference functionals.py:293-317)."""
    c = 1.0 / (2 * len(objectives))
    out = []
    for obj in objectives:
        w = _weight(obj)
        out.append(c * obj.target if w is None else c * w * obj.target)
    return out


def chis_hs(fw_states_T, objectives, tau_vals):
    """chi_k = w_k / (2K) (rho_target_k - rho_k(T))  (reference functionals.py:389-437)."""
    c = 1.0 / (2 * len(objectives))
    out = []
    for obj, rho in zip(objectives, fw_states_T):
        w = _weight(obj)
        out.append(c * (obj.target - rho) if w is None else c * w * (obj.target - rho))
    return out


def chi_stacked(chi_constructor, targets, weights, fw_T, tau):
    """(K, N) array form of the four constructors above, or None for any other
    ``chi_constructor``.  ``targets``, ``fw_T``: (K, N); ``weights``: (K,) or
    None; ``tau``: (K,) complex.  Same operation order as the list forms."""
    K = targets.shape[0]
    w = np.ones(K) if weights is None else weights
    if chi_constructor is chis_re:
        return ((1.0 / (2 * K)) * w)[:, None] * targets
    if chi_constructor is chis_ss:
        return ((np.asarray(tau) / K) * w)[:, None] * targets
    if chi_constructor is chis_sm:
        s = 0
        for wk, t in zip(w, tau):
            s += wk * t
        return ((1.0 / K**2) * w)[:, None] * targets * s
    if chi_constructor is chis_hs:
        return ((1.0 / (2 * K)) * w)[:, None] * (targets - fw_T)
    return None


def chi_coefficients(chi_constructor, weights, tau, K):
    """Per-objective scalars ``(c, d)`` with ``chi_k = c_k target_k + d_k phi_k(T)``
    for the four constructors above (None for any other ``chi_constructor``):
    what the device-side co-state construction (``kh_chi_boundary``) needs from
    the host.  ``weights``: (K,) or None; ``tau``: (K,) complex or None."""
    w = np.ones(K) if weights is None else np.asarray(weights, dtype=np.float64)
    zero = np.zeros(K, dtype=np.complex128)
    if chi_constructor is chis_re:
        return ((1.0 / (2 * K)) * w).astype(np.complex128), zero
    if chi_constructor is chis_hs:
        c = ((1.0 / (2 * K)) * w).astype(np.complex128)
        return c, -c
    if tau is None or any(t is None for t in tau):
        return None
    tau = np.asarray(tau, dtype=np.complex128)
    if chi_constructor is chis_ss:
        return (tau / K) * w, zero
    if chi_constructor is chis_sm:
        s = 0
        for wk, t in zip(w, tau):
            s += wk * t
        return ((1.0 / K**2) * w) * s, zero
    return None


# ---------------------------------------------------------------------------
# gate analysis (reference functionals.py:440-641); not on the optimisation path
# ---------------------------------------------------------------------------
def _arr(x):
    return np.asarray(x.full() if hasattr(x, 'full') else x, dtype=np.complex128)


def mapped_basis(O, basis_states):
    """The states ``sum_i O[i, j] basis_states[i]`` for every column j of the gate ``O`` (reference
    functionals.py:614-641); a tuple, each state of the basis states' type."""
    O = _arr(O)
    return tuple(
        sum(complex(O[i, j]) * basis_states[i] for i in range(O.shape[0])) for j in range(O.shape[1])
    )


def gate(basis_states, fw_states_T):
    """The N x N matrix ``U[i, j] = <basis_states[i]|fw_states_T[j]>`` that maps the basis states to the
    propagated states (reference functionals.py:590-611; a NumPy array here, there is no Qobj to wrap it)."""
    N = len(basis_states)
    U = np.zeros((N, N), dtype=np.complex128)
    for j in range(N):
        for i in range(N):
            U[i, j] = _overlap(basis_states[i], fw_states_T[j])
    return U


def F_avg(fw_states_T, basis_states, gate, mapped_basis_states=None, prec=1e-5):
    """Average gate fidelity with respect to ``gate`` (N x N, in the logical subspace) from the propagated
    logical basis (reference functionals.py:440-587): for N propagated state vectors
    ``(|tr(O^+ U)|^2 + tr(O^+ U U^+ O)) / (N (N + 1))`` with U from :func:`gate`; for the N^2 propagated dyads
    ``rho_ij = DynMap[|phi_i><phi_j|]`` (density matrices, ordered i N + j) the Liouville-space sum
    ``sum_ij <O phi_i|rho_ij|O phi_j> + <O phi_i|rho_jj|O phi_i>`` over ``N (N + 1)``.  ``prec`` bounds the
    imaginary part that errors in the states may leave."""
    make_gate = globals()['gate']  # (the argument shadows the function of the same name, as in the reference)
    N = len(basis_states)
    O = _arr(gate)
    if O.shape != (N, N):
        raise ValueError("Shape of gate is incompatible with number of basis states")
    first = _arr(fw_states_T[0])
    is_operator = first.ndim == 2 and first.shape[0] == first.shape[1] and first.shape[0] > 1
    if is_operator:
        if len(fw_states_T) != N * N:
            raise ValueError(
                "Evaluating F_avg for density matrices requires %d states (forward-propagation of all dyadic "
                "combinations of %d basis states), not %d" % (N * N, N, len(fw_states_T)))
        if mapped_basis_states is None:
            mapped_basis_states = mapped_basis(O, basis_states)
        mapped = [_arr(v).reshape(-1) for v in mapped_basis_states]
        F = 0.0
        for j in range(N):
            rho_jj = _arr(fw_states_T[j * N + j])
            for i in range(N):
                rho_ij = _arr(fw_states_T[i * N + j])
                F += np.vdot(mapped[i], rho_ij @ mapped[j]) + np.vdot(mapped[i], rho_jj @ mapped[i])
    else:
        if len(fw_states_T) != N:
            raise ValueError(
                "Evaluating F_avg for hilbert space states requires %d states (forward-propagation of all basis "
                "states), not %d" % (N, len(fw_states_T)))
        U = make_gate(basis_states, fw_states_T)
        OdU = O.conj().T @ U
        F = abs(np.trace(OdU)) ** 2 + np.trace(OdU @ OdU.conj().T)
    F = complex(F)
    assert abs(F.imag) < prec, "%.2e > %.2e" % (F.imag, prec)
    return F.real / (N * (N + 1))
