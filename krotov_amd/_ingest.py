"""Duck-typed ingestion of operators and states.

QuTiP is never imported.  Anything that looks like a ``qutip.Qobj`` (has
``.full()``; optionally ``.type``, ``.dims``) or is array-like is accepted.
Ingest rule: operator -> dense row-major complex128
``N x N``; ket/bra -> length-N vector; density matrix under a Liouvillian ->
``rho.full().ravel('F')`` (column stacking, reference propagators.py:255-257,
306-307).
"""
import numpy as np


def to_dense(op):
    """Dense complex128 2-D array of an operator-like object."""
    if hasattr(op, 'full'):
        arr = op.full()
    elif hasattr(op, 'toarray'):
        arr = op.toarray()
    else:
        arr = op
    arr = np.asarray(arr, dtype=np.complex128)
    if arr.ndim != 2 or arr.shape[0] != arr.shape[1]:
        raise ValueError("operator must be a square matrix, got shape %s" % (arr.shape,))
    return arr


def to_sparse(op):
    """``scipy.sparse.csr_matrix`` (complex128) of an operator-like object: a
    SciPy sparse matrix, a Qobj-like whose ``.data`` is one (QuTiP 4), or
    anything :func:`to_dense` accepts."""
    import scipy.sparse as sp

    data = getattr(op, 'data', None)
    if sp.issparse(op):
        mat = op
    elif data is not None and sp.issparse(data):
        mat = data
    else:
        mat = to_dense(op)
    mat = sp.csr_matrix(mat, dtype=np.complex128)
    if mat.shape[0] != mat.shape[1]:
        raise ValueError("operator must be a square matrix, got shape %s" % (mat.shape,))
    return mat


def obj_type(x):
    """``x.type`` if it has one ('ket', 'bra', 'oper', 'super'), else None."""
    return getattr(x, 'type', None)


def state_array(state):
    if hasattr(state, 'full'):
        return np.asarray(state.full(), dtype=np.complex128)
    return np.asarray(state, dtype=np.complex128)


def state_to_vector(state, dim, is_super):
    """Flatten ``state`` to the length-``dim`` vector the engine propagates.

    Kets / bras / 1-D or (dim, 1) arrays are raveled; a square ``d x d`` matrix
    with ``d*d == dim`` under a Liouvillian is column-stacked.  Returns None if
    the object cannot be a state of that dimension (e.g. the target 'PE').
    """
    if state is None or isinstance(state, str):
        return None
    try:
        arr = state_array(state)
    except (TypeError, ValueError):
        return None
    if arr.size != dim:
        return None
    if arr.ndim == 2 and arr.shape[0] == arr.shape[1] and arr.shape[0] > 1:
        if not is_super:
            return None  # an operator cannot be a Hilbert-space state
        return arr.ravel(order='F').copy()
    return arr.reshape(-1).copy()


def vector_to_state(vec, like):
    """Inverse of :func:`state_to_vector`: same kind of object as ``like``."""
    vec = np.asarray(vec, dtype=np.complex128)
    ref = state_array(like)
    if ref.ndim == 2 and ref.shape[0] == ref.shape[1] and ref.shape[0] > 1:
        arr = vec.reshape(ref.shape, order='F')
    else:
        arr = vec.reshape(ref.shape)
    if hasattr(like, 'full'):
        # Qobj-like: rebuild through its own constructor (qutip.Qobj(arr, dims=...))
        try:
            return like.__class__(arr, dims=like.dims)
        except Exception:
            try:
                return like.__class__(arr)
            except Exception:
                return arr
    return arr
