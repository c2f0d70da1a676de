"""A stand-in for ``qutip.Qobj`` (TESTS ONLY; QuTiP is not installed in the build image).

It offers exactly the surface SURVEY.md appendix C lists -- what the reference touches on or next to the hot path
(`propagators.py:96-117, 255-273, 299-307`, `mu.py:130`, `second_order.py:69-83`, `objectives.py:82, 675-1051`,
`optimize.py:243, 438, 466`, `functionals.py:375-385, 592`) -- with QuTiP 4's semantics:

* data: ``.data`` (SciPy CSR), ``.full()``, ``.dims``, ``.shape``, ``.type`` (derived from ``dims``: 'ket', 'bra', 'oper',
  'super', 'operator-ket'), ``.isherm``;
* ``.dag()``, ``.conj()``, ``.trans()``, ``.tr()``, ``.norm(kind)``, ``.overlap(other)``, ``.expm()``;
* ``* + - /`` and unary minus with scalars and with each other (``0 + q`` works, so ``sum([...])`` does), ``==`` (same
  ``dims``, elements within 1e-12), ``q[i, j]``, ``q(state)`` (operator on ket; super-operator on a density matrix through
  column stacking).

Deliberately NOT offered: ``__array__`` / the buffer protocol (``np.asarray(q)`` raises), ``len``, iteration, hashing --
code that only works because an object happens to convert to an ndarray would pass with arrays and fail with QuTiP
versions that do not convert.  ``N_CONSTRUCTED`` counts constructor calls (tests assert that states come back through
the class's own constructor).
"""
import numbers

import numpy as np
import scipy.linalg
import scipy.sparse as sp

ATOL = 1e-12


def _flat(dims_side):
    out = []
    for d in dims_side:
        if isinstance(d, list):
            out.extend(_flat(d))
        else:
            out.append(d)
    return out


def _type_of(dims):
    rows, cols = dims
    nested_r = len(rows) > 0 and isinstance(rows[0], list)
    nested_c = len(cols) > 0 and isinstance(cols[0], list)
    if nested_r and nested_c:
        return 'super'
    if nested_r:
        return 'operator-ket'
    if nested_c:
        return 'operator-bra'
    r1 = all(d == 1 for d in rows)
    c1 = all(d == 1 for d in cols)
    if c1 and not r1:
        return 'ket'
    if r1 and not c1:
        return 'bra'
    return 'oper'


class QobjDouble:
    __array_priority__ = 100  # NumPy scalars defer to __rmul__ / __radd__ (as qutip.Qobj arranges)
    N_CONSTRUCTED = 0

    def __init__(self, inpt, dims=None):
        type(self).N_CONSTRUCTED += 1
        if isinstance(inpt, QobjDouble):
            mat = inpt.data.copy()
            dims = inpt.dims if dims is None else dims
        elif sp.issparse(inpt):
            mat = sp.csr_matrix(inpt, dtype=np.complex128)
        else:
            arr = np.array(inpt, dtype=np.complex128)
            if arr.ndim == 0:
                arr = arr.reshape(1, 1)
            elif arr.ndim == 1:
                arr = arr.reshape(-1, 1)  # a flat list is a ket, as in QuTiP
            if arr.ndim != 2:
                raise TypeError("QobjDouble needs a matrix or a vector")
            mat = sp.csr_matrix(arr)
        self.data = mat
        if dims is None:
            dims = [[mat.shape[0]], [mat.shape[1]]]
        dims = [list(dims[0]), list(dims[1])]
        if int(np.prod(_flat(dims[0]) or [1])) != mat.shape[0] or int(np.prod(_flat(dims[1]) or [1])) != mat.shape[1]:
            raise ValueError("dims %s do not match the shape %s" % (dims, mat.shape))
        self.dims = dims

    # -- what the reference reads --------------------------------------------
    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def type(self):
        return _type_of(self.dims)

    @property
    def isherm(self):
        if self.shape[0] != self.shape[1]:
            return False
        a = self.data.toarray()
        return bool(np.all(np.abs(a - a.conj().T) <= ATOL))

    def full(self, order='C'):
        return np.array(self.data.toarray(), dtype=np.complex128, order=order)

    def __array__(self, *args, **kwargs):
        raise TypeError("QobjDouble does not convert to an ndarray implicitly: read it through .full() / .data "
                        "(SURVEY.md appendix C)")

    # -- structure -----------------------------------------------------------
    def _new(self, mat, dims):
        return type(self)(mat, dims=dims)

    def dag(self):
        return self._new(self.data.conj().T.tocsr(), [self.dims[1], self.dims[0]])

    def conj(self):
        return self._new(self.data.conj(), self.dims)

    def trans(self):
        return self._new(self.data.T.tocsr(), [self.dims[1], self.dims[0]])

    def tr(self):
        val = complex(self.data.diagonal().sum())
        return val.real if self.isherm else val

    def norm(self, norm=None):
        a = self.data.toarray()
        if self.type in ('ket', 'bra', 'operator-ket', 'operator-bra'):
            kind = 'l2' if norm is None else norm
            if kind == 'l2':
                return float(np.linalg.norm(a.reshape(-1)))
            if kind == 'max':
                return float(np.abs(a).max())
            raise ValueError("vector norm must be 'l2' or 'max'")
        kind = 'tr' if norm is None else norm
        if kind == 'tr':
            return float(np.linalg.svd(a, compute_uv=False).sum())
        if kind == 'fro':
            return float(np.linalg.norm(a, 'fro'))
        if kind == 'one':
            return float(np.linalg.norm(a, 1))
        if kind == 'max':
            return float(np.abs(a).max())
        raise ValueError("operator norm must be 'tr', 'fro', 'one' or 'max'")

    def overlap(self, other):
        if not isinstance(other, QobjDouble):
            raise TypeError("can only calculate the overlap with another quantum object")
        ta, tb = self.type, other.type
        if ta in ('ket', 'operator-ket') and tb == ta:
            return complex((self.data.conj().T @ other.data).toarray()[0, 0])
        if ta == 'bra' and tb == 'bra':
            return complex((self.data @ other.data.conj().T).toarray()[0, 0])
        if ta == 'bra' and tb == 'ket':
            return complex((self.data @ other.data).toarray()[0, 0])
        if ta == 'ket' and tb == 'bra':
            return complex((self.data.conj().T @ other.data.conj().T).toarray()[0, 0])
        if ta == 'oper' and tb == 'oper':
            return complex((self.data.conj().T @ other.data).diagonal().sum())
        raise TypeError("cannot calculate the overlap of a %s with a %s" % (ta, tb))

    def expm(self):
        if self.shape[0] != self.shape[1]:
            raise TypeError("expm needs a square operator")
        return self._new(scipy.linalg.expm(self.data.toarray()), self.dims)

    # -- arithmetic -----------------------------------------------------------
    def __add__(self, other):
        if isinstance(other, QobjDouble):
            if self.dims != other.dims:
                raise TypeError("Incompatible quantum object dimensions")
            return self._new(self.data + other.data, self.dims)
        if isinstance(other, numbers.Number):
            if other == 0:
                return self._new(self.data.copy(), self.dims)
            if self.shape[0] != self.shape[1]:
                raise TypeError("a scalar can only be added to a square operator")
            return self._new(self.data + other * sp.identity(self.shape[0], dtype=np.complex128, format='csr'), self.dims)
        return NotImplemented

    __radd__ = __add__

    def __neg__(self):
        return self._new(-self.data, self.dims)

    def __sub__(self, other):
        return self + (-other)

    def __rsub__(self, other):
        return (-self) + other

    def __mul__(self, other):
        if isinstance(other, QobjDouble):
            if self.shape[1] != other.shape[0]:
                raise TypeError("Incompatible Qobj shapes")
            return self._new(self.data @ other.data, [self.dims[0], other.dims[1]])
        if isinstance(other, numbers.Number):
            return self._new(self.data * complex(other), self.dims)
        return NotImplemented

    def __rmul__(self, other):
        if isinstance(other, numbers.Number):
            return self._new(complex(other) * self.data, self.dims)
        return NotImplemented

    def __truediv__(self, other):
        if isinstance(other, numbers.Number):
            return self._new(self.data / complex(other), self.dims)
        return NotImplemented

    def __eq__(self, other):
        if not isinstance(other, QobjDouble):
            return False
        if self.dims != other.dims:
            return False
        diff = (self.data - other.data)
        return not np.any(np.abs(diff.data) > ATOL)

    def __ne__(self, other):
        return not self == other

    __hash__ = None

    def __getitem__(self, idx):
        return self.full()[idx]

    def __call__(self, other):
        if not isinstance(other, QobjDouble):
            raise TypeError("Only defined for quantum objects.")
        if self.type == 'oper':
            if other.type == 'ket':
                return self * other
            raise TypeError("Can only act oper on ket.")
        if self.type == 'super':
            if other.type == 'ket':
                other = other * other.dag()
            if other.type != 'oper':
                raise TypeError("Can only act super on oper or ket.")
            vec = other.full().ravel(order='F')  # column stacking (operator_to_vector)
            out = (self.data @ vec).reshape(other.shape, order='F')
            return self._new(out, other.dims)
        raise TypeError("not an operator")

    def __repr__(self):
        return "QobjDouble(dims=%s, shape=%s, type=%s)" % (self.dims, self.shape, self.type)


# -- the few constructors the tests need (qutip.ket / basis / tensor / liouvillian are NOT part of the double) -------


def ket(index, dim):
    """|index> of a ``dim``-level system; ``index`` / ``dim`` tuples give product states with tensor dims."""
    if isinstance(index, int):
        index, dim = (index,), (dim,)
    vec = np.ones(1, dtype=np.complex128)
    for i, d in zip(index, dim):
        e = np.zeros(d, dtype=np.complex128)
        e[i] = 1.0
        vec = np.kron(vec, e)
    return QobjDouble(vec.reshape(-1, 1), dims=[list(dim), [1] * len(dim)])


def oper(arr, dim=None):
    arr = np.asarray(arr, dtype=np.complex128)
    dim = [arr.shape[0]] if dim is None else list(dim)
    return QobjDouble(arr, dims=[dim, dim])


def super_oper(arr, dim):
    """An N = d*d super-operator on column-stacked density matrices of a system with dims ``dim``."""
    dim = list(dim)
    return QobjDouble(np.asarray(arr, dtype=np.complex128), dims=[[dim, dim], [dim, dim]])
