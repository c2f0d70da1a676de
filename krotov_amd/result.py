"""Record of an optimisation, field-compatible with ``krotov.result.Result``
(reference src/krotov/result.py:18-78): ``tlist, objectives, iters,
iter_seconds, info_vals, tau_vals, guess_controls, optimized_controls,
controls_mapping, all_pulses, states, message, start_local_time,
end_local_time``.  Host bookkeeping only.

:meth:`Result.load` also reads dumps written by the reference itself
(``krotov.result.Result.dump`` under QuTiP 4): the pickled QuTiP objects are
rebuilt as NumPy arrays (state vectors 1-D, operators dense 2-D) by an
unpickler that maps the handful of reference / QuTiP classes such a dump
contains to stand-ins -- neither ``krotov`` nor ``qutip`` is imported -- so an
optimisation started with the reference can be continued here
(``continue_from=Result.load(path, objectives=...)``).
"""
import copy
import importlib
import logging
import pickle
import time

import numpy as np

__all__ = ['Result', 'ControlPlaceholder']

_FIELDS = (
    'objectives', 'tlist', 'iters', 'iter_seconds', 'info_vals', 'tau_vals',
    'guess_controls', 'optimized_controls', 'controls_mapping', 'all_pulses', 'states',
)


class Result:
    time_fmt = "%Y-%m-%d %H:%M:%S"

    def __init__(self):
        for name in _FIELDS:
            setattr(self, name, [])
        self.start_local_time = None
        self.end_local_time = None
        self.message = ''

    @staticmethod
    def _fmt(stamp, fmt):
        return 'n/a' if stamp is None else time.strftime(fmt, stamp)

    @property
    def start_local_time_str(self):
        return self._fmt(self.start_local_time, self.time_fmt)

    @property
    def end_local_time_str(self):
        return self._fmt(self.end_local_time, self.time_fmt)

    def __str__(self):
        return (
            "Krotov Optimization Result\n"
            "--------------------------\n"
            "- Started at %s\n- Number of objectives: %d\n- Number of iterations: %d\n"
            "- Reason for termination: %s\n- Ended at %s"
            % (
                self.start_local_time_str,
                len(self.objectives),
                max(len(self.iters) - 1, 0),
                self.message,
                self.end_local_time_str,
            )
        )

    __repr__ = __str__

    @property
    def optimized_objectives(self):
        """Copies of the objectives with every control replaced by its
        optimised array (reference result.py:127-130)."""
        return self.objectives_with_controls(self.optimized_controls)

    def objectives_with_controls(self, controls):
        """Copies of :attr:`objectives` with the given ``controls`` (one per
        entry of :attr:`guess_controls`, on the points of :attr:`tlist`)
        plugged into the nested lists (reference result.py:132-188).  Raises
        ``ValueError`` for a wrong number of controls or arrays that do not
        match the time grid."""
        if len(controls) != len(self.guess_controls):
            raise ValueError("Expected %d controls, %d given" % (len(self.guess_controls), len(controls)))
        for control in controls:
            try:
                if len(control) != len(self.tlist):
                    raise ValueError(
                        "controls are not defined on the points of the time grid: control has %d values "
                        "for %d time grid points" % (len(control), len(self.tlist))
                    )
            except TypeError:
                pass  # a callable
        out = []
        for i_obj, obj in enumerate(self.objectives):
            new = copy.copy(obj)  # nested lists copied, operators shared
            for i_control, control in enumerate(controls):
                for i in self.controls_mapping[i_obj][0][i_control]:
                    new.H[i][1] = control
                for i_c, _ in enumerate(new.c_ops):
                    for i in self.controls_mapping[i_obj][i_c + 1][i_control]:
                        new.c_ops[i_c][i][1] = control
            out.append(new)
        return out

    def dump(self, filename, reference=False):
        """Pickle the result.  Controls that are Python functions cannot be
        pickled: in the stored objectives they are replaced by
        :class:`ControlPlaceholder` s (pass the objectives again to
        :meth:`load`), as in the reference (result.py:247-262).

        ``reference=True`` writes the file in the reference's own format
        instead -- a pickled ``krotov.result.Result`` whose objectives are
        reduced through ``krotov.objectives._Objective_reduce_init`` and whose
        function controls are ``krotov.objectives._ControlPlaceholder`` s
        (result.py:247-262, objectives.py:581-636) -- with every state and
        operator as a NumPy array: the reference reads it with
        ``krotov.result.Result.load(filename, objectives=...)`` in its NumPy
        mode (``krotov.Objective.type_checking = False``, reference notebook
        09), e.g. to continue there an optimisation that was run here.  Neither
        ``krotov`` nor ``qutip`` is needed to write it; :meth:`load` reads it
        back as well.  (While the file is written, stand-in modules named
        ``krotov``, ``krotov.result``, ``krotov.objectives`` and
        ``numpy.core.multiarray`` sit in ``sys.modules``: do not import or
        unpickle from other threads at that moment.  Values of ``info_vals``
        that are instances of your own classes are pickled as they are.)"""
        if reference:
            return _dump_reference(self, filename)
        clone = copy.copy(self)
        clone.objectives = []
        for obj in self.objectives:
            new = copy.copy(obj)
            ids = {}
            for lst in [new.H] + list(new.c_ops):
                if isinstance(lst, list):
                    for term in lst:
                        if isinstance(term, list) and callable(term[1]):
                            term[1] = ControlPlaceholder(ids.setdefault(id(term[1]), len(ids)))
            clone.objectives.append(new)
        with open(filename, 'wb') as fh:
            pickle.dump(clone, fh)

    @classmethod
    def load(cls, filename, objectives=None, finalize=False):
        """Read a :meth:`dump`.  ``objectives`` replaces the stored ones (which
        have lost their function controls); with ``finalize=True`` optimized
        controls that were dumped mid-optimisation (``dump_result``), still on
        the intervals of the time grid, are mapped onto its points.  Warnings
        are logged like the reference's (result.py:190-244)."""
        from .conversions import pulse_onto_tlist

        logger = logging.getLogger('krotov')
        with open(filename, 'rb') as fh:
            res = _Unpickler(fh).load()
        if not isinstance(res, cls):
            raise pickle.UnpicklingError("%s does not contain a Result" % filename)
        for name, value in list(res.__dict__.items()):
            setattr(res, name, _without_qobj(value))
        if objectives is None:
            if any(_has_placeholder(obj.H) or any(_has_placeholder(c) for c in obj.c_ops) for obj in res.objectives):
                logger.warning(
                    "Result.objectives contains control placeholders. You should overwrite it by passing "
                    "`objectives`."
                )
        else:
            res.objectives = objectives
        nt = len(res.tlist)
        for i, control in enumerate(res.optimized_controls):
            if len(control) == nt:
                continue
            if len(control) == nt - 1:
                if finalize:
                    res.optimized_controls[i] = pulse_onto_tlist(control)
                    continue
                logger.warning("Result.optimized_controls are not finalized. Consider loading with `finalize=True`.")
            else:
                logger.error("Result.optimized_controls are incongruent with Result.tlist")
            break  # one message is enough
        return res


class ControlPlaceholder:
    """Stands in for a control function in a dumped objective."""

    def __init__(self, index):
        self.id = index

    def __eq__(self, other):
        return isinstance(other, ControlPlaceholder) and other.id == self.id

    def __hash__(self):
        return hash(('ControlPlaceholder', self.id))

    def __repr__(self):
        return "ControlPlaceholder(%d)" % self.id


def _has_placeholder(lst):
    if isinstance(lst, list):
        return any(_has_placeholder(v) for v in lst)
    return isinstance(lst, ControlPlaceholder)


# ---------------------------------------------------------------------------
# dumps written by the reference (krotov.result.Result.dump, result.py:247-262, under QuTiP 4)
# ---------------------------------------------------------------------------
class _RefState:
    """Receives the pickled attribute dict of a qutip.Qobj / qutip.fastsparse.fast_csr_matrix."""

    def __setstate__(self, state):
        self.__dict__.update(state)


class _RefQobj(_RefState):
    def to_array(self):
        m = self._data
        rows, cols = m._shape
        dense = np.zeros((rows, cols), dtype=np.complex128)
        indptr, indices, data = np.asarray(m.indptr), np.asarray(m.indices), np.asarray(m.data)
        for r in range(rows):
            lo, hi = indptr[r], indptr[r + 1]
            dense[r, indices[lo:hi]] += data[lo:hi]
        kind = getattr(self, '_type', None)
        if kind == 'ket' or (kind is None and cols == 1):
            return dense[:, 0].copy()
        if kind == 'bra':
            return dense[0, :].copy()
        return dense


class _RefCsr(_RefState):
    pass


def _ref_objective(initial_state, H, target, c_ops):
    """krotov.objectives._Objective_reduce_init (reference objectives.py:581-585)"""
    from .objectives import Objective

    obj = Objective.__new__(Objective)  # (the components are still stand-ins: no validation here)
    obj.initial_state, obj.H, obj.target, obj.c_ops = initial_state, H, target, ([] if c_ops is None else c_ops)
    return obj


class _Unpickler(pickle.Unpickler):
    """The classes of a reference dump are mapped to their stand-ins above (whether or not ``krotov`` / ``qutip``
    happen to be installed); everything else resolves as in :func:`pickle.load`."""

    _REFERENCE = {
        ('krotov.result', 'Result'): lambda: Result,
        ('krotov.objectives', 'Objective'): lambda: importlib.import_module('krotov_amd.objectives').Objective,
        ('krotov.objectives', '_Objective_reduce_init'): lambda: _ref_objective,
        ('krotov.objectives', '_ControlPlaceholder'): lambda: ControlPlaceholder,
        ('qutip.qobj', 'Qobj'): lambda: _RefQobj,
        ('qutip.fastsparse', 'fast_csr_matrix'): lambda: _RefCsr,
    }
    _PLAIN = {
        ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
        ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
        ('numpy', 'ndarray'), ('numpy', 'dtype'), ('time', 'struct_time'), ('builtins', 'complex'),
        ('builtins', 'set'), ('builtins', 'frozenset'), ('collections', 'OrderedDict'),
    }

    def find_class(self, module, name):
        if (module, name) in self._REFERENCE:
            return self._REFERENCE[(module, name)]()
        if (module, name) in self._PLAIN or module == 'krotov_amd' or module.startswith('krotov_amd.'):
            if module.startswith('numpy.core.'):  # NumPy 2 moved numpy.core to numpy._core
                try:
                    return getattr(importlib.import_module(module), name)
                except (ImportError, AttributeError):
                    module = module.replace('numpy.core.', 'numpy._core.', 1)
            return getattr(importlib.import_module(module), name)
        return super().find_class(module, name)  # user classes (custom attributes of objectives, info_vals)


def _without_qobj(value):
    """``value`` with every rebuilt QuTiP object replaced by its array (lists, tuples, dicts and objectives
    are walked)."""
    from .objectives import Objective

    if isinstance(value, _RefQobj):
        return value.to_array()
    if isinstance(value, list):
        return [_without_qobj(v) for v in value]
    if isinstance(value, tuple):
        return tuple(_without_qobj(v) for v in value)
    if isinstance(value, dict):
        return {k: _without_qobj(v) for k, v in value.items()}
    if isinstance(value, Objective):
        for k, v in list(value.__dict__.items()):
            setattr(value, k, _without_qobj(v))
    return value


# ---------------------------------------------------------------------------
# dumps the reference can load (Result.dump(..., reference=True))
# ---------------------------------------------------------------------------
def _plain(value):
    """``value`` in terms of what a process without this package unpickles: NumPy arrays, Python scalars, lists,
    tuples, dicts (states / operators -> arrays; lazy state lists -> lists)."""
    from ._ingest import to_dense

    if isinstance(value, np.generic):
        # (first: np.float64 / np.complex128 ARE Python floats / complex numbers by inheritance, and would otherwise
        # travel as NumPy scalars -- pickled under numpy._core.multiarray.scalar, a name NumPy 1.x cannot resolve)
        return value.item()
    if value is None or isinstance(value, (bool, int, float, complex, str, bytes, time.struct_time)):
        return value
    if isinstance(value, np.ndarray):
        return value if value.dtype != object else [_plain(v) for v in value]
    if isinstance(value, dict):
        return {k: _plain(v) for k, v in value.items()}
    if isinstance(value, tuple):
        return tuple(_plain(v) for v in value)
    if isinstance(value, list) or (hasattr(value, '__len__') and hasattr(value, '__getitem__') and not hasattr(value, 'shape')
                                   and not hasattr(value, 'full')):
        return [_plain(value[i]) for i in range(len(value))]
    if hasattr(value, 'full') or hasattr(value, 'shape') or hasattr(value, 'toarray'):
        return np.asarray(to_dense(value))
    return value  # (a user's own picklable object, e.g. an info_hook's return value)


def _dump_reference(result, filename):
    import sys
    import types

    # Stand-ins that pickle BY REFERENCE under the reference's names: the file then contains the global names
    # krotov.result.Result, krotov.objectives._Objective_reduce_init and krotov.objectives._ControlPlaceholder, which a
    # process that has the reference installed resolves to the real things.
    mod_result, mod_obj, mod_top = (types.ModuleType(n) for n in ('krotov.result', 'krotov.objectives', 'krotov'))

    class RefResult:
        pass

    class RefPlaceholder:
        def __init__(self, id):
            self.id = id

    def ref_reduce_init(initial_state, H, target, c_ops):  # (never called here: only its name is written)
        raise RuntimeError("stand-in for krotov.objectives._Objective_reduce_init")

    RefResult.__module__, RefResult.__qualname__, RefResult.__name__ = 'krotov.result', 'Result', 'Result'
    RefPlaceholder.__module__, RefPlaceholder.__qualname__ = 'krotov.objectives', '_ControlPlaceholder'
    RefPlaceholder.__name__ = '_ControlPlaceholder'
    ref_reduce_init.__module__, ref_reduce_init.__qualname__ = 'krotov.objectives', '_Objective_reduce_init'
    ref_reduce_init.__name__ = '_Objective_reduce_init'
    mod_result.Result = RefResult
    mod_obj._ControlPlaceholder, mod_obj._Objective_reduce_init = RefPlaceholder, ref_reduce_init
    mod_top.result, mod_top.objectives = mod_result, mod_obj

    class ObjectiveOut:
        """Pickles like the reference's ``_Objective_reduce`` (objectives.py:588-611)."""

        def __init__(self, obj):
            def nested(lst):
                if isinstance(lst, list):
                    return [nested(v) for v in lst]
                if isinstance(lst, ControlPlaceholder):
                    return RefPlaceholder(lst.id)
                if callable(lst) and not hasattr(lst, 'shape') and not hasattr(lst, 'full'):
                    return RefPlaceholder(id(lst))  # (the reference uses id() as well)
                return _plain(lst)

            self.args = (_plain(obj.initial_state), nested(obj.H), _plain(obj.target), nested(list(obj.c_ops)))
            self.extra = {k: _plain(v) for k, v in obj.__dict__.items() if k not in obj._default_attribs}

        def __reduce__(self):
            return (ref_reduce_init, self.args, self.extra)

    out = RefResult()
    for name in _FIELDS:
        value = getattr(result, name)
        out.__dict__[name] = [ObjectiveOut(o) for o in value] if name == 'objectives' else _plain(value)
    out.tlist = np.asarray(result.tlist, dtype=np.float64)
    out.guess_controls = [np.asarray(c) for c in result.guess_controls]
    out.optimized_controls = [np.asarray(c) for c in result.optimized_controls]
    out.start_local_time, out.end_local_time, out.message = result.start_local_time, result.end_local_time, result.message
    # Arrays are written with the reconstruction function under its NumPy-1 name, numpy.core.multiarray._reconstruct:
    # the reference's environment (QuTiP 4) has NumPy 1.x, which knows no numpy._core; NumPy 2 still resolves the old name.
    mod_np = types.ModuleType('numpy.core.multiarray')

    def np_reconstruct(*args):  # (never called here: only its name is written)
        raise RuntimeError("stand-in for numpy.core.multiarray._reconstruct")

    np_reconstruct.__module__, np_reconstruct.__qualname__ = 'numpy.core.multiarray', '_reconstruct'
    np_reconstruct.__name__ = '_reconstruct'
    mod_np._reconstruct = np_reconstruct

    class RefPickler(pickle.Pickler):
        def reducer_override(self, obj):
            if type(obj) is np.ndarray:
                red = obj.__reduce__()
                return (np_reconstruct,) + tuple(red[1:])
            return NotImplemented

    names = ('krotov', 'krotov.result', 'krotov.objectives', 'numpy.core.multiarray')
    saved = {n: sys.modules.get(n) for n in names}
    try:
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter('ignore')  # (NumPy 2 warns when its numpy.core shim is imported)
            importlib.import_module('numpy.core')
            sys.modules.update({'krotov': mod_top, 'krotov.result': mod_result, 'krotov.objectives': mod_obj,
                                'numpy.core.multiarray': mod_np})
            with open(filename, 'wb') as fh:
                RefPickler(fh, protocol=2).dump(out)
    finally:
        for name, module in saved.items():
            if module is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = module
