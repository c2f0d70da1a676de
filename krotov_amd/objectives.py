"""Control objectives: the problem container the optimiser consumes.

API-compatible with ``krotov.objectives`` (reference src/krotov/objectives.py:
96-258 ``Objective``, 704-1051 ``gate_objectives``, 1054-1094
``ensemble_objectives``, 1097-1121 ``liouvillian``) for the parts on or next to
the hot path.  Operators and states may be QuTiP-like objects (anything with
``.full()`` / ``.dag()``) or NumPy arrays; QuTiP itself is never imported, so
``Objective.propagate`` (338-433) runs on the engine's forward sweep,
``summarize`` / ``str`` (445-578) number the components like the reference,
``mesolve`` delegates to QuTiP when it is installed.
"""
import copy
import itertools
from collections import defaultdict

import numpy as np

__all__ = ['Objective', 'PropagationResult', 'gate_objectives', 'ensemble_objectives', 'liouvillian']


def _shallow_nested(l):
    if isinstance(l, list):
        return [list(h) if isinstance(h, list) else h for h in l]
    return l


def _adjoint(op, ignore_errors=False):
    """Adjoint of an operator/state or of a nested-list operator; controls stay
    untouched (reference objectives.py:51-93)."""
    if isinstance(op, list):
        out = []
        for item in op:
            if isinstance(item, list):
                if len(item) != 2:
                    if ignore_errors:
                        return op
                    raise ValueError(
                        "%s is not the in the expected format of the "
                        "two-element list '[operator, control]'" % item
                    )
                out.append([_adjoint(item[0]), item[1]])
            else:
                out.append(_adjoint(item))
        return out
    if op is None or isinstance(op, str):
        return op
    if hasattr(op, 'dag'):
        return op.dag()
    if hasattr(op, 'conj') and hasattr(op, 'T'):
        return op.conj().T
    if hasattr(op, 'conjugate') and hasattr(op, 'transpose'):
        return op.conjugate().transpose()
    if ignore_errors:
        return op
    raise ValueError("Cannot calculate adjoint of %s" % op)


def _same(a, b):
    """Equality that also works for arrays and nested lists."""
    if a is b:
        return True
    if isinstance(a, list) or isinstance(b, list):
        if not (isinstance(a, list) and isinstance(b, list)) or len(a) != len(b):
            return False
        return all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        try:
            return np.shape(a) == np.shape(b) and bool(np.all(np.asarray(a) == np.asarray(b)))
        except Exception:
            return False
    try:
        return bool(a == b)
    except Exception:
        return False


class Objective:
    """One control objective: ``initial_state`` evolving under ``H`` (nested
    list ``[H0, [H1, control], ...]``, or Liouvillian in the same format)
    should reach ``target``; ``c_ops`` are Lindblad operators.

    Custom attributes (e.g. ``weight``) survive copies and ``adjoint()``.
    With ``type_checking`` (class attribute, default True as in the reference,
    objectives.py:154-183) the constructor rejects what cannot be a state or a
    (nested list of) operator(s): ``ValueError("Invalid initial_state ...")`` /
    ``"Invalid H ..."`` / ``"Invalid c_ops ..."``.  Without QuTiP the test is
    structural (array-likes and objects with ``.full()`` pass; None, callables,
    tuples, strings do not).
    """

    type_checking = True
    str_use_unicode = True
    _default_attribs = ['initial_state', 'H', 'target', 'c_ops']
    _counter = defaultdict(int, {'u{count}(t)': 1})
    _count_cache = {}

    def __init__(self, *, initial_state, H, target, c_ops=None):
        if self.type_checking:
            if not _operator_like(initial_state):
                raise ValueError("Invalid initial_state: must be a state (an array or an object with .full()), "
                                 "not %s" % type(initial_state).__name__)
            if not (_operator_like(H) or _nested_operator_list(H)):
                raise ValueError("Invalid H, must be an operator or a nested list [H0, [H1, control], ...], not %s"
                                 % type(H).__name__)
            if c_ops is not None and not (isinstance(c_ops, list) and all(
                    _operator_like(c) or _nested_operator_list(c)
                    or (isinstance(c, list) and len(c) == 2 and _operator_like(c[0]))  # [operator, control]
                    for c in c_ops)):
                raise ValueError("Invalid c_ops, must be a list of operators or nested lists")
        self.H = H
        self.initial_state = initial_state
        self.target = target
        self.c_ops = [] if c_ops is None else c_ops

    def _extras(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._default_attribs}

    def __copy__(self):
        new = Objective(
            H=_shallow_nested(self.H),
            initial_state=self.initial_state,
            target=self.target,
            c_ops=[_shallow_nested(c) for c in self.c_ops],
        )
        new.__dict__.update(self._extras())
        return new

    def __deepcopy__(self, memo):
        new = Objective(
            H=copy.deepcopy(self.H, memo),
            initial_state=copy.deepcopy(self.initial_state, memo),
            target=copy.deepcopy(self.target, memo),
            c_ops=[copy.deepcopy(c, memo) for c in self.c_ops],
        )
        for k, v in self._extras().items():
            setattr(new, k, copy.deepcopy(v, memo))
        return new

    def __eq__(self, other):
        if other.__class__ is not self.__class__:
            return NotImplemented
        if self.__dict__.keys() != other.__dict__.keys():
            return False
        return all(_same(getattr(self, k), getattr(other, k)) for k in self.__dict__)

    def __ne__(self, other):
        res = self.__eq__(other)
        return res if res is NotImplemented else not res

    __hash__ = None

    def adjoint(self):
        """Objective with the adjoint of every component; controls are assumed
        real and are kept; a non-state ``target`` is kept as is (reference
        objectives.py:240-258)."""
        adj = Objective(
            H=_adjoint(self.H),
            initial_state=_adjoint(self.initial_state),
            target=_adjoint(self.target, ignore_errors=True),
            c_ops=[_adjoint(op) for op in self.c_ops],
        )
        adj.__dict__.update(self._extras())
        return adj

    def propagate(self, tlist, *, propagator, rho0=None, H=None, c_ops=None, e_ops=None, args=None, expect=None):
        """Propagate ``rho0`` (default: ``initial_state``) over the whole time grid with ``propagator`` -- the
        same piecewise-constant-on-the-intervals convention, the same pulses (controls sampled by
        :func:`~krotov_amd.conversions.discretize` and :func:`~krotov_amd.conversions.control_onto_interval`)
        and the same propagator as :func:`~krotov_amd.optimize.optimize_pulses` uses (reference
        objectives.py:338-433).  With ``krotov_amd.propagators.expm`` / ``HipExpm`` and no ``c_ops`` the whole
        sweep is ONE launch of the engine's forward sweep with storage (include/krotov_hip.h:
        ``kh_forward_store``); any other callable is stepped on the host as in the reference.

        Returns a :class:`PropagationResult` (attributes of ``qutip.solver.Result``): ``states`` at every
        grid point, or -- with ``e_ops`` -- ``expect[i]``, the expectation values of ``e_ops[i]``
        (``expect(operator, state)``, default <psi|O|psi> for vectors, tr(O rho) for matrices).
        """
        from .conversions import (control_onto_interval, discretize, extract_controls, extract_controls_mapping,
                                  plug_in_pulse_values)

        H = self.H if H is None else H
        c_ops = self.c_ops if c_ops is None else c_ops
        e_ops = [] if e_ops is None else e_ops
        args = {} if args is None else args
        expect = _expectation_value if expect is None else expect
        tlist = np.asarray(tlist, dtype=np.float64)
        state = self.initial_state if rho0 is None else rho0
        system = Objective(initial_state=state, H=H, target=self.target, c_ops=c_ops)
        controls = extract_controls([system])
        mapping = extract_controls_mapping([system], controls)
        pulses = [control_onto_interval(discretize(control, tlist, args=(args,))) for control in controls]
        result = PropagationResult()
        result.solver = getattr(propagator, '__name__', propagator.__class__.__name__)
        result.times = np.array(tlist)
        result.num_expect, result.num_collapse = len(e_ops), len(c_ops)

        from .optimize import _HipBackend, _use_device_path  # (imports this module)

        if _use_device_path(propagator, None, None, None, 'array', [system]) and len(tlist) > 1:
            backend = _HipBackend([system], mapping, tlist, len(controls), propagator)
            _, trajectories = backend.initial_forward(pulses, store=True)
            states = list(trajectories[0])
            backend.engine.close()
        else:
            states = [state]
            for n in range(len(tlist) - 1):
                H_n = plug_in_pulse_values(H, pulses, mapping[0][0], n)
                c_ops_n = [plug_in_pulse_values(c, pulses, mapping[0][1 + i], n) for i, c in enumerate(c_ops)]
                state = propagator(H_n, state, tlist[n + 1] - tlist[n], c_ops_n, initialize=True)
                states.append(state)
        if len(e_ops) == 0:
            result.states = states
        else:
            result.expect = [np.array([expect(oper, st) for st in states]) for oper in e_ops]
        return result

    def mesolve(self, tlist, *, rho0=None, H=None, c_ops=None, e_ops=None, args=None, **kwargs):
        """The reference delegates this to :func:`qutip.mesolve` (objectives.py:260-336: controls piecewise
        constant CENTERED on the grid points, QuTiP's ODE solver).  Done here the same way when QuTiP is
        installed and the components are QuTiP objects; otherwise use :meth:`propagate`."""
        try:
            import qutip
        except ImportError as exc:
            raise NotImplementedError(
                "Objective.mesolve delegates to qutip.mesolve and QuTiP is not installed; "
                "Objective.propagate(tlist, propagator=krotov_amd.propagators.expm) runs on the GPU") from exc
        return qutip.mesolve(H=self.H if H is None else H, rho0=self.initial_state if rho0 is None else rho0,
                             tlist=tlist, c_ops=self.c_ops if c_ops is None else c_ops,
                             e_ops=[] if e_ops is None else e_ops, args={} if args is None else args, **kwargs)

    @classmethod
    def reset_symbol_counters(cls):
        """Restart the numbering :meth:`summarize` gives to the objects it meets (reference
        objectives.py:435-443)."""
        cls._counter = defaultdict(int, {'u{count}(t)': 1})
        cls._count_cache = {}

    def summarize(self, use_unicode=True, reset_symbol_counters=False):
        """One-line summary, e.g. ``a₀[2] to a₁[2] via [a₂[2,2], [a₃[2,2], u₁(t)]]`` (reference
        objectives.py:445-572): every distinct object gets a symbol by category -- ``a`` NumPy arrays, ``u``
        control functions, and for QuTiP-like objects (``.type``, ``.dims``, ``.isherm``) ``Ψ`` states, ``ρ``
        density matrices, ``H`` / ``A`` (non-)Hermitian operators, ``𝓛`` super-operators, ``L`` Lindblad
        operators -- numbered per process in order of first appearance, the same object always with the same
        number; ``use_unicode=False`` gives the ASCII spelling."""
        if reset_symbol_counters:
            self.reset_symbol_counters()
        cls = type(self)
        part = lambda obj, role: _summarize_component(obj, role, cls._counter, cls._count_cache, use_unicode)  # noqa: E731
        res = part(self.initial_state, 'state')
        if self.target is not None:
            same_kind = (hasattr(self.initial_state, 'dims') and hasattr(self.target, 'dims')
                         and self.target.dims == self.initial_state.dims)
            res += " to " + part(self.target, 'state' if same_kind else 'target')
        res += " via "
        if len(self.c_ops) == 0:
            res += part(self.H, 'op')
        else:
            res += '{H:' + part(self.H, 'op') + ', c_ops:(' + ",".join(part(c, 'lindblad') for c in self.c_ops) + ')}'
        return res

    def __str__(self):
        return self.summarize(use_unicode=self.str_use_unicode)

    def __repr__(self):
        return "%s[%s]" % (self.__class__.__name__, str(self))


def _pattern_of(obj, role, use_unicode):
    """Category pattern of a component (reference objectives.py:1124-1171), or None for unknown objects."""
    if callable(obj) and not hasattr(obj, 'dims'):
        return 'u{count}(t)' if role == 'op' else None
    if isinstance(obj, np.ndarray):
        return 'a{count}[{dims}]'
    kind = getattr(obj, 'type', None)
    if kind is None or not hasattr(obj, 'dims'):
        return None
    if kind == 'ket':
        return '|Ψ{count}({dims})⟩' if use_unicode else '|Psi{count}({dims})>'
    if kind == 'bra':
        return '⟨Ψ{count}({dims})|' if use_unicode else '<Psi{count}({dims})|'
    if kind == 'oper':
        if role == 'lindblad':
            return 'L{count}[{dims}]'
        if getattr(obj, 'isherm', False):
            if role == 'state':
                return 'ρ{count}[{dims}]' if use_unicode else 'rho{count}[{dims}]'
            return 'H{count}[{dims}]'
        return 'A{count}[{dims}]'
    if kind == 'super':
        return '𝓛{count}[{dims}]' if use_unicode else 'Lv{count}[{dims}]'
    raise NotImplementedError("Unknown qobj type: %s" % kind)


def _dims_of(obj, use_unicode):
    """Shape / tensor structure of a component as text (reference objectives.py:1174-1200)."""
    times = '⊗' if use_unicode else '*'
    kind = getattr(obj, 'type', None)
    if kind is not None and hasattr(obj, 'dims'):
        join = lambda dim: times.join("%d" % d for d in dim)  # noqa: E731
        if kind == 'ket':
            return join(obj.dims[0])
        if kind == 'bra':
            return join(obj.dims[1])
        if kind == 'oper':
            return ",".join(join(dim) for dim in obj.dims)
        if kind == 'super':
            return ",".join('[%s,%s]' % (join(dim[0]), join(dim[1])) for dim in obj.dims)
        raise NotImplementedError("Unknown qobj type: %s" % kind)
    if hasattr(obj, 'shape'):
        return ",".join(str(int(d)) for d in obj.shape)
    return None


def _summarize_component(obj, role, counter=None, count_cache=None, use_unicode=True):
    """Text of one component of an objective (reference objectives.py:1203-1309)."""
    from .result import ControlPlaceholder

    if role not in ('state', 'target', 'op', 'lindblad'):
        raise ValueError("Unknown %s not in %s" % (role, ['state', 'target', 'op', 'lindblad']))
    counter = Objective._counter if counter is None else counter
    count_cache = Objective._count_cache if count_cache is None else count_cache
    if isinstance(obj, list):
        return '[' + ", ".join(_summarize_component(o, role, counter, count_cache, use_unicode) for o in obj) + ']'
    if isinstance(obj, (ControlPlaceholder, float, complex)):
        return str(obj)
    pattern = _pattern_of(obj, role, use_unicode)
    if pattern is None:  # unknown object: its own text, on one line, truncated to 40 characters
        res = str(obj).replace("\n", " ")
        if len(res) > 40:
            res = res[:39] + "…" if use_unicode else res[:37] + "..."
        return res
    key = id(obj)  # the same OBJECT keeps its number; equal copies get new ones
    if key in count_cache:
        count = count_cache[key]
    else:
        count = counter[pattern]
        count_cache[key] = count
        counter[pattern] += 1
        if pattern == 'A{count}[{dims}]':  # Hermitian and non-Hermitian operators share one numbering
            counter['H{count}[{dims}]'] += 1
        elif pattern == 'H{count}[{dims}]':
            counter['A{count}[{dims}]'] += 1
    count_str = str(count)
    if use_unicode:
        count_str = "".join(chr(ord(d) - ord('0') + 0x2080) for d in count_str)
    return pattern.format(count=count_str, dims=_dims_of(obj, use_unicode))


def _operator_like(x):
    if x is None or callable(x) and not hasattr(x, 'full') or isinstance(x, (str, bytes, tuple, list, dict)):
        return False
    return hasattr(x, 'full') or hasattr(x, 'shape') or isinstance(x, (int, float, complex))


def _nested_operator_list(lst):
    if not isinstance(lst, list) or len(lst) == 0:
        return False
    return all(_operator_like(t) or (isinstance(t, list) and len(t) == 2 and _operator_like(t[0])) for t in lst)


def _remove_functions_from_nested_list(lst):
    """Copy of a nested list with every callable control replaced by a placeholder numbered in order of first
    appearance (reference objectives.py:629-636)."""
    from .result import ControlPlaceholder

    ids = {}

    def walk(v):
        if isinstance(v, list):
            return [walk(x) for x in v]
        if callable(v) and not hasattr(v, 'full'):
            return ControlPlaceholder(ids.setdefault(id(v), len(ids)))
        return v

    return walk(lst)


def _Objective_reduce_init(initial_state, H, target, c_ops):
    return Objective(initial_state=initial_state, H=H, target=target, c_ops=c_ops)


def _Objective_reduce(obj):
    """Reduction function for pickling an :class:`Objective` whose controls are functions, for a pickler's
    ``dispatch_table`` / :func:`copyreg.pickle` (reference objectives.py:588-610): the functions -- which the
    standard pickle cannot store -- become :class:`~krotov_amd.result.ControlPlaceholder` s."""
    extras = {k: v for k, v in obj.__dict__.items() if k not in obj._default_attribs}
    return (
        _Objective_reduce_init,
        (obj.initial_state, _remove_functions_from_nested_list(obj.H), obj.target,
         _remove_functions_from_nested_list(obj.c_ops)),
        extras,
    )


class PropagationResult:
    """What :meth:`Objective.propagate` returns: the attributes of ``qutip.solver.Result`` that the reference
    fills (objectives.py:384-433)."""

    def __init__(self):
        self.solver = 'n/a'
        self.times = np.array([])
        self.states = []
        self.expect = []
        self.num_expect = 0
        self.num_collapse = 0


def _expectation_value(oper, state):
    """<psi|O|psi> for a state vector, tr(O rho) for a density matrix; real for Hermitian ``oper``."""
    O = np.asarray(oper.full() if hasattr(oper, 'full') else oper, dtype=complex)
    psi = np.asarray(state.full() if hasattr(state, 'full') else state, dtype=complex)
    if psi.ndim == 2 and psi.shape[0] == psi.shape[1] and psi.shape[0] > 1:
        val = np.trace(O @ psi)
    else:
        vec = psi.reshape(-1)
        val = np.vdot(vec, O @ vec)
    return val.real if np.array_equal(O, O.conj().T) else val


# ---------------------------------------------------------------------------
# constructors
# ---------------------------------------------------------------------------


def _ketbra(a, b):
    """|a><b| for Qobj-like or array kets."""
    if hasattr(a, 'dag') and hasattr(b, 'dag'):
        return a * b.dag()
    return np.outer(np.asarray(a).reshape(-1), np.conj(np.asarray(b).reshape(-1)))


def _rho1(basis):
    d = len(basis)
    return sum((2 * (d - i) / (d * (d + 1))) * _ketbra(p, p) for i, p in enumerate(basis))


def _rho2(basis):
    d = len(basis)
    return (1.0 / d) * sum(_ketbra(pi, pj) for pi, pj in itertools.product(basis, repeat=2))


def _rho3(basis):
    d = len(basis)
    return (1.0 / d) * sum(_ketbra(p, p) for p in basis)


def _li_pe_objectives(basis_states, gate, H, c_ops):
    """Bell-basis objectives for local-invariants / perfect-entangler
    optimisation (reference objectives.py:1035-1051)."""
    if len(basis_states) != 4:
        raise ValueError("Optimization towards a two-qubit gate requires 4 basis_states")
    b = basis_states
    psis = [
        (b[0] + b[3]) / np.sqrt(2),
        (1j * b[1] + 1j * b[2]) / np.sqrt(2),
        (b[1] - b[2]) / np.sqrt(2),
        (1j * b[0] - 1j * b[3]) / np.sqrt(2),
    ]
    return [Objective(initial_state=psi, target=gate, H=H, c_ops=c_ops) for psi in psis]


def gate_objectives(
    basis_states,
    gate,
    H,
    *,
    c_ops=None,
    local_invariants=False,
    liouville_states_set=None,
    weights=None,
    normalize_weights=True,
):
    """Objectives for optimising towards a quantum gate (reference
    objectives.py:704-1032).

    ``gate`` is an ``n x n`` matrix-like (``gate[i, j]``, ``.shape``), or the
    string ``'PE'`` / ``'perfect_entangler'``.  ``liouville_states_set`` in
    {None, 'full', '3states', 'd+1'} selects density-matrix objectives.
    """
    if isinstance(gate, str):
        if gate.lower().replace(' ', '_') in ('pe', 'perfect_entangler'):
            return _li_pe_objectives(basis_states, 'PE', H, c_ops)
        raise ValueError(
            "gate must be either a square matrix, or one of the strings "
            "'PE' or 'perfect_entangler', not '" + gate + "'"
        )
    if local_invariants:
        if tuple(gate.shape) != (4, 4):
            raise ValueError(
                "If local_invariants is True, gate must be a 4 × 4 matrix, not " + str(gate.shape)
            )
        return _li_pe_objectives(basis_states, gate, H, c_ops)
    n = len(basis_states)
    if not (gate.shape[0] == gate.shape[1] == n):
        raise ValueError("gate must be a matrix of the same dimension as the number of basis states")
    mapped = [sum(complex(gate[i, j]) * basis_states[i] for i in range(n)) for j in range(n)]
    for i, state in enumerate(mapped):  # reuse identical basis objects (permutation gates)
        for basis_state in basis_states:
            if _same(state, basis_state):
                mapped[i] = basis_state
    if liouville_states_set is None:
        initial, target = list(basis_states), mapped
    else:
        key = liouville_states_set.replace(' ', '').lower()
        if key == 'full':
            initial = [_ketbra(a, b) for a, b in itertools.product(basis_states, repeat=2)]
            target = [_ketbra(a, b) for a, b in itertools.product(mapped, repeat=2)]
        elif key == '3states':
            initial = [_rho1(basis_states), _rho2(basis_states), _rho3(basis_states)]
            target = [_rho1(mapped), _rho2(mapped), _rho3(mapped)]
        elif key == 'd+1':
            initial = [_ketbra(p, p) for p in basis_states] + [_rho2(basis_states)]
            target = [_ketbra(p, p) for p in mapped] + [_rho2(mapped)]
        else:
            raise ValueError("Invalid `liouville_states_set`: %s" % liouville_states_set)
    objectives = [
        Objective(initial_state=psi, target=tgt, H=H, c_ops=c_ops) for psi, tgt in zip(initial, target)
    ]
    if weights is not None:
        if len(weights) != len(objectives):
            raise ValueError("If weight are given, there must be a weight for each objective")
        if normalize_weights:
            weights = len(objectives) * np.array(weights) / np.sum(weights)
        for i in reversed(range(len(objectives))):
            w = float(weights[i])
            if w < 0:
                raise ValueError("weights must be greater than zero")
            objectives[i].weight = w
            if w == 0:
                del objectives[i]
    return objectives


def ensemble_objectives(objectives, Hs, *, keep_original_objectives=True):
    """One copy of every objective per Hamiltonian in ``Hs`` (robustness
    ensemble; reference objectives.py:1054-1094)."""
    out = list(objectives) if keep_original_objectives else []
    for H in Hs:
        for obj in objectives:
            out.append(Objective(H=H, initial_state=obj.initial_state, target=obj.target, c_ops=obj.c_ops))
    return out


def _dense_liouvillian(H, c_ops):
    """Column-stacking Liouvillian (cf. configs.liouvillian_dense): a dense array for array operators; for a
    QuTiP-like ``H`` (``.full()`` and ``.dims``) an object of the same class built through its constructor with
    the super-operator dims ``[[rows, rows], [cols, cols]]`` of ``qutip.liouvillian`` -- so that it carries
    ``.type == 'super'``, which the propagators and ``mu`` dispatch on (reference propagators.py:96-114, mu.py:130)."""
    from .configs import liouvillian_dense
    from ._ingest import to_dense

    arr = liouvillian_dense(to_dense(H), [to_dense(c) for c in c_ops])
    dims = getattr(H, 'dims', None)
    if hasattr(H, 'full') and dims is not None:
        return H.__class__(arr, dims=[[list(dims[0]), list(dims[0])], [list(dims[1]), list(dims[1])]])
    return arr


def liouvillian(H, c_ops):
    """Liouvillian of a (possibly nested-list) Hamiltonian and constant
    Lindblad operators for column-stacked vec(rho): dense arrays for array
    operators, super-operators of the operators' own class for QuTiP-like ones
    (reference objectives.py:1097-1121 delegates to qutip.liouvillian)."""
    c_ops = list(c_ops or [])
    if not isinstance(H, list):
        return _dense_liouvillian(H, c_ops)
    out = []
    for spec in H:
        if isinstance(spec, list):
            out.append([_dense_liouvillian(spec[0], []), spec[1]])
        else:
            out.append(_dense_liouvillian(spec, c_ops))
            c_ops = []
    assert len(c_ops) == 0, "No drift Hamiltonian"
    return out
