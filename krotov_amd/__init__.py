"""krotov_amd -- MI355X-native Krotov optimal-control engine.

Keeps the plugin surface of qucontrol/krotov (``optimize_pulses`` /
``Objective`` / ``propagator=``) and replaces the per-iteration
backward/forward propagation and pulse-update loop by hand-written CDNA4 HIP
kernels behind a C ABI (include/krotov_hip.h).  See DESIGN.md.
"""
from . import (
    configs,
    convergence,
    conversions,
    functionals,
    info_hooks,
    mu,
    objectives,
    parallelization,
    propagators,
    result,
    second_order,
    shapes,
)
from .objectives import Objective, ensemble_objectives, gate_objectives
from .optimize import optimize_pulses
from .result import Result

__version__ = '0.1.0'

__all__ = [
    'Objective',
    'Result',
    'convergence',
    'conversions',
    'ensemble_objectives',
    'functionals',
    'gate_objectives',
    'info_hooks',
    'mu',
    'objectives',
    'optimize_pulses',
    'parallelization',
    'propagators',
    'result',
    'second_order',
    'shapes',
]
