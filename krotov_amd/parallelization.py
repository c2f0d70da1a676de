"""Objective-level dispatch ("parallel_map") plug-in point.

The reference parallelises over objectives with process pools
(src/krotov/parallelization.py:233-604).  Here all objectives of a GPU are
batched inside one kernel launch, so the only map this package ships is the
serial one (the signature of ``qutip.parallel.serial_map`` that the reference
defaults to, optimize.py:266-269).  A user-supplied ``parallel_map`` is honoured
by the generic (plugin) loop of :func:`krotov_amd.optimize_pulses`.
"""

__all__ = ['serial_map']


def serial_map(task, values, task_args=(), task_kwargs=None, **kwargs):
    """``[task(v, *task_args, **task_kwargs) for v in values]``."""
    task_kwargs = {} if task_kwargs is None else task_kwargs
    return [task(v, *task_args, **task_kwargs) for v in values]
