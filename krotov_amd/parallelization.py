"""Objective-level dispatch ("parallel_map") plug-in point.

The reference parallelises over objectives with process pools
(src/krotov/parallelization.py:233-604).  Here all objectives of a GPU are
batched inside one kernel launch, so there is nothing left to distribute on the
device path: every map of this module is the serial one (the signature of
``qutip.parallel.serial_map`` that the reference defaults to,
optimize.py:266-269).  The reference's names exist so that scripts which pass

    parallel_map=(krotov.parallelization.parallel_map,
                  krotov.parallelization.parallel_map,
                  krotov.parallelization.parallel_map_fw_prop_step)

or call :func:`set_parallelization` keep running unchanged: with the GPU
propagator the maps are not used at all, with a plugin propagator the generic
loop of :func:`krotov_amd.optimize_pulses` calls them (serially).  The worker
classes ``Consumer`` / ``FwPropStepTask`` of the reference are not provided.
"""

__all__ = ['set_parallelization', 'parallel_map', 'serial_map', 'parallel_map_fw_prop_step']

USE_LOKY = False
"""Kept for scripts that set it (reference parallelization.py:102-128); no effect."""

USE_THREADPOOL_LIMITS = True
"""Kept for scripts that set it (reference parallelization.py:130-158); no effect."""


def set_parallelization(use_loky=False, start_method=None, loky_pickler=None, use_threadpool_limits=True):
    """Accepts the reference's arguments (parallelization.py:172-230) and records the two flags; there are
    no worker processes to configure.  ``start_method`` is validated as in the reference."""
    global USE_LOKY, USE_THREADPOOL_LIMITS
    allowed = ['fork', 'spawn', 'forkserver'] + (['loky', 'loky_int_main'] if use_loky else [])
    if start_method is not None and start_method not in allowed:
        raise ValueError("start_method not in %s" % str(allowed))
    USE_LOKY = bool(use_loky)
    USE_THREADPOOL_LIMITS = bool(use_threadpool_limits)


def serial_map(task, values, task_args=(), task_kwargs=None, **kwargs):
    """``[task(v, *task_args, **task_kwargs) for v in values]``."""
    task_kwargs = {} if task_kwargs is None else task_kwargs
    return [task(v, *task_args, **task_kwargs) for v in values]


def parallel_map(task, values, task_args=(), task_kwargs=None, num_cpus=None, progress_bar=None):
    """Signature of the reference's process-pool map (parallelization.py:233-299); evaluated serially."""
    return serial_map(task, values, task_args, task_kwargs)


def parallel_map_fw_prop_step(shared, values, task_args):
    """Signature of the reference's map for the per-interval forward step (parallelization.py:433-495), which
    :func:`optimize_pulses` calls as ``map(task, range(K), task_args)``; evaluated serially."""
    return [shared(v, *task_args) for v in values]
