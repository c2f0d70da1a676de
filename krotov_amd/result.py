"""Record of an optimisation, field-compatible with ``krotov.result.Result``
(reference src/krotov/result.py:18-78): ``tlist, objectives, iters,
iter_seconds, info_vals, tau_vals, guess_controls, optimized_controls,
controls_mapping, all_pulses, states, message, start_local_time,
end_local_time``.  Host bookkeeping only.
"""
import copy
import pickle
import time

__all__ = ['Result']

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
        optimised array (reference result.py:124-152)."""
        out = []
        for i_obj, obj in enumerate(self.objectives):
            new = copy.copy(obj)
            for i_control, control in enumerate(self.optimized_controls):
                for i in self.controls_mapping[i_obj][0][i_control]:
                    new.H[i][1] = control
                for i_c, _ in enumerate(new.c_ops):
                    for i in self.controls_mapping[i_obj][i_c + 1][i_control]:
                        new.c_ops[i_c][i][1] = control
            out.append(new)
        return out

    def dump(self, filename):
        """Pickle the numeric record (controls that are functions are dropped
        from the stored objectives' nested lists by replacing them with their
        discretised guess arrays)."""
        clone = copy.copy(self)
        clone.objectives = []
        for i_obj, obj in enumerate(self.objectives):
            new = copy.copy(obj)
            for i_control, control in enumerate(self.guess_controls):
                for i in self.controls_mapping[i_obj][0][i_control]:
                    new.H[i][1] = control
            clone.objectives.append(new)
        with open(filename, 'wb') as fh:
            pickle.dump(clone, fh)

    @classmethod
    def load(cls, filename, objectives=None):
        with open(filename, 'rb') as fh:
            res = pickle.load(fh)
        if objectives is not None:
            res.objectives = objectives
        return res
