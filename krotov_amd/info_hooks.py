"""Minimal ``info_hook`` helpers.

Only :func:`chain` (reference src/krotov/info_hooks.py:24-56) is provided; the
reference's table printers are formatting-only and outside the accelerated
path.  An ``info_hook`` receives the keyword arguments listed in reference
info_hooks.py:59-86.
"""

__all__ = ['chain']


def chain(*hooks):
    """Call ``hooks`` in order with the same keyword arguments; the results
    that are not None are returned as a tuple (None if there are none, the bare
    value if there is exactly one)."""

    def info_hook(**kwargs):
        results = []
        for hook in hooks:
            res = hook(**kwargs)
            if res is not None:
                results.append(res)
        if len(results) == 0:
            return None
        if len(results) == 1:
            return results[0]
        return tuple(results)

    return info_hook
