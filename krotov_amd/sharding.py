"""Objective sharding over the GPUs of one node (one process per GPU).

The only inter-objective dependence inside a Krotov iteration is the sum over
objectives in the pulse update (reference src/krotov/optimize.py:470; the
reference's own parallelization notes, parallelization.py:4-21).  Objectives are
therefore split into contiguous blocks, one per rank; every rank keeps its
operators, co-state store and running states private, and per time interval the
``L`` partial update sums are all-reduced (``torch.distributed`` on the
``nccl`` = RCCL backend over xGMI on GPUs, ``gloo`` in the CPU tests).  Once
per iteration the final states / tau_k are all-gathered for the chi
constructor.  The message is 8*L bytes, so the step is latency-bound, not
bandwidth-bound.

Nothing here touches device code: the functions are written against a small
"stepper" protocol so that the same loop drives the HIP engine
(:meth:`HipKrotovEngine.forward_update_sharded`) and, in the CPU tests, an
oracle-backed stand-in.
"""
import numpy as np

__all__ = ['shard_range', 'gather_rows', 'run_update_loop']


def shard_range(K, world, rank):
    """Contiguous block ``[k0, k1)`` of the ``K`` objectives owned by ``rank``.

    Balanced: the first ``K % world`` ranks own one objective more, so no rank is empty whenever
    ``K >= world`` (K = 5 over 4 ranks is 2 + 1 + 1 + 1, not 2 + 2 + 1 + 0)."""
    base, extra = divmod(K, world)
    k0 = rank * base + min(rank, extra)
    return k0, k0 + base + (1 if rank < extra else 0)


def gather_rows(local, K, world, group, device):
    """All-gather the per-rank row blocks of a (K, ...) host array.

    ``local`` holds this rank's rows (``shard_range`` order); returns the full
    (K, ...) array on every rank.  Blocks are padded to equal length for the
    collective.
    """
    if world == 1:
        return local
    import torch
    import torch.distributed as dist

    per = (K + world - 1) // world
    pad = np.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype)
    pad[: local.shape[0]] = local
    send = torch.from_numpy(np.ascontiguousarray(pad)).to(device)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    sizes = [shard_range(K, world, r) for r in range(world)]
    return np.concatenate([r.cpu().numpy()[: k1 - k0] for r, (k0, k1) in zip(recv, sizes)], axis=0)


def run_update_loop(stepper, n_intervals, all_reduce):
    """Forward sweep with sequential update, cut at the cross-objective sum.

    ``stepper`` provides
      * ``begin() -> partial``      tensor of L local partial sums of interval 0
      * ``step(n, D) -> partial``   apply the all-reduced sums ``D`` of interval
        ``n`` (update eps[n], propagate the local states over interval n) and
        return the local partial sums of interval ``n+1`` (may return the same
        tensor object; contents are ignored after the last interval)
      * ``end()``                   finish (e.g. copy out the final states)
    ``all_reduce(t)`` sums the tensor ``t`` over all ranks in place.  Every rank
    executes exactly the same sequence, so the collectives match up.
    """
    partial = stepper.begin()
    for n in range(n_intervals):
        all_reduce(partial)
        partial = stepper.step(n, partial)
    return stepper.end()
