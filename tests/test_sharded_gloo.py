"""The N > 1 path on CPU: world_size-2 ``gloo`` process groups drive the product's
sharding code (krotov_amd.sharding, the process_group branch of optimize_pulses)
with an oracle-backed engine stand-in, and must reproduce the single-process
oracle -- identically on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import krotov_amd
from krotov_amd import configs, sharding

from helpers import oracle_optimize

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, K, queue):
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import krotov_amd.engine as engine_mod
        from oracle_engine_double import OracleEngineDouble

        engine_mod.HipKrotovEngine = OracleEngineDouble  # the GPU engine's CPU stand-in
        spec = configs.config_c5(K=K, N=6, nt=41, L=2, distinct=True)
        spec.chi = 'sm'  # needs every rank's tau: exercises the all-gather
        objectives, pulse_options = configs.spec_to_objectives(spec, krotov_amd)
        res = krotov_amd.optimize_pulses(
            objectives, pulse_options, spec.tlist, propagator=krotov_amd.propagators.expm,
            chi_constructor=krotov_amd.functionals.chis_sm, iter_stop=2, store_all_pulses=True,
            process_group=dist.group.WORLD)
        queue.put((rank, np.array(res.all_pulses), np.array(res.tau_vals),
                   np.array([np.asarray(s).ravel() for s in res.states])))
        # gather_rows with an uneven split
        k0, k1 = sharding.shard_range(5, world, rank)
        local = np.arange(5 * 3, dtype=np.float64).reshape(5, 3)[k0:k1]
        full = sharding.gather_rows(local, 5, world, dist.group.WORLD, torch.device('cpu'))
        assert np.array_equal(full, np.arange(15, dtype=np.float64).reshape(5, 3))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('K,world', [(4, 2), (5, 2), (7, 4)])
def test_two_rank_gloo_matches_single_process_oracle(K, world):
    """world 2 (even and uneven shards) and world 4 (7 objectives as 2 + 2 + 2 + 1)."""
    ctx = mp.get_context('spawn')
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, queue)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([queue.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    spec = configs.config_c5(K=K, N=6, nt=41, L=2, distinct=True)
    spec.chi = 'sm'
    ref = oracle_optimize(spec, 2)
    for rank, pulses, tau, states in out:
        assert np.abs(pulses - ref['all_pulses']).max() < 1e-12
        assert np.abs(tau - ref['tau_vals']).max() < 1e-12
        assert np.abs(states - ref['fw_T']).max() < 1e-12
    # every rank derived bit-identical pulses (same all-reduced sums, same arithmetic)
    assert all(np.array_equal(o[1], out[0][1]) and np.array_equal(o[2], out[0][2]) for o in out)


def test_shard_range_covers_everything():
    for K in (1, 2, 7, 256, 257):
        for world in (1, 2, 3, 8):
            blocks = [sharding.shard_range(K, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == K
            for (a0, a1), (b0, b1) in zip(blocks, blocks[1:]):
                assert a1 == b0 and a0 <= a1


def test_update_loop_protocol_order():
    calls = []

    class S:
        def begin(self):
            calls.append('begin')
            return torch.zeros(1)

        def step(self, n, D):
            calls.append(('step', n, float(D[0])))
            return torch.full((1,), float(n + 1))

        def end(self):
            calls.append('end')
            return 'done'

    assert sharding.run_update_loop(S(), 3, lambda t: t.mul_(2)) == 'done'
    assert calls == ['begin', ('step', 0, 0.0), ('step', 1, 2.0), ('step', 2, 4.0), 'end']
