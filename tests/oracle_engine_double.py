"""Oracle-backed stand-in for ``krotov_amd.engine.HipKrotovEngine`` (TESTS ONLY).

Same Python interface, CPU torch tensors, arithmetic by the NumPy oracle.  It lets
the world_size-2 ``gloo`` tests drive the product's multi-rank host logic
(sharding, tau all-gather, the per-interval all-reduce loop of
``krotov_amd.sharding.run_update_loop``) on a machine without GPUs.  The stepwise
update follows the C ABI's begin/step/end contract (include/krotov_hip.h).
"""
import numpy as np
import torch

from krotov_amd.sharding import run_update_loop
from oracle import krotov_oracle as ko


class OracleEngineDouble:
    def __init__(self, ops, dt, is_super=False, **kw):
        def dense(o):  # (CSR operators of the sparse engine form: the oracle works on dense arrays)
            return np.asarray(o.toarray() if hasattr(o, 'toarray') else o, dtype=np.complex128)

        self.ops = [[None if o is None else dense(o) for o in row] for row in ops]
        self.K, self.L = len(ops), len(ops[0]) - 1
        self.N = self.ops[0][0].shape[0]
        self.dt = np.asarray(dt, dtype=np.float64)
        self.nt = len(self.dt) + 1
        self.is_super = bool(is_super)
        self.device = torch.device('cpu')
        self.kernel = 'oracle-double'

    def dev(self, x, dtype):
        if isinstance(x, torch.Tensor):
            return x.to(dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(np.asarray(x)), dtype=dtype)

    def _prob(self, init):
        tl = np.concatenate([[0.0], np.cumsum(self.dt)])
        return ko.OracleProblem(self.ops, init, np.zeros_like(init), tl, self.is_super)

    def forward(self, pulses, init, store=False):
        init = np.asarray(self.dev(init, torch.complex128).numpy())
        pulses = list(self.dev(pulses, torch.float64).numpy())
        if store:
            fw, states = ko.forward_propagation(self._prob(init), pulses, store=True)
            return torch.from_numpy(fw), torch.from_numpy(states)
        return torch.from_numpy(ko.forward_propagation(self._prob(init), pulses))

    _so = None

    def set_second_order(self, fw_prev=None, fw_store=None, sigma_vals=None):
        """As HipKrotovEngine.set_second_order: the following update sweeps add 0.5 sigma_n <phi - phi_prev|mu|phi>
        and write their trajectory into ``fw_store`` (in place: the caller swaps the two buffers)."""
        self._so = None if fw_prev is None else (fw_prev, fw_store, self.dev(sigma_vals, torch.float64).numpy())

    def backward(self, chi_T, pulses, out=None):
        chi_T = self.dev(chi_T, torch.complex128).numpy()
        res = ko.backward_sweep(self._prob(chi_T), chi_T, list(self.dev(pulses, torch.float64).numpy()))
        return torch.from_numpy(res)

    def forward_update(self, chi_store, chi_norms, init, guess, shape, lambdas):
        if self._so is not None:
            fw_prev, fw_store, sig = self._so
            init_h = np.asarray(self.dev(init, torch.complex128).numpy())
            opt, fw, g_a, out = ko.forward_update_sweep(
                self._prob(init_h), self.dev(chi_store, torch.complex128).numpy(),
                self.dev(chi_norms, torch.float64).numpy(), list(self.dev(guess, torch.float64).numpy()),
                list(self.dev(shape, torch.float64).numpy()), list(self.dev(lambdas, torch.float64).numpy()),
                sigma_vals=sig, fw_prev=fw_prev.numpy(), store=True)
            fw_store.copy_(torch.from_numpy(out))
            return torch.from_numpy(np.array(opt)), torch.from_numpy(fw), torch.from_numpy(g_a)
        return self.forward_update_sharded(chi_store, chi_norms, init, guess, shape, lambdas, lambda t: t)

    def forward_update_sharded(self, chi_store, chi_norms, init, guess, shape, lambdas, all_reduce, graph_chunk=None):
        chi = self.dev(chi_store, torch.complex128).numpy()
        norms = self.dev(chi_norms, torch.float64).numpy()
        guess = self.dev(guess, torch.float64).numpy()
        shape = self.dev(shape, torch.float64).numpy()
        lam = self.dev(lambdas, torch.float64).numpy()
        phi = [v.copy() for v in self.dev(init, torch.complex128).numpy()]
        opt = guess.copy()
        g_a = np.zeros(self.L)
        partial = torch.zeros(self.L, dtype=torch.float64)
        mu = 1j if self.is_super else 1.0
        eng = self

        def local_partials(n):
            for l in range(eng.L):
                acc = 0.0
                for k in range(eng.K):
                    op = eng.ops[k][1 + l]
                    if op is not None:
                        acc += norms[k] * (mu * np.vdot(chi[k, n], op @ phi[k])).imag
                partial[l] = acc

        class Stepper:
            def begin(self):
                local_partials(0)
                return partial

            def step(self, n, D):
                for l in range(eng.L):
                    d1 = float(D[l])
                    opt[l, n] = guess[l, n] + shape[l, n] / lam[l] * d1
                    g_a[l] += shape[l, n] / lam[l] * d1 * d1 * eng.dt[n]
                for k in range(eng.K):
                    phi[k] = ko.step(eng.ops[k], list(opt[:, n]), eng.dt[n], phi[k], eng.is_super, False)
                if n + 1 < eng.nt - 1:
                    local_partials(n + 1)
                return partial

            def end(self):
                return None

        run_update_loop(Stepper(), self.nt - 1, all_reduce)
        return torch.from_numpy(opt), torch.from_numpy(np.array(phi)), torch.from_numpy(g_a)

    def tau(self, targets, psi_T):
        t = self.dev(targets, torch.complex128).numpy()
        p = self.dev(psi_T, torch.complex128).numpy()
        return torch.from_numpy(np.array([np.vdot(a, b) for a, b in zip(t, p)]))

    def chi_boundary(self, targets, psi_T, c, d):
        t = self.dev(targets, torch.complex128).numpy()
        p = self.dev(psi_T, torch.complex128).numpy()
        c = self.dev(c, torch.complex128).numpy()
        d = self.dev(d, torch.complex128).numpy()
        v = c[:, None] * t + d[:, None] * p
        norms = np.linalg.norm(v, axis=1)
        return torch.from_numpy(v / norms[:, None]), torch.from_numpy(norms)

    def check(self):
        pass

    def enable_p2p(self, group, rounds=8):
        return False  # no peer windows on the CPU: the per-interval all-reduce path is exercised

    def close(self):
        pass
