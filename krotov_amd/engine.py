"""Python handle on the HIP Krotov engine (thin layer over the C ABI).

PyTorch-ROCm is used for device memory and streams only: every array the
engine touches is a ``torch`` tensor in HBM whose ``data_ptr()`` is handed to
``libkrotov_hip.so``; all arithmetic happens in the HIP kernels.

The three sweeps map onto the reference's three ``parallel_map`` dispatches
(reference src/krotov/optimize.py:302-313, 413-425, 444-501).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .sharding import run_update_loop

__all__ = ['HipKrotovEngine', 'LAST_ENGINE']

_last_engine = None


def _is_sparse(op):
    return op is not None and hasattr(op, 'tocsr') and hasattr(op, 'nnz')


def LAST_ENGINE():
    """The most recently created engine (for benchmarks and tests that need the per-launch timings or the
    kernel family of an engine built inside optimize_pulses).  Held strongly until the next engine is created:
    nothing in a :class:`~krotov_amd.result.Result` is guaranteed to keep it alive."""
    return _last_engine


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError(
            "krotov_amd: no HIP device visible (torch.cuda.is_available() is False); "
            "the engine has no CPU fallback"
        )


class HipKrotovEngine:
    """K objectives x N-dimensional states x L controls on one GPU.

    Args:
        ops: list (K) of lists ``[H0, H_1, ..., H_L]`` of (N, N) complex arrays
            (NumPy or torch); ``None`` where a control does not occur in an
            objective.  Entries that are the *same object* are uploaded once
            and shared on the device.
        dt: (nt-1,) interval lengths.
        is_super: operators are Liouvillians acting on column-stacked vec(rho).
        op_norms: optional (K, 1+L) spectral-norm bounds; computed on the host
            with ``numpy.linalg.norm(., 2)`` per distinct operator when omitted.
        device: torch device (default: current CUDA/HIP device).
    """

    def __init__(self, ops, dt, is_super=False, op_norms=None, device=None, tol=0.0, theta_max=0.0):
        _require_gpu()
        self._lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.K = len(ops)
        self.L = len(ops[0]) - 1
        self.is_super = bool(is_super)
        dt = np.ascontiguousarray(np.asarray(dt, dtype=np.float64))
        self.nt = len(dt) + 1
        self._dt = dt
        self._handle = ctypes.c_void_p()
        self._op_tensors = {}
        self.N = None
        for k, row in enumerate(ops):
            if len(row) != 1 + self.L:
                raise ValueError("objective %d has %d operators, expected %d" % (k, len(row), 1 + self.L))
        with torch.cuda.device(self.device):
            if any(_is_sparse(op) for row in ops for op in row):
                norms = self._create_sparse(ops, dt, op_norms, tol, theta_max)
            else:
                norms = self._create_dense(ops, dt, op_norms, tol, theta_max)
        self.op_norms = norms.reshape(self.K, 1 + self.L)
        self.kernel = self._lib.kh_engine_kernel(self._handle).decode()
        # optional per-launch timing with HIP events on the launch stream
        self.profile = os.environ.get('KH_PROFILE', '0') == '1'
        self._events = {'forward': [], 'backward': [], 'update': []}
        global _last_engine
        _last_engine = self

    def _check_dim(self, shape):
        if len(shape) != 2 or shape[0] != shape[1]:
            raise ValueError("operators must be square matrices")
        if self.N is None:
            self.N = shape[0]
        elif shape[0] != self.N:
            raise ValueError("all operators must have the same dimension")

    def _norms(self, norms, op_norms):
        if op_norms is not None:
            norms = np.ascontiguousarray(np.asarray(op_norms, dtype=np.float64).reshape(-1))
            if norms.size != self.K * (1 + self.L):
                raise ValueError("op_norms must have K*(1+L) entries")
        return norms

    def _create_dense(self, ops, dt, op_norms, tol, theta_max):
        """Dense row-major operators; each distinct object is uploaded once."""
        n_ops = self.K * (1 + self.L)
        ptrs = (ctypes.c_void_p * n_ops)()
        norms = np.zeros(n_ops, dtype=np.float64)
        norms_cache = {}
        for k, row in enumerate(ops):
            for j, op in enumerate(row):
                idx = k * (1 + self.L) + j
                if op is None:
                    ptrs[idx] = None
                    continue
                key = id(op)
                if key not in self._op_tensors:
                    host = op.detach().cpu().numpy() if isinstance(op, torch.Tensor) else np.asarray(op)
                    host = np.ascontiguousarray(host, dtype=np.complex128)
                    self._check_dim(host.shape)
                    t = torch.from_numpy(host).to(self.device)
                    self._op_tensors[key] = (t, op)  # keep `op` alive: id() stays unique
                    norms_cache[key] = float(np.linalg.norm(host, 2)) if host.size else 0.0
                ptrs[idx] = self._op_tensors[key][0].data_ptr()
                norms[idx] = norms_cache[key]
        norms = self._norms(norms, op_norms)
        pr = _lib.kh_problem()
        pr.K, pr.N, pr.L, pr.nt = self.K, self.N, self.L, self.nt
        pr.is_super = 1 if self.is_super else 0
        pr.dt = dt.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        pr.ops = ctypes.cast(ptrs, ctypes.POINTER(ctypes.c_void_p))
        pr.op_norms = norms.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        pr.tol = float(tol)
        pr.theta_max = float(theta_max)
        _lib.check(self._lib.kh_engine_create(ctypes.byref(pr), ctypes.byref(self._handle)))
        return norms

    def _create_sparse(self, ops, dt, op_norms, tol, theta_max):
        """``scipy.sparse`` operators (any mix with dense ones, which are converted):
        CSR arrays of every distinct operator and of its conjugate transpose go to
        the device once; spectral norms are bounded by sqrt(||A||_1 ||A||_inf)."""
        import scipy.sparse as sp

        n_ops = self.K * (1 + self.L)
        fw = (_lib.kh_csr * n_ops)()
        bw = (_lib.kh_csr * n_ops)()
        norms = np.zeros(n_ops, dtype=np.float64)
        cache = {}

        def upload(mat):
            mat = sp.csr_matrix(mat, dtype=np.complex128)
            mat.sum_duplicates()
            arrays = (
                torch.from_numpy(np.ascontiguousarray(mat.indptr, dtype=np.int32)).to(self.device),
                torch.from_numpy(np.ascontiguousarray(mat.indices, dtype=np.int32)).to(self.device),
                torch.from_numpy(np.ascontiguousarray(mat.data, dtype=np.complex128)).to(self.device),
            )
            return arrays, int(mat.nnz)

        def fill(slot, arrays, nnz):
            slot.nnz = nnz
            slot.indptr, slot.indices, slot.data = (a.data_ptr() for a in arrays)

        for k, row in enumerate(ops):
            for j, op in enumerate(row):
                idx = k * (1 + self.L) + j
                if op is None:
                    continue
                key = id(op)
                if key not in cache:
                    mat = sp.csr_matrix(op.detach().cpu().numpy() if isinstance(op, torch.Tensor) else op)
                    self._check_dim(mat.shape)
                    a_fw, nnz = upload(mat)
                    a_bw, _ = upload(mat.conj().T)
                    bound = float(np.sqrt(abs(mat).sum(axis=0).max() * abs(mat).sum(axis=1).max())) if nnz else 0.0
                    cache[key] = (a_fw, a_bw, nnz, bound)
                    self._op_tensors[key] = ((a_fw, a_bw), op)
                a_fw, a_bw, nnz, bound = cache[key]
                fill(fw[idx], a_fw, nnz)
                fill(bw[idx], a_bw, nnz)
                norms[idx] = bound
        norms = self._norms(norms, op_norms)
        pr = _lib.kh_problem_csr()
        pr.K, pr.N, pr.L, pr.nt = self.K, self.N, self.L, self.nt
        pr.is_super = 1 if self.is_super else 0
        pr.dt = dt.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        pr.ops = fw
        pr.ops_adj = bw
        pr.op_norms = norms.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        pr.tol = float(tol)
        pr.theta_max = float(theta_max)
        _lib.check(self._lib.kh_engine_create_csr(ctypes.byref(pr), ctypes.byref(self._handle)))
        return norms

    def _timed(self, name):
        """Context manager recording HIP events around a launch when profiling."""
        eng = self

        class _T:
            def __enter__(self_t):
                if eng.profile:
                    self_t.a = torch.cuda.Event(enable_timing=True)
                    self_t.b = torch.cuda.Event(enable_timing=True)
                    self_t.a.record(torch.cuda.current_stream(eng.device))

            def __exit__(self_t, *exc):
                if eng.profile:
                    self_t.b.record(torch.cuda.current_stream(eng.device))
                    eng._events[name].append((self_t.a, self_t.b))
                return False

        return _T()

    def kernel_times_ms(self, reset=True):
        """Per-launch durations (ms) measured by HIP events since the last reset."""
        torch.cuda.synchronize(self.device)
        out = {k: [a.elapsed_time(b) for a, b in v] for k, v in self._events.items()}
        if reset:
            self._events = {k: [] for k in self._events}
        return out

    # -- helpers -----------------------------------------------------------
    def close(self):
        if getattr(self, '_handle', None) is not None and self._handle.value:
            self._lib.kh_engine_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def dev(self, x, dtype):
        """Contiguous tensor of ``dtype`` on the engine's device."""
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(np.asarray(x)), dtype=dtype).to(self.device).contiguous()

    def _c(self, x, shape):
        t = self.dev(x, torch.complex128)
        if tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (shape, tuple(t.shape)))
        return t

    def _f(self, x, shape):
        t = self.dev(x, torch.float64)
        if tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (shape, tuple(t.shape)))
        return t

    # -- sweeps ------------------------------------------------------------
    def forward(self, pulses, init, store=False):
        """Propagate ``init`` (K, N) over the grid under ``pulses`` (L, nt-1).

        Returns ``psi_T`` or ``(psi_T, states)`` with states (K, nt, N).
        """
        pulses = self._f(pulses, (self.L, self.nt - 1))
        init = self._c(init, (self.K, self.N))
        psi_T = torch.empty_like(init)
        states = torch.empty((self.K, self.nt, self.N), dtype=torch.complex128, device=self.device) if store else None
        with self._timed('forward'):
            _lib.check(self._lib.kh_forward_store(
                self._handle, pulses.data_ptr(), init.data_ptr(),
                states.data_ptr() if store else None, psi_T.data_ptr(), self._stream()))
        return (psi_T, states) if store else psi_T

    def backward(self, chi_T, pulses, out=None):
        """Backward sweep storing chi(t_n); returns (K, nt, N)."""
        pulses = self._f(pulses, (self.L, self.nt - 1))
        chi_T = self._c(chi_T, (self.K, self.N))
        if out is None:
            out = torch.empty((self.K, self.nt, self.N), dtype=torch.complex128, device=self.device)
        with self._timed('backward'):
            _lib.check(self._lib.kh_backward_store(
                self._handle, chi_T.data_ptr(), pulses.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def set_second_order(self, fw_prev=None, fw_store=None, sigma_vals=None):
        """Switch the following update sweeps to the second-order update
        (``fw_prev``, ``fw_store``: (K, nt, N) device tensors; ``sigma_vals``:
        sigma at the nt-1 interval mid-points), or back to first order (no
        arguments)."""
        if fw_prev is None:
            self._so = None
            _lib.check(self._lib.kh_set_second_order(self._handle, None, None, None))
            return
        fw_prev = self._c(fw_prev, (self.K, self.nt, self.N))
        if tuple(fw_store.shape) != (self.K, self.nt, self.N) or fw_store.dtype != torch.complex128:
            raise ValueError("fw_store must be a (K, nt, N) complex128 device tensor")
        sig = self._f(sigma_vals, (self.nt - 1,))
        self._so = (fw_prev, fw_store, sig)  # keep the tensors alive while the engine points at them
        _lib.check(self._lib.kh_set_second_order(
            self._handle, fw_prev.data_ptr(), fw_store.data_ptr(), sig.data_ptr()))

    def set_update_workgroups(self, max_workgroups=0):
        """Run the following single-launch update sweeps on at most ``max_workgroups`` workgroups (0: the engine's own
        choice again); returns the grid the next sweep will use (``kh_set_update_workgroups``).  Raises
        ``KrotovHipError`` (``KH_ERR_UNSUPPORTED``) for kernel families without such a form."""
        import ctypes

        chosen = ctypes.c_int32(0)
        _lib.check(self._lib.kh_set_update_workgroups(self._handle, int(max_workgroups), ctypes.byref(chosen)))
        return int(chosen.value)

    def forward_update(self, chi_store, chi_norms, init, guess, shape, lambdas):
        """Forward sweep with sequential update; returns ``(opt, psi_T, g_a)``."""
        chi_store = self._c(chi_store, (self.K, self.nt, self.N))
        chi_norms = self._f(chi_norms, (self.K,))
        init = self._c(init, (self.K, self.N))
        guess = self._f(guess, (self.L, self.nt - 1))
        shape = self._f(shape, (self.L, self.nt - 1))
        lambdas = self._f(lambdas, (self.L,))
        opt = torch.empty_like(guess)
        psi_T = torch.empty_like(init)
        g_a = torch.empty((self.L,), dtype=torch.float64, device=self.device)
        with self._timed('update'):
            _lib.check(self._lib.kh_forward_update(
                self._handle, chi_store.data_ptr(), chi_norms.data_ptr(), init.data_ptr(), guess.data_ptr(),
                shape.data_ptr(), lambdas.data_ptr(), opt.data_ptr(), psi_T.data_ptr(), g_a.data_ptr(),
                self._stream()))
        return opt, psi_T, g_a

    def forward_update_sharded(self, chi_store, chi_norms, init, guess, shape, lambdas, all_reduce,
                               graph_chunk=None):
        """The same sweep cut at the cross-objective sum: after every interval
        ``all_reduce(partial)`` (in place, L doubles on the device) must return
        the sum over all ranks -- ``torch.distributed.all_reduce`` on the
        ``nccl`` (= RCCL over xGMI) backend.

        With ``graph_chunk`` > 0 (default: env ``KH_GRAPH_CHUNK``, 64) a block of
        that many intervals -- all-reduce + ``kh_update_step_dev`` each -- is
        captured once as a HIP graph and replayed, so the host issues one graph
        launch per block instead of three launches per interval.  The interval
        index lives in device memory, which makes every replay identical; replays
        past the last interval are no-ops.  ``graph_chunk=0`` runs the plain loop.
        """
        if graph_chunk is None:
            graph_chunk = int(os.environ.get('KH_GRAPH_CHUNK', '64'))
        nt, L, K, N = self.nt, self.L, self.K, self.N
        # persistent buffers: stable addresses let the captured graph be reused
        b = getattr(self, '_sh', None)
        if b is None:
            c128, f64, dev = torch.complex128, torch.float64, self.device
            b = self._sh = dict(
                chi_norms=torch.empty((K,), dtype=f64, device=dev),
                init=torch.empty((K, N), dtype=c128, device=dev),
                guess=torch.empty((L, nt - 1), dtype=f64, device=dev),
                shape=torch.empty((L, nt - 1), dtype=f64, device=dev),
                lambdas=torch.empty((L,), dtype=f64, device=dev),
                opt=torch.empty((L, nt - 1), dtype=f64, device=dev),
                psi_T=torch.empty((K, N), dtype=c128, device=dev),
                g_a=torch.empty((L,), dtype=f64, device=dev),
                partial=torch.zeros((L,), dtype=f64, device=dev),
                n_dev=torch.zeros((1,), dtype=torch.int32, device=dev),
                graph=None, graph_key=None,
            )
        chi_store = self._c(chi_store, (K, nt, N))
        b['chi_norms'].copy_(self._f(chi_norms, (K,)))
        b['init'].copy_(self._c(init, (K, N)))
        b['guess'].copy_(self._f(guess, (L, nt - 1)))
        b['shape'].copy_(self._f(shape, (L, nt - 1)))
        b['lambdas'].copy_(self._f(lambdas, (L,)))
        chi_norms, init, guess, shape, lambdas = b['chi_norms'], b['init'], b['guess'], b['shape'], b['lambdas']
        opt, psi_T, g_a, partial, n_dev = b['opt'], b['psi_T'], b['g_a'], b['partial'], b['n_dev']
        lib, h, eng = self._lib, self._handle, self

        def step_dev():
            _lib.check(lib.kh_update_step_dev(
                h, n_dev.data_ptr(), partial.data_ptr(), chi_store.data_ptr(), chi_norms.data_ptr(),
                shape.data_ptr(), lambdas.data_ptr(), opt.data_ptr(), g_a.data_ptr(), partial.data_ptr(),
                eng._stream()))

        class _Stepper:
            """kh_update_begin / kh_update_step / kh_update_end of the C ABI."""

            def begin(self_s):
                _lib.check(lib.kh_update_begin(
                    h, chi_store.data_ptr(), chi_norms.data_ptr(), init.data_ptr(), guess.data_ptr(),
                    opt.data_ptr(), g_a.data_ptr(), partial.data_ptr(), eng._stream()))
                return partial

            def step(self_s, n, D):
                _lib.check(lib.kh_update_step(
                    h, n, D.data_ptr(), chi_store.data_ptr(), chi_norms.data_ptr(), shape.data_ptr(),
                    lambdas.data_ptr(), opt.data_ptr(), g_a.data_ptr(), partial.data_ptr(), eng._stream()))
                return partial

            def end(self_s):
                _lib.check(lib.kh_update_end(h, psi_T.data_ptr(), eng._stream()))

        with self._timed('update'):
            done = False
            # (second order: kh_update_step_dev bakes the trajectory / sigma pointers of kh_set_second_order
            # into the captured launches, and those buffers are swapped every iteration -- no replay there)
            if graph_chunk > 0 and nt - 1 > 2 * graph_chunk and getattr(self, '_so', None) is None:
                try:
                    stepper = _Stepper()
                    stepper.begin()
                    n_dev.zero_()
                    # first interval eagerly: also warms up the communicator outside the capture
                    all_reduce(partial)
                    step_dev()
                    key = (chi_store.data_ptr(), graph_chunk)
                    if b['graph'] is None or b['graph_key'] != key:
                        torch.cuda.synchronize(self.device)
                        g = torch.cuda.CUDAGraph()
                        # the capture itself runs the block once, but on a scratch copy of
                        # nothing: captured work is only recorded, not executed
                        with torch.cuda.graph(g):
                            for _ in range(graph_chunk):
                                all_reduce(partial)
                                step_dev()
                        b['graph'], b['graph_key'] = g, key
                    replays = (nt - 2 + graph_chunk - 1) // graph_chunk
                    for _ in range(replays):
                        b['graph'].replay()
                    stepper.end()
                    done = True
                except Exception as exc:  # capture unsupported: plain loop below
                    import warnings

                    warnings.warn("krotov_amd: graph capture of the sharded sweep failed (%s); "
                                  "falling back to per-interval launches" % exc)
                    b['graph'] = None
                    torch.cuda.synchronize(self.device)
            if not done:
                run_update_loop(_Stepper(), nt - 1, all_reduce)
        return opt.clone(), psi_T.clone(), g_a.clone()

    def enable_p2p(self, group, rounds=8):
        """Set up the device-side exchange across the ranks of ``group`` (one per
        GPU of a node): allocate this rank's window, trade IPC handles, map the
        peers' windows and run the in-kernel self-test.  Collective.  Returns True
        when EVERY rank passed -- only then does :meth:`forward_update` include the
        cross-GPU stage; otherwise the engine stays on the per-interval RCCL path."""
        import torch.distributed as dist

        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lib, h = self._lib, self._handle

        def all_ok(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return bool(flag.item())

        def stage(name, ok):
            """Collective verdict of one set-up stage; on failure every rank remembers which stage failed and -- where
            it was this rank -- the library's own words (``p2p_why``: printed by ``bench.py --gpus N``)."""
            mine_failed = not ok
            if all_ok(ok):
                return True
            self.p2p_why = "peer windows not used: %s failed%s" % (
                name, (" on this rank (%s)" % lib.kh_last_error().decode('utf-8', 'replace')) if mine_failed else " on another rank")
            lib.kh_p2p_disable(h)
            return False

        self.p2p_why = None
        handle = (ctypes.c_ubyte * 64)()
        ok = lib.kh_p2p_create_window(h, world, rank, handle) == 0
        mine = torch.tensor(list(bytes(handle)), dtype=torch.uint8, device=self.device)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
        if not stage('kh_p2p_create_window', ok):
            return False
        blob = b''.join(bytes(g.cpu().numpy().tobytes()) for g in gathered)
        buf = (ctypes.c_ubyte * len(blob)).from_buffer_copy(blob)
        ok = lib.kh_p2p_open_peers(h, buf) == 0
        if not stage('kh_p2p_open_peers (hipIpcOpenMemHandle)', ok):
            return False
        torch.cuda.synchronize(self.device)
        dist.barrier(group=group)  # every window exists and is mapped before anyone writes
        ok = lib.kh_p2p_selftest(h, int(rounds), self._stream()) == 0
        if not stage('kh_p2p_selftest (in-kernel exchange over the windows)', ok):
            return False
        return True

    def disable_p2p(self):
        self._lib.kh_p2p_disable(self._handle)

    def p2p_stats(self):
        """``kh_p2p_stats``: us per interval workgroup 0 waited inside its GPU / across the GPUs in the last sharded
        single-launch update sweep, us per round of the set-up self-test, ranks."""
        buf = (ctypes.c_double * 4)()
        torch.cuda.synchronize(self.device)
        _lib.check(self._lib.kh_p2p_stats(self._handle, buf))
        return dict(local_wait_us=buf[0], cross_gpu_wait_us=buf[1], selftest_round_us=buf[2], ranks=int(buf[3]))

    def tau(self, targets, psi_T):
        targets = self._c(targets, (self.K, self.N))
        psi_T = self._c(psi_T, (self.K, self.N))
        out = torch.empty((self.K,), dtype=torch.complex128, device=self.device)
        _lib.check(self._lib.kh_tau(self._handle, targets.data_ptr(), psi_T.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def chi_boundary(self, targets, psi_T, c, d):
        """Normalised boundary co-states ``(c_k target_k + d_k psi_k(T)) / ||.||`` and
        their norms, ``(K, N)`` and ``(K,)`` device tensors (kh_chi_boundary)."""
        targets = self._c(targets, (self.K, self.N))
        psi_T = self._c(psi_T, (self.K, self.N))
        c = self._c(c, (self.K,))
        d = self._c(d, (self.K,))
        chi = torch.empty((self.K, self.N), dtype=torch.complex128, device=self.device)
        norms = torch.empty((self.K,), dtype=torch.float64, device=self.device)
        _lib.check(self._lib.kh_chi_boundary(
            self._handle, targets.data_ptr(), psi_T.data_ptr(), c.data_ptr(), d.data_ptr(), chi.data_ptr(),
            norms.data_ptr(), self._stream()))
        return chi, norms

    def check(self):
        """Synchronise and raise if an in-kernel exchange timed out."""
        torch.cuda.synchronize(self.device)
        _lib.check(self._lib.kh_check(self._handle))

    def stats(self):
        buf = (ctypes.c_double * 4)()
        torch.cuda.synchronize(self.device)
        _lib.check(self._lib.kh_last_stats(self._handle, buf))
        return dict(matvecs=buf[0], intervals=buf[1], workgroups=buf[2])
