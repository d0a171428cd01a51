"""LGSSMEngine — thin object wrapper over the rxhip C ABI (one handle = one batch of chains)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import RxHipError, c_double_p


def _c(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    return a.ctypes.data_as(c_double_p)


class LGSSMEngine:
    """Batch of linear Gaussian state-space factor graphs on one MI355X.

    A, B, P, Q, m0, V0 may carry a leading model axis ([n_models, ...]); chain_model maps chains
    to models.  P is the state-noise covariance, Q the observation-noise covariance (the
    benchmark notebook's naming; test/models/statespace/mlgssm_test.jl swaps the two names).
    """

    def __init__(self, A, B, P, Q, m0, V0, T, n_chains=1, chain_model=None, prior_through_transition=False,
                 segments=0, device=-1, stream=None, horizon=0, allow_missing=False, step_model=None, state_offset=None, obs_offset=None):
        L = _lib.lib()
        A = _c(A)
        B = _c(B)
        if A.ndim == 2:
            A, B, P, Q, m0, V0 = (np.asarray(x, dtype=np.float64)[None] for x in (A, B, P, Q, m0, V0))
        self.n_models = A.shape[0]
        self.d = A.shape[-1]
        self.dy = _c(B).shape[-2]
        self.T = int(T)
        self.horizon = int(horizon)
        self.n_chains = int(n_chains)
        d, dy, M = self.d, self.dy, self.n_models
        self._keep = [_c(A, (M, d, d)), _c(B, (M, dy, d)), _c(P, (M, d, d)), _c(Q, (M, dy, dy)), _c(m0, (M, d)),
                      _c(V0, (M, d, d))]
        desc = _lib.LgssmDesc()
        desc.d, desc.dy, desc.T, desc.n_chains, desc.n_models = d, dy, self.T, self.n_chains, M
        desc.prior_through_transition = int(bool(prior_through_transition))
        desc.A, desc.B, desc.P, desc.Q, desc.m0, desc.V0 = (_p(x) for x in self._keep)
        if chain_model is not None:
            cm = np.ascontiguousarray(chain_model, dtype=np.int32)
            if cm.shape != (self.n_chains,):
                raise ValueError("chain_model must have one entry per chain")
            self._keep.append(cm)
            desc.chain_model = cm.ctypes.data_as(_lib.c_int32_p)
        desc.segments = int(segments)
        desc.device = int(device)
        desc.stream = ctypes.c_void_p(stream) if stream else None
        desc.horizon = self.horizon
        desc.allow_missing = int(bool(allow_missing))
        if step_model is not None:  # time-varying constants: the model of every time index
            sm = np.ascontiguousarray(step_model, dtype=np.int32)
            if sm.shape != (self.T + self.horizon,):
                raise ValueError("step_model must have one entry per time index (T + horizon)")
            self._keep.append(sm)
            desc.step_model = sm.ctypes.data_as(_lib.c_int32_p)
        # known inputs: x[t] ~ N(A x[t-1] + c[t], P), y[t] ~ N(B x[t] + d[t], Q); a single vector is the same offset at every step
        for name, off, k in (("state_offset", state_offset, d), ("obs_offset", obs_offset, dy)):
            if off is not None:
                a = _c(np.broadcast_to(np.asarray(off, dtype=np.float64), (self.T + self.horizon, k)))
                self._keep.append(a)
                setattr(desc, name, _p(a))
        self._h = ctypes.c_void_p()
        st = self._create(L, desc)
        if st != _lib.OK:
            msg = L.rxhip_last_error(self._h).decode() if self._h else (L.rxhip_lowering_error().decode() if st == _lib.ERR_UNSUPPORTED else "") or L.rxhip_status_string(st).decode()
            if self._h:
                L.rxhip_destroy(self._h)
                self._h = None
            raise RxHipError(st, msg or L.rxhip_status_string(st).decode())
        self._data_ref = None
        self._iters = 0

    def _create(self, L, desc):
        return L.rxhip_lgssm_create(ctypes.byref(desc), ctypes.byref(self._h))

    # -- plumbing ---------------------------------------------------------------------------
    def _chk(self, st):
        if st != _lib.OK:
            raise RxHipError(st, _lib.lib().rxhip_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().rxhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- data -------------------------------------------------------------------------------
    def set_data(self, y, layout="time_chain"):
        """y: [T][chain][dy] (layout='time_chain') or [chain][T][dy] ('chain_time'), host array."""
        y = _c(y)
        lay = _lib.LAYOUT_TIME_CHAIN if layout == "time_chain" else _lib.LAYOUT_CHAIN_TIME
        self._chk(_lib.lib().rxhip_set_data(self._h, _lib.VAR_Y, _p(y), y.size, lay))

    def set_inputs(self, u, layout="time_chain"):
        """Data inputs u[t] of `A * x[t-1] + B_u * u[t]` (engines built from a graph with such inputs): [T][chain][du] / [chain][T][du]."""
        u = _c(u)
        lay = _lib.LAYOUT_TIME_CHAIN if layout == "time_chain" else _lib.LAYOUT_CHAIN_TIME
        self._chk(_lib.lib().rxhip_set_data(self._h, _lib.VAR_U, _p(u), u.size, lay))

    def set_data_device(self, ptr, n, layout="time_chain", keepalive=None):
        """Observations already in device memory (e.g. a torch tensor's data_ptr())."""
        lay = _lib.LAYOUT_TIME_CHAIN if layout == "time_chain" else _lib.LAYOUT_CHAIN_TIME
        self._data_ref = keepalive
        self._chk(_lib.lib().rxhip_set_data_device(self._h, _lib.VAR_Y, ctypes.c_void_p(ptr), n, lay))

    # -- inference --------------------------------------------------------------------------
    def run(self, iterations=1, free_energy=True):
        self._chk(_lib.lib().rxhip_run(self._h, int(iterations), int(bool(free_energy))))
        self._iters = int(iterations)

    def infer(self, y, iterations=1, free_energy=True, want_cov=True, filtering=False):
        """set_data + run + marginals (+ per-chain free energy) in one round trip (rxhip_lgssm_infer); y: [T][chain][dy].
        Returns mean [T+H][chain][d], cov | None, fe [chain] | None."""
        y = _c(y)
        To, C, d = self.T + self.horizon, self.n_chains, self.d
        mean = np.empty((To, C, d))
        cov = np.empty((To, C, d, d)) if want_cov else None
        fe = np.empty(C) if free_energy else None
        self._chk(_lib.lib().rxhip_lgssm_infer(self._h, _p(y), y.size, int(iterations), int(bool(free_energy)), int(bool(filtering)), _p(mean),
                                               _p(cov) if want_cov else None, _p(fe) if free_energy else None))
        self._iters = int(iterations)
        return mean, cov, fe

    def run_async(self, iterations=1, free_energy=True):
        self._chk(_lib.lib().rxhip_run_async(self._h, int(iterations), int(bool(free_energy))))
        self._iters = int(iterations)

    def run_filter(self, free_energy=True):
        """Streaming / filtering run (rxhip_run_filter): afterwards `marginals()` holds q(x_t | y_1..t) and
        `free_energy()` ONE value — Σ_chains of the mean-over-observations free energy."""
        self._chk(_lib.lib().rxhip_run_filter(self._h, int(bool(free_energy))))
        self._iters = 1

    def run_filter_async(self, free_energy=True):
        self._chk(_lib.lib().rxhip_run_filter_async(self._h, int(bool(free_energy))))
        self._iters = 1

    def filter_step(self, y, want_cov=True, free_energy=True):
        """One observation of every chain (y: [chain][dy]; NaN = missing) through the streaming driver (rxhip_filter_step):
        returns the posterior mean [chain][d], covariance [chain][d][d] | None, −log p(y_k | y_<k) [chain] | None."""
        y = _c(y, (self.n_chains, self.dy))
        mean = np.empty((self.n_chains, self.d))
        cov = np.empty((self.n_chains, self.d, self.d)) if want_cov else None
        fe = np.empty(self.n_chains) if free_energy else None
        self._chk(_lib.lib().rxhip_filter_step(self._h, _p(y), _p(mean), _p(cov) if want_cov else None, _p(fe) if free_energy else None))
        return mean, cov, fe

    def set_offsets(self, state_offset=None, obs_offset=None):
        """New known inputs (rxhip_lgssm_set_offsets); the engine must have been created with offsets."""
        To = self.T + self.horizon
        cx = None if state_offset is None else _c(np.broadcast_to(np.asarray(state_offset, dtype=np.float64), (To, self.d)))
        cy = None if obs_offset is None else _c(np.broadcast_to(np.asarray(obs_offset, dtype=np.float64), (To, self.dy)))
        self._chk(_lib.lib().rxhip_lgssm_set_offsets(self._h, _p(cx) if cx is not None else None, _p(cy) if cy is not None else None))

    def set_chain_offsets(self, state_offset=None, obs_offset=None, layout="time_chain"):
        """Known inputs per chain (rxhip_lgssm_set_chain_offsets): [T+horizon][chain][d] / [..][dy] ('time_chain') or
        [chain][T+horizon][·] ('chain_time')."""
        To, C = self.T + self.horizon, self.n_chains
        shp = (lambda k: (To, C, k)) if layout == "time_chain" else (lambda k: (C, To, k))
        cx = None if state_offset is None else _c(state_offset, shp(self.d))
        cy = None if obs_offset is None else _c(obs_offset, shp(self.dy))
        lay = _lib.LAYOUT_TIME_CHAIN if layout == "time_chain" else _lib.LAYOUT_CHAIN_TIME
        self._chk(_lib.lib().rxhip_lgssm_set_chain_offsets(self._h, _p(cx) if cx is not None else None,
                                                           _p(cy) if cy is not None else None, lay))

    def filter_reset(self):
        self._chk(_lib.lib().rxhip_filter_reset(self._h))

    def sync(self):
        self._chk(_lib.lib().rxhip_sync(self._h))

    def predictions(self, layout="time_chain", want_cov=True):
        """Messages toward y[1..T+horizon] (`predictvars`): mean [T+H][chain][dy], cov [...][dy][dy] (rxhip_get_predictions)."""
        dy, T, C = self.dy, self.T + getattr(self, "horizon", 0), self.n_chains
        lay = _lib.LAYOUT_TIME_CHAIN if layout == "time_chain" else _lib.LAYOUT_CHAIN_TIME
        shp = (T, C) if layout == "time_chain" else (C, T)
        mean = np.empty(shp + (dy,))
        cov = np.empty(shp + (dy, dy)) if want_cov else None
        self._chk(_lib.lib().rxhip_get_predictions(self._h, _lib.VAR_Y, _p(mean), _p(cov) if want_cov else None, lay))
        return mean, cov

    def node_marginals(self, layout="time_chain"):
        """Node-local joints q(x[t], A x[t-1]) of the transition nodes (rxhip_get_node_marginals): mean [T-1][chain][2d],
        cov [T-1][chain][2d][2d] in (out, μ) order."""
        d2, T, C = 2 * self.d, self.T - 1, self.n_chains
        lay = _lib.LAYOUT_TIME_CHAIN if layout == "time_chain" else _lib.LAYOUT_CHAIN_TIME
        shp = (T, C) if layout == "time_chain" else (C, T)
        mean, cov = np.empty(shp + (d2,)), np.empty(shp + (d2, d2))
        self._chk(_lib.lib().rxhip_get_node_marginals(self._h, _lib.NODE_MVNORMAL_MEAN_COV, _p(mean), _p(cov), lay))
        return mean, cov

    def marginals(self, layout="time_chain", want_cov=True):
        d, T, C = self.d, self.T + getattr(self, "horizon", 0), self.n_chains
        lay = _lib.LAYOUT_TIME_CHAIN if layout == "time_chain" else _lib.LAYOUT_CHAIN_TIME
        shp = (T, C) if layout == "time_chain" else (C, T)
        mean = np.empty(shp + (d,))
        cov = np.empty(shp + (d, d)) if want_cov else None
        self._chk(_lib.lib().rxhip_get_marginals(self._h, _lib.VAR_X, _p(mean), _p(cov) if want_cov else None, lay))
        return mean, cov

    def marginals_of_chains(self, chains, want_cov=True):
        """Posteriors of the selected chains only, chain-major: mean [n][T][d], cov [n][T][d][d] (strided gather on the
        device + one copy; rxhip_get_marginals_chains)."""
        ch = np.ascontiguousarray(chains, dtype=np.int64).ravel()
        T = self.T + getattr(self, "horizon", 0)
        mean = np.empty((ch.size, T, self.d))
        cov = np.empty((ch.size, T, self.d, self.d)) if want_cov else None
        self._chk(_lib.lib().rxhip_get_marginals_chains(self._h, _lib.VAR_X, ch.ctypes.data_as(_lib.c_int64_p), ch.size,
                                                         _p(mean), _p(cov) if want_cov else None))
        return mean, cov

    def marginals_device(self):
        m, c = ctypes.c_void_p(), ctypes.c_void_p()
        self._chk(_lib.lib().rxhip_get_marginals_device(self._h, _lib.VAR_X, ctypes.byref(m), ctypes.byref(c)))
        return m.value, c.value

    def free_energy(self):
        out = np.empty(self._iters)
        self._chk(_lib.lib().rxhip_get_free_energy(self._h, _p(out)))
        return out

    def free_energy_per_chain(self):
        out = np.empty(self.n_chains)
        self._chk(_lib.lib().rxhip_get_free_energy_per_chain(self._h, _p(out)))
        return out

    def free_energy_device(self):
        p = ctypes.c_void_p()
        self._chk(_lib.lib().rxhip_get_free_energy_device(self._h, ctypes.byref(p)))
        return p.value

    def copy_free_energy_to_device(self, dst_ptr):
        self._chk(_lib.lib().rxhip_copy_free_energy_to_device(self._h, ctypes.c_void_p(dst_ptr)))

    def allreduce_free_energy(self, comm):
        """Sum the free energies of the last run over all ranks of the RCCL communicator `comm` (a `Communicator` or a raw
        ncclComm_t address), in place on the device, on the engine's stream."""
        self._chk(_lib.lib().rxhip_allreduce_free_energy(self._h, ctypes.c_void_p(int(getattr(comm, "handle", comm) or 0))))

    def counters(self):
        r, p, m = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._chk(_lib.lib().rxhip_counters(self._h, ctypes.byref(r), ctypes.byref(p), ctypes.byref(m)))
        return {"rule_calls": r.value, "products": p.value, "marginals": m.value}

    # -- measurement ------------------------------------------------------------------------
    def set_profiling(self, on=True):
        self._chk(_lib.lib().rxhip_set_profiling(self._h, int(bool(on))))

    def reset_kernel_times(self):
        self._chk(_lib.lib().rxhip_reset_kernel_times(self._h))

    def kernel_times(self):
        ms = (ctypes.c_double * _lib.K_COUNT)()
        n = (ctypes.c_uint64 * _lib.K_COUNT)()
        self._chk(_lib.lib().rxhip_get_kernel_times(self._h, ms, n))
        return {name: {"ms_avg": ms[i], "launches": n[i]} for i, name in enumerate(_lib.KERNEL_NAMES)}

    def model_tables_ms(self):
        """device time of the once-per-engine kernels (tables that depend on the model only); 0 without such tables"""
        ms = ctypes.c_double()
        self._chk(_lib.lib().rxhip_get_model_tables_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def create_stages(self):
        """host milliseconds of the engine's creation by stage (rxhip_get_create_stages)"""
        ms = (ctypes.c_double * 4)()
        self._chk(_lib.lib().rxhip_get_create_stages(self._h, ms))
        return {"tables_host_ms": ms[0], "tables_device_ms": ms[1], "upload_ms": ms[2], "alloc_ms": ms[3]}

    def set_fixed_point_exits(self, enabled):
        """False: every recursion of this engine's later sweeps runs in full — no frozen stretches, no early exits (rxhip_set_fixed_point_exits)"""
        self._chk(_lib.lib().rxhip_set_fixed_point_exits(self._h, 1 if enabled else 0))

    def set_covariance_mode(self, mode):
        """0: every sweep writes the covariance of every chain (default); 1: shared-model batches on the MFMA path write the per-chain
        array when it is asked for (rxhip_set_covariance_mode)"""
        self._chk(_lib.lib().rxhip_set_covariance_mode(self._h, int(mode)))

    def schedule(self):
        s, l = ctypes.c_int32(), ctypes.c_int64()
        self._chk(_lib.lib().rxhip_get_schedule(self._h, ctypes.byref(s), ctypes.byref(l)))
        return {"segments": s.value, "segment_len": l.value}

    def stream(self):
        p = ctypes.c_void_p()
        self._chk(_lib.lib().rxhip_get_stream(self._h, ctypes.byref(p)))
        return p.value


class LGSSMNoiseEngine(LGSSMEngine):
    """State-space chains with an UNKNOWN observation-noise precision, `W ~ Wishart(nu0, S0); y[t] ~ MvNormal(μ = B * x[t], Λ = W)` with
    `q(x, W) = q(x) q(W)` (include/rxhip.h rxhip_lgssm_noise_create): every chain its own W.  run(iterations, free_energy) alternates one
    belief-propagation sweep of all chains with the Wishart update of all chains, on the device; free_energy() per iteration,
    noise_posterior() -> (nu [chains], V [chains][dy][dy]) after the last one."""

    def __init__(self, A, B, P, m0, V0, T, nu0, S0, init_nu=None, init_V=None, n_chains=1, prior_through_transition=False, segments=0,
                 device=-1, stream=None):
        dy = _c(B).shape[-2]
        self._pri = _lib.NoisePrior()
        self._pri_keep = [_c(S0, (dy, dy)), _c(S0 if init_V is None else init_V, (dy, dy))]
        self._pri.nu0, self._pri.init_nu = float(nu0), float(nu0 if init_nu is None else init_nu)
        self._pri.S0, self._pri.init_V = _p(self._pri_keep[0]), _p(self._pri_keep[1])
        super().__init__(A, B, P, np.eye(dy), m0, V0, T, n_chains=n_chains, prior_through_transition=prior_through_transition,
                         segments=segments, device=device, stream=stream)

    def _create(self, L, desc):
        return L.rxhip_lgssm_noise_create(ctypes.byref(desc), ctypes.byref(self._pri), ctypes.byref(self._h))

    def continue_runs(self, on=True):
        """later run() calls go on from the current q(W) instead of the initial marginal (rxhip_lgssm_noise_continue)"""
        self._chk(_lib.lib().rxhip_lgssm_noise_continue(self._h, int(bool(on))))

    def noise_posterior(self):
        nu, V = np.empty(self.n_chains), np.empty((self.n_chains, self.dy, self.dy))
        self._chk(_lib.lib().rxhip_lgssm_noise_get(self._h, _p(nu), _p(V)))
        return nu, V


class GMMEngine:
    """Mean-field VMP for the univariate Gaussian mixture on one MI355X (see include/rxhip.h rxhip_gmm_desc).
    K = 1 is the iid Gaussian with unknown mean and precision."""

    def __init__(self, N, mu0, v0, a0, b0, alpha0, init_m_mean, init_m_var, init_p_shape, init_p_rate, init_s_alpha,
                 materialize_responsibilities=False, device=-1, stream=None):
        L = _lib.lib()
        arrs = [_c(a).ravel() for a in (mu0, v0, a0, b0, alpha0, init_m_mean, init_m_var, init_p_shape, init_p_rate,
                                        init_s_alpha)]
        self.K = arrs[0].size
        if any(a.size != self.K for a in arrs):
            raise ValueError("all prior / initialisation arrays must have K entries")
        self.N = int(N)
        self._keep = arrs
        desc = _lib.GmmDesc()
        desc.N, desc.K = self.N, self.K
        for name, a in zip(("mu0", "v0", "a0", "b0", "alpha0", "init_m_mean", "init_m_var", "init_p_shape", "init_p_rate",
                            "init_s_alpha"), arrs):
            setattr(desc, name, _p(a))
        desc.materialize_responsibilities = int(bool(materialize_responsibilities))
        desc.device = int(device)
        desc.stream = ctypes.c_void_p(stream) if stream else None
        self._h = ctypes.c_void_p()
        st = L.rxhip_gmm_create(ctypes.byref(desc), ctypes.byref(self._h))
        if st != _lib.OK:
            msg = L.rxhip_last_error(self._h).decode() if self._h else L.rxhip_status_string(st).decode()
            if self._h:
                L.rxhip_destroy(self._h)
                self._h = None
            raise RxHipError(st, msg or L.rxhip_status_string(st).decode())
        self._iters = 0
        self._data_ref = None

    _chk = LGSSMEngine._chk
    close = LGSSMEngine.close
    __del__ = LGSSMEngine.__del__
    __enter__ = LGSSMEngine.__enter__
    __exit__ = LGSSMEngine.__exit__
    sync = LGSSMEngine.sync
    free_energy = LGSSMEngine.free_energy
    free_energy_device = LGSSMEngine.free_energy_device
    counters = LGSSMEngine.counters
    set_profiling = LGSSMEngine.set_profiling
    reset_kernel_times = LGSSMEngine.reset_kernel_times
    kernel_times = LGSSMEngine.kernel_times
    stream = LGSSMEngine.stream

    def set_data(self, y):
        y = _c(y).ravel()
        self._chk(_lib.lib().rxhip_set_data(self._h, _lib.VAR_Y, _p(y), y.size, _lib.LAYOUT_TIME_CHAIN))

    def set_data_device(self, ptr, n, keepalive=None):
        self._data_ref = keepalive
        self._chk(_lib.lib().rxhip_set_data_device(self._h, _lib.VAR_Y, ctypes.c_void_p(ptr), n, _lib.LAYOUT_TIME_CHAIN))

    def run(self, iterations=1, free_energy=True):
        self._chk(_lib.lib().rxhip_run(self._h, int(iterations), int(bool(free_energy))))
        self._iters = int(iterations)

    def run_async(self, iterations=1, free_energy=True):
        self._chk(_lib.lib().rxhip_run_async(self._h, int(iterations), int(bool(free_energy))))
        self._iters = int(iterations)

    # split-phase iteration (several GPUs: all-reduce the statistics between accumulate and update)
    def begin_run(self, iterations):
        self._chk(_lib.lib().rxhip_gmm_begin_run(self._h, int(iterations)))
        self._iters = int(iterations)

    def accumulate(self):
        self._chk(_lib.lib().rxhip_gmm_accumulate(self._h))

    def statistics_device(self):
        p, n = ctypes.c_void_p(), ctypes.c_int32()
        self._chk(_lib.lib().rxhip_gmm_statistics_device(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def statistics_tensor(self):
        """torch tensor ALIASING the device statistics buffer (3K'+1 doubles) — what the multi-GPU host hands to
        `torch.distributed.all_reduce` in place (RCCL) between accumulate() and update()."""
        import torch

        ptr, n = self.statistics_device()

        class _View:  # zero-copy: torch adopts foreign device memory through the array interface
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        self._stats_view = _View()  # keep the descriptor object alive as long as the engine
        return torch.as_tensor(self._stats_view, device=f"cuda:{torch.cuda.current_device()}")

    def allreduce_statistics(self, comm):
        """Sum the statistics of this iteration over all ranks of the RCCL communicator (between accumulate and update)."""
        self._chk(_lib.lib().rxhip_gmm_allreduce_statistics(self._h, ctypes.c_void_p(int(getattr(comm, "handle", comm) or 0))))

    allreduce_free_energy = LGSSMEngine.allreduce_free_energy

    def update(self, free_energy=True):
        self._chk(_lib.lib().rxhip_gmm_update(self._h, int(bool(free_energy))))

    def history(self):
        """[iterations][5][K]: mean m, var m, shape p, rate p, alpha s after every iteration (KeepEach)."""
        h = np.empty((self._iters, 5, self.K))
        self._chk(_lib.lib().rxhip_gmm_get_history(self._h, _p(h)))
        return h

    def responsibilities(self):
        r = np.empty((self.N, self.K))
        self._chk(_lib.lib().rxhip_gmm_get_responsibilities(self._h, _p(r)))
        return r


class MvGMMEngine(GMMEngine):
    """Mean-field VMP for the multivariate Gaussian mixture, d = 1…4 (include/rxhip.h rxhip_mvgmm_desc).
    priors: mu0 [K][d], S0 [K][d][d], nu0 [K], V0 [K][d][d], alpha0 [K]; init: the same five blocks of the initial marginals."""

    def __init__(self, N, mu0, S0, nu0, V0, alpha0, init_m_mean, init_m_cov, init_w_nu, init_w_V, init_s_alpha,
                 materialize_responsibilities=False, device=-1, stream=None):
        L = _lib.lib()
        arrs = [_c(a) for a in (mu0, S0, nu0, V0, alpha0, init_m_mean, init_m_cov, init_w_nu, init_w_V, init_s_alpha)]
        self.K, self.d = arrs[0].shape
        K, d = self.K, self.d
        shapes = [(K, d), (K, d, d), (K,), (K, d, d), (K,)] * 2
        if any(a.shape != sh for a, sh in zip(arrs, shapes)):
            raise ValueError("prior / initialisation arrays have inconsistent shapes")
        self.N = int(N)
        self._keep = arrs
        desc = _lib.MvGmmDesc()
        desc.N, desc.K, desc.d = self.N, K, d
        for name, a in zip(("mu0", "S0", "nu0", "V0", "alpha0", "init_m_mean", "init_m_cov", "init_w_nu", "init_w_V", "init_s_alpha"), arrs):
            setattr(desc, name, _p(a))
        desc.materialize_responsibilities = int(bool(materialize_responsibilities))
        desc.device = int(device)
        desc.stream = ctypes.c_void_p(stream) if stream else None
        self._h = ctypes.c_void_p()
        st = L.rxhip_mvgmm_create(ctypes.byref(desc), ctypes.byref(self._h))
        if st != _lib.OK:
            msg = L.rxhip_last_error(self._h).decode() if self._h else L.rxhip_status_string(st).decode()
            if self._h:
                L.rxhip_destroy(self._h)
                self._h = None
            raise RxHipError(st, msg or L.rxhip_status_string(st).decode())
        self._iters = 0
        self._data_ref = None

    def set_data(self, y):
        y = _c(y)
        if y.shape != (self.N, self.d):
            raise ValueError(f"observations must be [N][d] = {(self.N, self.d)}")
        self._chk(_lib.lib().rxhip_set_data(self._h, _lib.VAR_Y, _p(y), y.size, _lib.LAYOUT_TIME_CHAIN))

    def history(self):
        """dict of arrays over (iteration, component): mean [.., d], cov [.., d, d], nu, V [.., d, d], alpha (KeepEach)."""
        d, dd = self.d, self.d * self.d
        h = np.empty((self._iters, self.K, 2 + d + 2 * dd))
        self._chk(_lib.lib().rxhip_gmm_get_history(self._h, _p(h)))
        return dict(mean=h[..., :d], cov=h[..., d:d + dd].reshape(h.shape[:-1] + (d, d)), nu=h[..., d + dd],
                    V=h[..., d + dd + 1:d + 2 * dd + 1].reshape(h.shape[:-1] + (d, d)), alpha=h[..., -1], raw=h)


class HGFEngine:
    """Online hierarchical Gaussian filter (GCV node) for n_series independent series (include/rxhip.h rxhip_hgf_desc)."""

    def __init__(self, T, n_series, kappa, omega, z_variance, y_variance, z0=(0.0, 5.0), x0=(0.0, 5.0), n_gh=31,
                 device=-1, stream=None):
        L = _lib.lib()
        desc = _lib.HgfDesc()
        desc.T, desc.n_series = int(T), int(n_series)
        desc.kappa, desc.omega, desc.z_variance, desc.y_variance = float(kappa), float(omega), float(z_variance), float(y_variance)
        desc.z0_mean, desc.z0_var, desc.x0_mean, desc.x0_var = float(z0[0]), float(z0[1]), float(x0[0]), float(x0[1])
        desc.n_gh, desc.device = int(n_gh), int(device)
        desc.stream = ctypes.c_void_p(stream) if stream else None
        self.T, self.n_series, self.n_chains = int(T), int(n_series), int(n_series)
        self._h = ctypes.c_void_p()
        st = L.rxhip_hgf_create(ctypes.byref(desc), ctypes.byref(self._h))
        if st != _lib.OK:
            msg = L.rxhip_last_error(self._h).decode() if self._h else L.rxhip_status_string(st).decode()
            if self._h:
                L.rxhip_destroy(self._h)
                self._h = None
            raise RxHipError(st, msg or L.rxhip_status_string(st).decode())
        self._iters = 0
        self._data_ref = None

    _chk = LGSSMEngine._chk
    close = LGSSMEngine.close
    __del__ = LGSSMEngine.__del__
    __enter__ = LGSSMEngine.__enter__
    __exit__ = LGSSMEngine.__exit__
    sync = LGSSMEngine.sync
    run = LGSSMEngine.run
    run_async = LGSSMEngine.run_async
    free_energy = LGSSMEngine.free_energy
    free_energy_per_chain = LGSSMEngine.free_energy_per_chain
    free_energy_device = LGSSMEngine.free_energy_device
    copy_free_energy_to_device = LGSSMEngine.copy_free_energy_to_device
    allreduce_free_energy = LGSSMEngine.allreduce_free_energy
    counters = LGSSMEngine.counters
    set_profiling = LGSSMEngine.set_profiling
    reset_kernel_times = LGSSMEngine.reset_kernel_times
    kernel_times = LGSSMEngine.kernel_times
    stream = LGSSMEngine.stream

    def set_data(self, y, layout="time_chain"):
        """y: [T][series] ('time_chain') or [series][T] ('chain_time')."""
        y = _c(y)
        lay = _lib.LAYOUT_TIME_CHAIN if layout == "time_chain" else _lib.LAYOUT_CHAIN_TIME
        self._chk(_lib.lib().rxhip_set_data(self._h, _lib.VAR_Y, _p(y), y.size, lay))

    def set_data_device(self, ptr, n, layout="time_chain", keepalive=None):
        lay = _lib.LAYOUT_TIME_CHAIN if layout == "time_chain" else _lib.LAYOUT_CHAIN_TIME
        self._data_ref = keepalive
        self._chk(_lib.lib().rxhip_set_data_device(self._h, _lib.VAR_Y, ctypes.c_void_p(ptr), n, lay))

    def history(self, layout="time_chain"):
        """(z_mean, z_var, x_mean, x_var), each [T][series] (or [series][T])."""
        lay = _lib.LAYOUT_TIME_CHAIN if layout == "time_chain" else _lib.LAYOUT_CHAIN_TIME
        shp = (self.T, self.n_series) if layout == "time_chain" else (self.n_series, self.T)
        outs = [np.empty(shp) for _ in range(4)]
        self._chk(_lib.lib().rxhip_hgf_get_history(self._h, *[_p(o) for o in outs], lay))
        return tuple(outs)


class DriftChainEngine:
    """Noise-free drift chain `x_prior ~ Normal(m0, v0); x[t] ~ x[t-1] + c; y[t] ~ Normal(x[t], obs_var)`
    (test/models/statespace/ulgssm_tests.jl:8-15) for n_chains independent chains (include/rxhip.h rxhip_drift_chain_desc)."""

    def __init__(self, T, m0, v0, c, obs_var, n_chains=1, prior_through_transition=True, device=-1, stream=None):
        L = _lib.lib()
        desc = _lib.DriftChainDesc()
        desc.T, desc.n_chains = int(T), int(n_chains)
        desc.m0, desc.v0, desc.c, desc.obs_var = float(m0), float(v0), float(c), float(obs_var)
        desc.prior_through_transition = int(bool(prior_through_transition))
        desc.device = int(device)
        desc.stream = ctypes.c_void_p(stream) if stream else None
        self.T, self.n_chains, self.d, self.dy, self.n_models = int(T), int(n_chains), 1, 1, 1
        self._h = ctypes.c_void_p()
        st = L.rxhip_drift_chain_create(ctypes.byref(desc), ctypes.byref(self._h))
        if st != _lib.OK:
            msg = L.rxhip_last_error(self._h).decode() if self._h else L.rxhip_status_string(st).decode()
            if self._h:
                L.rxhip_destroy(self._h)
                self._h = None
            raise RxHipError(st, msg or L.rxhip_status_string(st).decode())
        self._iters = 0
        self._data_ref = None
        self._keep = []

    _chk = LGSSMEngine._chk
    close = LGSSMEngine.close
    __del__ = LGSSMEngine.__del__
    __enter__ = LGSSMEngine.__enter__
    __exit__ = LGSSMEngine.__exit__
    sync = LGSSMEngine.sync
    run = LGSSMEngine.run
    run_async = LGSSMEngine.run_async
    set_data = LGSSMEngine.set_data
    set_data_device = LGSSMEngine.set_data_device
    marginals = LGSSMEngine.marginals
    marginals_of_chains = LGSSMEngine.marginals_of_chains
    free_energy = LGSSMEngine.free_energy
    free_energy_per_chain = LGSSMEngine.free_energy_per_chain
    allreduce_free_energy = LGSSMEngine.allreduce_free_energy
    counters = LGSSMEngine.counters
    set_profiling = LGSSMEngine.set_profiling
    reset_kernel_times = LGSSMEngine.reset_kernel_times
    kernel_times = LGSSMEngine.kernel_times
    stream = LGSSMEngine.stream


class Communicator:
    """RCCL communicator made through the C ABI (rxhip_comm_*): rank 0 calls `Communicator.unique_id()`, the 128 bytes
    travel to the other ranks by any host-side means, every rank constructs `Communicator(nranks, id, rank)`."""

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        st = _lib.lib().rxhip_comm_unique_id(buf)
        if st != _lib.OK:
            raise RxHipError(st, _lib.lib().rxhip_comm_last_error().decode())
        return buf.raw

    def __init__(self, nranks, unique_id, rank, device=-1):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of Communicator.unique_id()")
        self.nranks, self.rank = int(nranks), int(rank)
        h = ctypes.c_void_p()
        st = _lib.lib().rxhip_comm_init_rank(ctypes.byref(h), self.nranks, unique_id, self.rank, int(device))
        if st != _lib.OK:
            raise RxHipError(st, _lib.lib().rxhip_comm_last_error().decode())
        self.handle = h.value

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().rxhip_comm_destroy(ctypes.c_void_p(self.handle))
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
