"""Thin object layer over the C ABI: one Engine = one omc_ctx = the chains held by one GPU.

PyTorch is used only for device memory and the stream; every numerical call goes to
libomcmc_hip.so.  Tensors are float64 ROCm tensors; per-chain vectors are (C, n) row-major
(chain-major, include/omcmc_hip.h), per-chain scalars are (C,).
"""

import ctypes as C

import numpy as np

from openmcmc_amd import _abi
from openmcmc_amd._abi import check, lib


_TORCH = None


def _torch():
    global _TORCH
    if _TORCH is None:
        import torch

        _TORCH = torch
    return _TORCH


class Engine:
    """Owns an omc_ctx bound to torch's current stream on `device`."""

    def __init__(self, n_chains, seed=0, device=0, chain_id_offset=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("openmcmc_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.n_chains = int(n_chains)
        self.seed = int(seed)
        self.chain_id_offset = int(chain_id_offset)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.device)
        self._stream = torch.cuda.current_stream(self.device)
        ctx = C.c_void_p()
        check(lib.omc_ctx_create(self.device_index, self.n_chains, self.seed, self.chain_id_offset,
                                 C.c_void_p(self._stream.cuda_stream), 0, C.byref(ctx)))
        self._ctx = ctx
        self._keep = []
        # library writes into caller tensors that torch's version counter does not see (see note_write)
        self._write_serial = 0
        self._written = {}

    # ------------------------------------------------------------------ memory helpers
    def empty(self, *shape):
        return _torch().empty(*shape, dtype=_torch().float64, device=self.device)

    def zeros(self, *shape):
        return _torch().zeros(*shape, dtype=_torch().float64, device=self.device)

    def full(self, shape, value):
        return _torch().full(shape, float(value), dtype=_torch().float64, device=self.device)

    def to_device(self, array):
        torch = _torch()
        if isinstance(array, torch.Tensor):
            return array.to(device=self.device, dtype=torch.float64).contiguous()
        return torch.as_tensor(np.ascontiguousarray(array, dtype=np.float64), device=self.device)

    def _p(self, t, rows=None, min_cols=None):
        """Device pointer of a float64 tensor (None passes through as NULL)."""
        if t is None:
            return None
        # hot: ~1000 calls per sweep of a reversible-jump model.  A plain int is a valid c_void_p argument.
        try:
            if t.dtype is not _TORCH.float64 or not t.is_cuda:
                raise TypeError("expected a float64 ROCm tensor")
        except AttributeError:
            raise TypeError("expected a float64 ROCm tensor") from None
        shape = t.shape
        if len(shape) >= 1 and shape[-1] != 1 and t.stride(-1) != 1:
            raise ValueError("last dimension must be contiguous")
        if rows is not None and (len(shape) != 2 or shape[0] != rows or shape[1] < min_cols):
            raise ValueError(f"expected shape ({rows}, >={min_cols}), got {tuple(shape)}")
        return t.data_ptr()

    def _vec(self, t, n):
        if t is None:
            return None
        if t.numel() < n:
            raise ValueError(f"shared vector shorter than {n}")
        return self._p(t)

    def _chain_scalar(self, t):
        if t is None:
            return None
        if t.numel() != self.n_chains:
            raise ValueError(f"per-chain scalar must have {self.n_chains} entries, got {t.numel()}")
        return self._p(t)

    # ------------------------------------------------------------------ who wrote what
    def note_write(self, *tensors):
        """Record that the library wrote into these tensors.  torch's version counter sees every write but the library's
        own; a sampler that caches something derived from a state tensor (the whitened state of the fused
        Metropolis-Hastings steps) asks `written_since` before it trusts the cache."""
        self._write_serial += 1
        for t in tensors:
            if t is not None:
                self._written[t.untyped_storage().data_ptr()] = self._write_serial
        return self._write_serial

    def written_since(self, t, serial):
        """Has the library written into t's storage after write number `serial` (a value note_write returned)?"""
        return self._written.get(t.untyped_storage().data_ptr(), 0) > serial

    # ------------------------------------------------------------------ quadratic forms a draw left behind
    def quad_cache_put(self, dist, quad, inputs):
        """The fused quadratic form of a draw IS the residual statistic r'Mr of `dist` for the state the draw leaves
        (sampler.py:276,284; gmrf.py:343-344).  Kept with what it was computed from -- the device tensors of the state
        entries the residual reads -- and handed out by `quad_cache_get` while none of them has been replaced or written."""
        if not hasattr(self, "_quad_cache"):
            self._quad_cache = {}
        self._quad_cache[id(dist)] = (dist, quad, [(t, t.data_ptr(), t._version) for t in inputs], self._write_serial)

    def quad_cache_get(self, dist, inputs):
        hit = getattr(self, "_quad_cache", {}).get(id(dist))
        if hit is None or hit[0] is not dist or len(hit[2]) != len(inputs):
            return None
        for (t0, ptr, ver), t in zip(hit[2], inputs):
            if t.data_ptr() != ptr or t.shape != t0.shape or t._version != ver or self.written_since(t, hit[3]):
                return None
        return hit[1]

    # ------------------------------------------------------------------ context
    def close(self):
        if self._ctx is not None:
            lib.omc_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(lib.omc_ctx_synchronize(self._ctx))

    def check_status(self):
        """Synchronise and raise numpy.linalg.LinAlgError if any chain's precision was not
        positive definite (the reference raises from np.linalg.cholesky, gmrf.py:518)."""
        bad = C.c_int64(-1)
        st = lib.omc_ctx_status(self._ctx, C.byref(bad))
        if st == _abi.NOT_POSDEF:
            raise np.linalg.LinAlgError(f"Matrix is not positive definite (chain {bad.value})")
        check(st)

    def counter(self, name):
        """Diagnostic counter of the context (omc_ctx_counter), e.g. "tridiag_join_fallbacks"."""
        v = C.c_int64(0)
        check(lib.omc_ctx_counter(self._ctx, name.encode(), C.byref(v)))
        return int(v.value)

    def set_option(self, name, value):
        check(lib.omc_ctx_set_option(self._ctx, name.encode(), int(value)))

    # ------------------------------------------------------------------ diagnostics of omc_gmrf_run
    def sweep_clock(self, capacity):
        """Switch on the sweep clock: a device ring [capacity][C][2] of int64 into which the workgroup of every (sweep, chain)
        of omc_gmrf_run leaves the device's constant-rate counter at its entry and exit.  Returns the ring (a tensor the
        caller keeps); capacity 0 switches it off."""
        if capacity <= 0:
            self.set_option("sweep_times_cap", 0)
            self._sweep_ring = None
            return None
        torch = _torch()
        ring = torch.zeros((int(capacity), self.n_chains, 2), dtype=torch.int64, device=self.device)
        self.set_option("sweep_times_cap", int(capacity))
        self.set_option("sweep_times_ptr", ring.data_ptr())
        self._sweep_ring = ring
        return ring

    def launch_log(self):
        """Launches of the last gmrf_run call: (total, [dict(t_begin, t_end, n_sweeps, form, ring_pos)]) with the host
        times on time.perf_counter's clock (CLOCK_MONOTONIC)."""
        cap = 64
        buf = (C.c_double * (5 * cap))()
        n = C.c_int64(0)
        check(lib.omc_ctx_launch_log(self._ctx, buf, cap, C.byref(n)))
        recs = [dict(t_begin=buf[5 * i], t_end=buf[5 * i + 1], n_sweeps=int(buf[5 * i + 2]), form=int(buf[5 * i + 3]),
                     ring_pos=int(buf[5 * i + 4])) for i in range(min(cap, n.value))]
        return int(n.value), recs

    # ------------------------------------------------------------------ tridiagonal GMRF
    def tridiag_terms(self, terms, n):
        """terms: list of dicts with optional keys diag, off, rhs, center (shared, length n / n-1)
        and scale (per-chain, length C).  Returns the ctypes struct (keeps the tensors alive)."""
        if not 1 <= len(terms) <= _abi.OMC_MAX_TERMS:
            raise ValueError("1..4 terms supported")
        T = _abi.TridiagTerms()
        T.n_terms = len(terms)
        keep = []
        for k, t in enumerate(terms):
            T.diag[k] = self._vec(t.get("diag"), n)
            T.off[k] = self._vec(t.get("off"), n - 1) if n > 1 else None
            T.rhs[k] = self._vec(t.get("rhs"), n)
            T.center[k] = self._vec(t.get("center"), n)
            T.scale[k] = self._chain_scalar(t.get("scale"))
            keep.append(dict(t))
        T._keep = keep
        self.set_center_chain(T, [t.get("center_chain") for t in terms], n)
        return T

    def set_center_chain(self, T, vectors, n):
        """Per-chain part of the terms' centres (omc_tridiag_terms.center_chain): one (C, n) tensor or None per term, all with the
        same row stride.  Patched into an existing struct: a sampled prior mean is a new tensor every sweep."""
        ld = 0
        held = []
        for k in range(T.n_terms):
            v = vectors[k] if k < len(vectors) else None
            if v is None:
                T.center_chain[k] = None
                continue
            if v.dim() != 2 or v.shape[0] != self.n_chains or v.shape[1] != n or v.stride(1) != 1 or (ld and v.stride(0) != ld):
                raise ValueError("center_chain: (C, n) tensors of one row stride")
            ld = v.stride(0)
            T.center_chain[k] = v.data_ptr()
            held.append(v)
        T.ld_center_chain = ld
        T._keep_cc = held

    def tridiag_takes_center_chain(self, n):
        return bool(lib.omc_tridiag_takes_center_chain(self._ctx, int(n)))

    def tridiag_sample_canonical(self, n, terms, x_out, z=None, rhs_chain=None, draw_index=0,
                                 mean_out=None, quad_out=None, logdet_out=None):
        T = terms if isinstance(terms, _abi.TridiagTerms) else self.tridiag_terms(terms, n)
        Cn = self.n_chains
        ld = lambda t: 0 if t is None else t.stride(0)  # noqa: E731
        if quad_out is not None and quad_out.numel() < T.n_terms * Cn:
            raise ValueError("quad_out too small")
        check(lib.omc_tridiag_sample_canonical(
            self._ctx, n, C.byref(T),
            self._p(rhs_chain, Cn, n), ld(rhs_chain), self._p(z, Cn, n), ld(z), int(draw_index),
            self._p(x_out, Cn, n), ld(x_out), self._p(mean_out, Cn, n), ld(mean_out),
            self._p(quad_out), self._chain_scalar(logdet_out)))

    def gamma_blocks(self, blocks, n_terms):
        """blocks: list (one per term) of None or dict(a0, b0, n_pos, g=None, store=None, logdet=None)."""
        arr = (_abi.GammaBlock * _abi.OMC_MAX_TERMS)()
        keep = []
        for k in range(n_terms):
            b = blocks[k] if k < len(blocks) else None
            if b is None:
                continue
            arr[k].enabled = int(b.get("enabled", True))
            arr[k].a0, arr[k].b0, arr[k].n_pos = float(b.get("a0", 0.0)), float(b.get("b0", 0.0)), int(b.get("n_pos", 0))
            arr[k].g_inject = self._chain_scalar(b.get("g"))
            arr[k].draw_index = int(b.get("draw_index", 0))
            arr[k].store = self._chain_scalar(b.get("store"))
            arr[k].logdet_unscaled = self._p(b.get("logdet"))
            keep.append(dict(b))
        arr._keep = keep
        return arr

    def gmrf_sweep(self, n, terms, blocks, x_out, z=None, rhs_chain=None, draw_index=0, log_post_out=None,
                   gamma_draw_base=None):
        """One fused sweep [NormalNormal(x), NormalGamma(scale_k)..., log_post]; the precision scalars
        of enabled blocks (terms[k]["scale"]) are updated in place."""
        T = terms if isinstance(terms, _abi.TridiagTerms) else self.tridiag_terms(terms, n)
        B = blocks if not isinstance(blocks, (list, tuple)) else self.gamma_blocks(blocks, T.n_terms)
        if gamma_draw_base is not None:  # block k draws from stream gamma_draw_base + k
            for k in range(T.n_terms):
                B[k].draw_index = int(gamma_draw_base) + k
        Cn = self.n_chains
        ld = lambda t: 0 if t is None else t.stride(0)  # noqa: E731
        check(lib.omc_gmrf_sweep(self._ctx, n, C.byref(T), B, self._p(rhs_chain, Cn, n), ld(rhs_chain),
                                 self._p(z, Cn, n), ld(z), int(draw_index), self._p(x_out, Cn, n), ld(x_out),
                                 self._chain_scalar(log_post_out)))

    def gmrf_run(self, n, terms, blocks, n_burn, n_iter, n_thin, x_store, scratch_x, draw_index0=0,
                 draws_per_sweep=1, first_slot=0, log_post_store=None):
        """The whole run_mcmc loop for [NormalNormal, NormalGamma...] issued from C (omc_gmrf_run).
        x_store: (n_slots, C, n); blocks[k]["store"]: (n_slots, C) or None; blocks[k]["draw_index"] is the
        block's offset inside a sweep's draw indices."""
        T = terms if isinstance(terms, _abi.TridiagTerms) else self.tridiag_terms(terms, n)
        B = blocks if not isinstance(blocks, (list, tuple)) else self._gamma_blocks_strided(blocks, T.n_terms)
        n_slots = x_store.shape[0]
        if x_store.dim() != 3 or x_store.shape[1] != self.n_chains or x_store.shape[2] < n or not x_store.is_contiguous():
            raise ValueError("x_store must be a contiguous (n_slots, C, >=n) tensor")
        check(lib.omc_gmrf_run(self._ctx, n, C.byref(T), B, int(n_burn), int(n_iter), int(n_thin), int(draw_index0),
                               int(draws_per_sweep), self._p(x_store), x_store.stride(1), x_store.stride(0),
                               int(first_slot), int(n_slots), self._p(log_post_store),
                               self._p(scratch_x, self.n_chains, n)))

    def _gamma_blocks_strided(self, blocks, n_terms):
        arr = (_abi.GammaBlock * _abi.OMC_MAX_TERMS)()
        keep = []
        for k in range(n_terms):
            b = blocks[k] if k < len(blocks) else None
            if b is None:
                continue
            arr[k].enabled = int(b.get("enabled", True))
            arr[k].a0, arr[k].b0, arr[k].n_pos = float(b.get("a0", 0.0)), float(b.get("b0", 0.0)), int(b.get("n_pos", 0))
            arr[k].draw_index = int(b.get("draw_index", 0))
            arr[k].store = self._p(b.get("store"))
            arr[k].logdet_unscaled = self._p(b.get("logdet"))
            keep.append(dict(b))
        arr._keep = keep
        return arr

    def tridiag_quadform(self, n, terms, x, quad_out):
        T = terms if isinstance(terms, _abi.TridiagTerms) else self.tridiag_terms(terms, n)
        check(lib.omc_tridiag_quadform(self._ctx, n, C.byref(T), self._p(x, self.n_chains, n), x.stride(0),
                                       self._p(quad_out)))

    def tridiag_matvec_chain(self, n, diag, off, v, scale=None, out=None, accumulate=False):
        """out[c] (+)= scale[c] * M v_c for a per-chain (C, n) vector (omc_tridiag_matvec_chain)."""
        out = self.empty(self.n_chains, n) if out is None else out
        check(lib.omc_tridiag_matvec_chain(self._ctx, n, self._vec(diag, n), self._vec(off, n - 1) if (off is not None and n > 1) else None,
                                           self._p(v, self.n_chains, n), v.stride(0), self._chain_scalar(scale),
                                           self._p(out, self.n_chains, n), out.stride(0), int(accumulate)))
        return out

    def chain_lincomb(self, a, x, b, y, out=None):
        """a x_c + b y_c; y is (C, n) or a shared (n,) vector (omc_chain_lincomb)."""
        n = x.shape[1]
        out = self.empty(self.n_chains, n) if out is None else out
        shared = y.dim() == 1
        check(lib.omc_chain_lincomb(self._ctx, n, float(a), self._p(x, self.n_chains, n), x.stride(0), float(b),
                                    self._vec(y, n) if shared else self._p(y, self.n_chains, n), 0 if shared else y.stride(0),
                                    self._p(out, self.n_chains, n), out.stride(0)))
        return out

    def chain_copy(self, src, dst):
        """dst[c] = src[c] for (C, n) tensors with unit inner stride (omc_chain_copy: into a store slab)."""
        n = src.shape[1]
        check(lib.omc_chain_copy(self._ctx, n, self._p(src, self.n_chains, n), src.stride(0), self._p(dst, self.n_chains, n), dst.stride(0)))
        return dst

    def tridiag_matvec(self, n, diag, off, v):
        out = self.empty(n)
        check(lib.omc_tridiag_matvec(self._ctx, n, self._vec(diag, n), self._vec(off, n - 1) if n > 1 else None,
                                     self._vec(v, n), self._p(out)))
        return out

    def tridiag_logdet(self, n, diag, off):
        out = self.empty(1)
        check(lib.omc_tridiag_logdet(self._ctx, n, self._vec(diag, n), self._vec(off, n - 1) if n > 1 else None,
                                     self._p(out)))
        return out

    # ------------------------------------------------------------------ dense Normal-Normal
    def dense_terms(self, terms, p):
        """terms: list of dicts with optional keys mat ((p,p) shared symmetric; None = identity),
        rhs ((p,) shared), scale ((C,) per chain)."""
        if not 1 <= len(terms) <= _abi.OMC_MAX_TERMS:
            raise ValueError("1..4 terms supported")
        T = _abi.DenseTerms()
        T.n_terms = len(terms)
        keep = []
        for k, t in enumerate(terms):
            m = t.get("mat")
            if m is not None and (m.dim() != 2 or m.shape[0] != p or m.shape[1] != p or not m.is_contiguous()):
                raise ValueError("mat must be a contiguous (p, p) tensor")
            T.mat[k] = self._p(m)
            T.rhs[k] = self._vec(t.get("rhs"), p)
            T.scale[k] = self._chain_scalar(t.get("scale"))
            keep.append(dict(t))
        T.diag_chain = None
        T._keep = keep
        return T

    def dense_sample_canonical(self, p, terms, x_out, z=None, rhs_chain=None, draw_index=0, mean_out=None,
                               logdet_out=None, diag_chain=None):
        """diag_chain: optional (C, p) per-chain diagonal added to Q_c (mixture prior precision)."""
        T = terms if isinstance(terms, _abi.DenseTerms) else self.dense_terms(terms, p)
        Cn = self.n_chains
        T.diag_chain = self._p(diag_chain, Cn, p)
        ld = lambda t: 0 if t is None else t.stride(0)  # noqa: E731
        check(lib.omc_dense_sample_canonical(
            self._ctx, p, C.byref(T), self._p(rhs_chain, Cn, p), ld(rhs_chain), self._p(z, Cn, p), ld(z),
            int(draw_index), self._p(x_out, Cn, p), ld(x_out), self._p(mean_out, Cn, p), ld(mean_out),
            self._chain_scalar(logdet_out)))

    def dense_spectral_prepare(self, M):
        """(V, ev) with M = V diag(ev) V' (omc_dense_spectral_prepare): once per model, for the spectral route."""
        p = M.shape[0]
        V, ev = self.empty(p, p), self.empty(p)
        check(lib.omc_dense_spectral_prepare(self._ctx, p, self._p(M), self._p(V), self._p(ev)))
        return V, ev

    def dense_spectral_sample(self, p, terms, k_mat, V, ev, x_out, z=None, rhs_chain=None, draw_index=0, mean_out=None,
                              logdet_out=None):
        """The conditional draw for Q_c = a_c I + b_c M in M's eigenbasis (omc_dense_spectral_sample)."""
        T = terms if isinstance(terms, _abi.DenseTerms) else self.dense_terms(terms, p)
        Cn = self.n_chains
        T.diag_chain = None
        ld = lambda t: 0 if t is None else t.stride(0)  # noqa: E731
        check(lib.omc_dense_spectral_sample(
            self._ctx, p, C.byref(T), int(k_mat), self._p(V), self._p(ev), self._p(rhs_chain, Cn, p), ld(rhs_chain),
            self._p(z, Cn, p), ld(z), int(draw_index), self._p(x_out, Cn, p), ld(x_out), self._p(mean_out, Cn, p),
            ld(mean_out), self._chain_scalar(logdet_out)))

    def gram(self, X, w=None):
        """G = X' diag(w) X for a shared (n, p) design matrix."""
        n, p = X.shape
        G = self.empty(p, p)
        check(lib.omc_gram(self._ctx, n, p, self._p(X), self._vec(w, n), self._p(G)))
        return G

    def design_rhs(self, X, y, w=None):
        """X' diag(w) y."""
        n, p = X.shape
        out = self.empty(p)
        check(lib.omc_design_rhs(self._ctx, n, p, self._p(X), self._vec(w, n), self._vec(y, n), self._p(out)))
        return out

    def design_predict(self, X, beta, fitted=None):
        """fitted[c] = X beta_c for all chains (one GEMM)."""
        n, p = X.shape
        fitted = self.empty(self.n_chains, n) if fitted is None else fitted
        check(lib.omc_design_predict(self._ctx, n, p, self._p(X), self._p(beta, self.n_chains, p), beta.stride(0),
                                     self._p(fitted, self.n_chains, n), fitted.stride(0)))
        return fitted

    def weighted_resid_sq(self, y, fitted, out, w=None):
        n = y.numel()
        check(lib.omc_weighted_resid_sq(self._ctx, n, self._vec(y, n), self._p(fitted, self.n_chains, n),
                                        fitted.stride(0), self._vec(w, n), self._chain_scalar(out)))

    # ------------------------------------------------------------------ Metropolis-Hastings steps
    def _ip(self, t):
        """Device pointer of an int64 (C,) counter tensor."""
        if t is None:
            return None
        torch = _torch()
        if t.dtype != torch.int64 or not t.is_cuda or t.numel() != self.n_chains:
            raise TypeError("expected an int64 ROCm tensor with one entry per chain")
        return C.c_void_p(t.data_ptr())

    def dense_cholesky(self, A, scale=1.0):
        """(L, sum log L_ii) with L = chol(scale * A) lower, shared by all chains."""
        d = A.shape[0]
        L, sl = self.empty(d, d), self.empty(1)
        check(lib.omc_dense_cholesky(self._ctx, d, self._p(A), float(scale), self._p(L), self._p(sl)))
        return L, sl

    def mh_invalidate(self):
        """Drop what the library derived from the last (Q, L, step) of the fused Metropolis-Hastings routes."""
        check(lib.omc_mh_invalidate(self._ctx))

    def mala_step(self, Q, mu, L, sumlogL, step, x, z=None, u=None, draw_index=0, accept_count=None,
                  proposal_count=None):
        d = Q.shape[0]
        check(lib.omc_mala_step(self._ctx, d, self._p(Q), self._vec(mu, d), self._p(L), self._p(sumlogL), float(step),
                                self._p(z, self.n_chains, d), 0 if z is None else z.stride(0), self._chain_scalar(u),
                                int(draw_index), self._p(x, self.n_chains, d), x.stride(0), self._ip(accept_count),
                                self._ip(proposal_count)))

    def mala_step_white(self, mu, L, sumlogL, step, x, state_is_current=False, z=None, u=None, draw_index=0, accept_count=None,
                        proposal_count=None, log_p_out=None):
        """omc_mala_step_white: the ManifoldMALA step for L = chol(Q / step^2) in whitened coordinates; log_p_out (C,) takes
        the target's log density at the state the step leaves behind."""
        d = L.shape[0]
        check(lib.omc_mala_step_white(self._ctx, d, self._vec(mu, d), self._p(L), self._p(sumlogL), float(step),
                                      self._p(z, self.n_chains, d), 0 if z is None else z.stride(0), self._chain_scalar(u),
                                      int(draw_index), self._p(x, self.n_chains, d), x.stride(0), int(bool(state_is_current)),
                                      self._ip(accept_count), self._ip(proposal_count), self._chain_scalar(log_p_out)))

    def mala_run_white(self, mu, L, sumlogL, step, x, n_steps, state_is_current=False, z=None, u=None, draw_index0=0, draw_stride=1,
                       x_store=None, logp_store=None, accept_count=None, proposal_count=None, log_p_out=None):
        """omc_mala_run_white: n_steps whitened ManifoldMALA steps, one launch per block of 32 steps and one product per block
        into x_store (n_steps, C, d) / logp_store (n_steps, C); z (n_steps, C, d) and u (n_steps, C) inject the draws."""
        d = L.shape[0]
        n_steps = int(n_steps)
        if z is not None and (z.dim() != 3 or z.shape[0] < n_steps or z.shape[1] != self.n_chains or not z.is_contiguous()):
            raise ValueError("z must be a contiguous (n_steps, C, d) tensor")
        if u is not None and (u.dim() != 2 or u.shape[0] < n_steps or u.shape[1] != self.n_chains or not u.is_contiguous()):
            raise ValueError("u must be a contiguous (n_steps, C) tensor")
        if x_store is not None and (tuple(x_store.shape) != (n_steps, self.n_chains, d) or not x_store.is_contiguous()):
            raise ValueError("x_store must be a contiguous (n_steps, C, d) tensor")
        if logp_store is not None and (tuple(logp_store.shape) != (n_steps, self.n_chains) or not logp_store.is_contiguous()):
            raise ValueError("logp_store must be a contiguous (n_steps, C) tensor")
        check(lib.omc_mala_run_white(self._ctx, d, self._vec(mu, d), self._p(L), self._p(sumlogL), float(step), self._p(z),
                                     0 if z is None else z.stride(1), self._p(u), int(draw_index0), int(draw_stride), n_steps,
                                     self._p(x, self.n_chains, d), x.stride(0), int(bool(state_is_current)), self._p(x_store),
                                     self._p(logp_store), self._ip(accept_count), self._ip(proposal_count),
                                     self._chain_scalar(log_p_out)))

    def rw_step(self, mu, LQ, sumlogLQ, step, x, z=None, u=None, draw_index=0, accept_count=None,
                proposal_count=None):
        d = LQ.shape[0]
        check(lib.omc_rw_step(self._ctx, d, self._vec(mu, d), self._p(LQ), self._p(sumlogLQ), float(step),
                              self._p(z, self.n_chains, d), 0 if z is None else z.stride(0), self._chain_scalar(u),
                              int(draw_index), self._p(x, self.n_chains, d), x.stride(0), self._ip(accept_count),
                              self._ip(proposal_count)))

    def rw_step_white(self, mu, LQ, sumlogLQ, step, x, state_is_current=False, z=None, u=None, draw_index=0,
                      accept_count=None, proposal_count=None, log_p_out=None):
        """omc_rw_step_white: the fused random-walk step with L_Q'(x - mu) carried from step to step."""
        d = LQ.shape[0]
        check(lib.omc_rw_step_white(self._ctx, d, self._vec(mu, d), self._p(LQ), self._p(sumlogLQ), float(step),
                                    self._p(z, self.n_chains, d), 0 if z is None else z.stride(0), self._chain_scalar(u),
                                    int(draw_index), self._p(x, self.n_chains, d), x.stride(0), int(bool(state_is_current)),
                                    self._ip(accept_count), self._ip(proposal_count), self._chain_scalar(log_p_out)))

    # ------------------------------------------------------------------ per-model constants
    def matrix_logdet(self, st):
        """Device scalar log det M of a Normal's unscaled precision (tridiagonal bands), cached."""
        if not hasattr(self, "_logdet_cache"):
            self._logdet_cache = {}
        hit = self._logdet_cache.get(id(st.matrix))
        if hit is not None and hit[0] is st.matrix:
            return hit[1]
        if st.diag is False and st.band is None:
            # dense: one Cholesky of M itself (gmrf.py:339: 2 sum log L_ii)
            _, sumlog = self.dense_cholesky(self.shared(st.matrix), 1.0)
            ld = sumlog * 2.0
        elif st.diag is False:
            # banded: one factorisation of M itself on the device (every chain computes the same number)
            x, out = self.empty(self.n_chains, st.n), self.empty(self.n_chains)
            self.band_sample_canonical(st.n, [{"band": self.to_device(st.band)}], x, logdet_out=out)
            ld = out[:1].clone()
        elif st.diag is None and st.off is None:
            ld = self.zeros(1)
        else:
            diag = self.to_device(st.diag) if st.diag is not None else self.full((st.n,), 1.0)
            ld = self.tridiag_logdet(st.n, diag, None if st.off is None else self.to_device(st.off))
        self._logdet_cache[id(st.matrix)] = (st.matrix, ld)
        return ld

    def shared(self, array):
        """Device copy of a shared host array (dense or scipy.sparse -> dense), cached by identity."""
        from scipy import sparse

        if not hasattr(self, "_shared_cache"):
            self._shared_cache = {}
        hit = self._shared_cache.get(id(array))
        if hit is not None and hit[0] is array:
            return hit[1]
        dense = array.toarray() if sparse.issparse(array) else np.asarray(array, dtype=np.float64)
        t = self.to_device(np.ascontiguousarray(dense, dtype=np.float64))
        self._shared_cache[id(array)] = (array, t)
        return t

    def band_cache(self, dist, st, center):
        """Device copies of a banded Normal's shared pieces: band rows of M, the vector m its residual is taken around
        and M m (host product, once per model)."""
        if not hasattr(self, "_band_cache"):
            self._band_cache = {}
        key = (id(dist), id(st.matrix))
        c_host = np.ascontiguousarray(center, dtype=np.float64).reshape(-1)
        hit = self._band_cache.get(key)
        if hit is not None and np.array_equal(hit["center_host"], c_host):
            return hit
        rows = st.band_rows()
        entry = {"band": None if rows is None else self.to_device(rows), "center_host": c_host,
                 "center": self.to_device(c_host) if c_host.any() else None,
                 "rhs": self.to_device(np.asarray(st.matrix @ c_host).reshape(-1)) if c_host.any() else None}
        self._band_cache[key] = entry
        return entry

    def model_cache(self, dist, state, st, center):
        """Device copies of one Normal's shared pieces (bands of M, the vector m its residual is taken
        around, M m, log det M), built once per (distribution, matrix, vector) and reused every sweep."""
        if not hasattr(self, "_model_cache"):
            self._model_cache = {}
        key = (id(dist), id(st.matrix))
        hit = self._model_cache.get(key)
        c_host = np.ascontiguousarray(center, dtype=np.float64).reshape(-1)
        # ids can be recycled after garbage collection: the entry keeps the objects and is checked by identity
        if hit is not None and hit["dist"] is dist and hit["matrix"] is st.matrix and np.array_equal(hit["center_host"], c_host):
            return hit
        n = st.n
        diag = None if st.diag is None else self.to_device(st.diag)
        off = None if st.off is None else self.to_device(st.off)
        cvec = self.to_device(c_host) if c_host.any() else None
        if cvec is None:
            rhs = None
        elif diag is None and off is None:
            rhs = cvec
        else:
            rhs = self.tridiag_matvec(n, diag, off, cvec)
        logdet = self.zeros(1) if (diag is None and off is None) else self.tridiag_logdet(n, diag, off)
        entry = {"dist": dist, "matrix": st.matrix, "diag": diag, "off": off, "center": cvec, "rhs": rhs, "logdet": logdet,
                 "center_host": c_host,
                 "terms_unit": self.tridiag_terms([{"diag": diag, "off": off, "center": cvec}], n)}
        self._model_cache[key] = entry
        return entry

    # ------------------------------------------------------------------ scalars
    def normal_gamma_update(self, a0, b0, n_pos, quad, out, g=None, draw_index=0):
        check(lib.omc_normal_gamma_update(self._ctx, float(a0), float(b0), int(n_pos), self._chain_scalar(quad),
                                          self._chain_scalar(g), int(draw_index), self._chain_scalar(out)))

    def scaled_gauss_logpdf(self, n, scale, logdet_unscaled, quad, out, accumulate=False):
        check(lib.omc_scaled_gauss_logpdf(self._ctx, n, self._chain_scalar(scale), self._p(logdet_unscaled),
                                          self._chain_scalar(quad), self._chain_scalar(out), int(accumulate)))

    def log_post_sum(self, pieces, host_const, out):
        """omc_log_post_sum.  pieces: ("gauss", n, scale | None, logdet, mult, quad) or ("gamma", x, shape, rate), in order."""
        arr = (_abi.LogpPiece * len(pieces))()
        for i, pc in enumerate(pieces):
            if pc[0] == "gauss":
                arr[i].kind, arr[i].n = 0, float(pc[1])
                arr[i].scale, arr[i].logdet, arr[i].logdet_mult = self._chain_scalar(pc[2]), self._p(pc[3]), float(pc[4])
                arr[i].quad = self._chain_scalar(pc[5])
            else:
                arr[i].kind, arr[i].x, arr[i].shape, arr[i].rate = 1, self._chain_scalar(pc[1]), float(pc[2]), float(pc[3])
        check(lib.omc_log_post_sum(self._ctx, len(pieces), arr, float(host_const), self._chain_scalar(out)))

    def gamma_logpdf(self, x, shape, rate, out, accumulate=False):
        check(lib.omc_gamma_logpdf(self._ctx, self._chain_scalar(x), float(shape), float(rate),
                                   self._chain_scalar(out), int(accumulate)))

    # ------------------------------------------------------------------ reversible-jump bookkeeping
    def rj_move(self, n, n_max, birth_probability, u=None, idx=None, draw_index=0):
        """(birth int32 (C,), p_birth, p_death, del_index int64) for every chain; n: int64 (C,) tensor."""
        torch = _torch()
        Cn = self.n_chains
        if n.dtype != torch.int64 or n.numel() != Cn or not n.is_cuda:
            raise TypeError("n must be an int64 ROCm tensor with one entry per chain")
        birth = torch.empty(Cn, dtype=torch.int32, device=self.device)
        dele = torch.empty(Cn, dtype=torch.int64, device=self.device)
        pb, pd = self.empty(Cn), self.empty(Cn)
        check(lib.omc_rj_move(self._ctx, int(n_max), float(birth_probability), C.c_void_p(n.data_ptr()),
                              self._chain_scalar(u), None if idx is None else C.c_void_p(idx.data_ptr()),
                              int(draw_index), C.c_void_p(birth.data_ptr()), self._p(pb), self._p(pd),
                              C.c_void_p(dele.data_ptr())))
        return birth, pb, pd, dele

    def rj_move_densities(self, count, n_max, birth_probability, density_chain=None, density_const=0.0, u=None, idx=None,
                          draw_index=0):
        """(birth int32 (C,), del_index int64, count_prop, lq_fwd, lq_rev) for every chain from the float64 counts the
        state holds: the move of rj_move with the proposed count and the two proposal log-densities
        (reversible_jump.py:142-144, 189-191) made in the same launch."""
        torch = _torch()
        Cn = self.n_chains
        birth = torch.empty(Cn, dtype=torch.int32, device=self.device)
        dele = torch.empty(Cn, dtype=torch.int64, device=self.device)
        cp, lf, lr = self.empty(Cn), self.empty(Cn), self.empty(Cn)
        check(lib.omc_rj_move_densities(self._ctx, int(n_max), float(birth_probability), self._chain_scalar(count),
                                        self._chain_scalar(u), None if idx is None else C.c_void_p(idx.data_ptr()),
                                        int(draw_index), self._chain_scalar(density_chain), float(density_const),
                                        C.c_void_p(birth.data_ptr()), C.c_void_p(dele.data_ptr()), self._p(cp), self._p(lf),
                                        self._p(lr)))
        return birth, dele, cp, lf, lr

    def store_ragged(self, src, count, dst):
        """dst[c, j] = src[c, j] for j < count[c], NaN beyond (src (C, width) rows contiguous, dst (C, >= width))."""
        Cn, width = src.shape
        if src.stride(1) != 1 or dst.stride(1) != 1 or dst.shape[0] != Cn or dst.shape[1] < width:
            raise ValueError("store_ragged wants row-contiguous (C, width) source and (C, >= width) destination")
        check(lib.omc_store_ragged(self._ctx, width, self._p(src), src.stride(0), self._chain_scalar(count), self._p(dst),
                                   dst.stride(0)))

    # ------------------------------------------------------------------ banded precisions
    def band_terms(self, terms, n):
        """terms: list of dicts with optional keys band ((bw+1, n) shared tensor, sub-diagonal d in row d; None =
        identity), rhs ((n,) shared), scale ((C,) per chain).  Returns (struct, overall bandwidth)."""
        if not 1 <= len(terms) <= _abi.OMC_MAX_TERMS:
            raise ValueError("1..4 terms supported")
        T = _abi.BandTerms()
        T.n_terms = len(terms)
        keep, w = [], 0
        for k, t in enumerate(terms):
            band = t.get("band")
            if band is not None:
                if band.dim() != 2 or band.shape[1] != n or not band.is_contiguous():
                    raise ValueError("band must be a contiguous (bw+1, n) tensor")
                T.bw[k] = band.shape[0] - 1
                w = max(w, band.shape[0] - 1)
            T.band[k] = self._p(band)
            T.rhs[k] = self._vec(t.get("rhs"), n)
            T.scale[k] = self._chain_scalar(t.get("scale"))
            keep.append(dict(t))
        T._keep = keep
        return T, w

    def band_sample_canonical(self, n, terms, x_out, z=None, rhs_chain=None, draw_index=0, mean_out=None, logdet_out=None):
        T, w = terms if isinstance(terms, tuple) else self.band_terms(terms, n)
        Cn = self.n_chains
        ld = lambda t: 0 if t is None else t.stride(0)  # noqa: E731
        check(lib.omc_band_sample_canonical(self._ctx, n, w, C.byref(T), self._p(rhs_chain, Cn, n), ld(rhs_chain),
                                            self._p(z, Cn, n), ld(z), int(draw_index), self._p(x_out, Cn, n), ld(x_out),
                                            self._p(mean_out, Cn, n), ld(mean_out), self._chain_scalar(logdet_out)))
        return x_out

    def band_quadform(self, n, band, x, quad_out, center=None):
        w = 0 if band is None else band.shape[0] - 1
        check(lib.omc_band_quadform(self._ctx, n, w, self._p(band), self._vec(center, n), self._p(x, self.n_chains, n),
                                    x.stride(0), self._chain_scalar(quad_out)))
        return quad_out

    def band_matvec_chain(self, n, band, v, scale=None, out=None, accumulate=False):
        """out[c] (+)= scale[c] * M v_c for a shared band matrix (band rows as in band_terms; None = identity)."""
        out = self.empty(self.n_chains, n) if out is None else out
        w = 0 if band is None else band.shape[0] - 1
        check(lib.omc_band_matvec_chain(self._ctx, n, w, self._p(band), self._p(v, self.n_chains, n), v.stride(0),
                                        self._chain_scalar(scale), self._p(out, self.n_chains, n), out.stride(0), int(accumulate)))
        return out

    # ------------------------------------------------------------------ truncated Gaussian conditional
    def band_gibbs_truncated(self, n, terms, x, lower=None, upper=None, u=None, rhs_chain=None, draw_index=0):
        """One scan of single-site truncated updates under a banded precision (omc_band_gibbs_truncated); x in place."""
        T, w = terms if isinstance(terms, tuple) else self.band_terms(terms, n)
        Cn = self.n_chains
        ld = lambda t: 0 if t is None else t.stride(0)  # noqa: E731
        check(lib.omc_band_gibbs_truncated(self._ctx, n, w, C.byref(T), self._p(rhs_chain, Cn, n), ld(rhs_chain),
                                           self._vec(lower, n), self._vec(upper, n), self._p(u, Cn, n), ld(u), int(draw_index),
                                           self._p(x, Cn, n), ld(x)))
        return x

    def tridiag_gibbs_truncated(self, n, terms, x, lower=None, upper=None, u=None, rhs_chain=None, draw_index=0):
        """One in-place scan of gmrf.gibbs_canonical_truncated_normal on x (C, n); lower / upper: device (n,) or None."""
        T = terms if isinstance(terms, _abi.TridiagTerms) else self.tridiag_terms(terms, n)
        Cn = self.n_chains
        ld = lambda t: 0 if t is None else t.stride(0)  # noqa: E731
        check(lib.omc_tridiag_gibbs_truncated(self._ctx, n, C.byref(T), self._p(rhs_chain, Cn, n), ld(rhs_chain),
                                              self._vec(lower, n), self._vec(upper, n), self._p(u, Cn, n), ld(u),
                                              int(draw_index), self._p(x, Cn, n), ld(x)))
        return x

    def dense_gibbs_truncated(self, p, terms, x, lower=None, upper=None, u=None, rhs_chain=None, draw_index=0):
        T = terms if isinstance(terms, _abi.DenseTerms) else self.dense_terms(terms, p)
        Cn = self.n_chains
        ld = lambda t: 0 if t is None else t.stride(0)  # noqa: E731
        check(lib.omc_dense_gibbs_truncated(self._ctx, p, C.byref(T), self._p(rhs_chain, Cn, p), ld(rhs_chain),
                                            self._vec(lower, p), self._vec(upper, p), self._p(u, Cn, p), ld(u),
                                            int(draw_index), self._p(x, Cn, p), ld(x)))
        return x

    def domain_penalty(self, x, out, lower=None, upper=None):
        """out[c] = -inf where any element of x[c] (C, n) lies outside [lower, upper]."""
        Cn, n = x.shape
        check(lib.omc_domain_penalty(self._ctx, n, self._p(x, Cn, n), x.stride(0), self._vec(lower, n), self._vec(upper, n),
                                     self._chain_scalar(out)))
        return out

    # ------------------------------------------------------------------ generic MH / ragged state / RJ transitions
    def _i32(self, t):
        torch = _torch()
        if t.dtype != torch.int32 or not t.is_cuda or t.numel() != self.n_chains:
            raise TypeError("expected an int32 ROCm tensor with one entry per chain")
        return C.c_void_p(t.data_ptr())

    def _i32_block(self, t, rows):
        """(rows, C) int32 output block, or None."""
        if t is None:
            return None
        torch = _torch()
        if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != (rows, self.n_chains):
            raise TypeError(f"expected a contiguous ({rows}, {self.n_chains}) int32 ROCm tensor")
        return C.c_void_p(t.data_ptr())

    def _i64(self, t):
        torch = _torch()
        if t.dtype != torch.int64 or not t.is_cuda or t.numel() != self.n_chains:
            raise TypeError("expected an int64 ROCm tensor with one entry per chain")
        return C.c_void_p(t.data_ptr())

    def rw_propose(self, x, z_out, step, lower=None, upper=None, column=None, count=None, inject=None, draw_index=0,
                   sub=0):
        """RandomWalk.proposal for column `column` (None: n_rep must be 1) of the (C, p, n_rep) tensor x,
        written into the same column of z_out.  step/lower/upper: device tensors (step of 1 or p entries).
        Returns (lq_fwd, lq_rev), each (C,)."""
        Cn, p, n_rep = x.shape
        if z_out.shape != x.shape or not x.is_contiguous() or not z_out.is_contiguous():
            raise ValueError("x and z_out must be contiguous tensors of the same (C, p, n_rep) shape")
        col = 0 if column is None else int(column)
        if column is None and n_rep != 1:
            raise ValueError("column is required when the parameter has replicates")
        lq_f, lq_r = self.empty(Cn), self.empty(Cn)
        esz = x.element_size()
        check(lib.omc_rw_propose(self._ctx, p, C.c_void_p(x.data_ptr() + col * esz), p * n_rep, n_rep, self._p(step),
                                 0 if step.numel() == 1 else 1, self._p(lower), self._p(upper),
                                 self._chain_scalar(count), col, self._p(inject), int(draw_index), int(sub),
                                 C.c_void_p(z_out.data_ptr() + col * esz), p * n_rep, n_rep, self._p(lq_f), self._p(lq_r)))
        return lq_f, lq_r

    def mh_accept(self, lp_cur, lp_prop, lq_fwd=None, lq_rev=None, count=None, index=0, u=None, draw_index=0, sub=0,
                  accept_count=None, proposal_count=None, log_alpha=None):
        """accept (C,) int32 of the Metropolis-Hastings test; counters are updated in place."""
        torch = _torch()
        acc = torch.empty(self.n_chains, dtype=torch.int32, device=self.device)
        check(lib.omc_mh_accept(self._ctx, self._chain_scalar(lp_cur), self._chain_scalar(lp_prop),
                                self._chain_scalar(lq_fwd), self._chain_scalar(lq_rev), self._chain_scalar(count),
                                int(index), self._chain_scalar(u), int(draw_index), int(sub), self._i32(acc),
                                self._chain_scalar(log_alpha),
                                None if accept_count is None else self._i64(accept_count),
                                None if proposal_count is None else self._i64(proposal_count)))
        return acc

    def chain_select(self, accept, src, dst):
        """dst[c] = src[c] for accepted chains (tensors of identical shape, chain-major, contiguous)."""
        if src.shape != dst.shape or not src.is_contiguous() or not dst.is_contiguous():
            raise ValueError("src and dst must be contiguous and of the same shape")
        width = src.numel() // self.n_chains
        check(lib.omc_chain_select(self._ctx, self._i32(accept), width, self._p(src.view(self.n_chains, -1)),
                                   self._p(dst.view(self.n_chains, -1))))
        return dst

    def chain_select_many(self, accept, pairs):
        """dst[c] = src[c] on the accepted chains for every (src, dst) pair, in one launch per OMC_SELECT_MAX pairs."""
        n = self.n_chains
        for k in range(0, len(pairs), _abi.OMC_SELECT_MAX):
            part = pairs[k:k + _abi.OMC_SELECT_MAX]
            m = len(part)
            widths, srcs, dsts = (C.c_int64 * m)(), (C.c_void_p * m)(), (C.c_void_p * m)()
            for e, (src, dst) in enumerate(part):
                if src.shape != dst.shape or not src.is_contiguous() or not dst.is_contiguous():
                    raise ValueError("src and dst must be contiguous and of the same shape")
                widths[e] = src.numel() // n
                srcs[e], dsts[e] = self._p(src.view(n, -1)), self._p(dst.view(n, -1))
            check(lib.omc_chain_select_multi(self._ctx, self._i32(accept), m, widths, srcs, dsts))
        self.note_write(*[d for _, d in pairs])

    def ragged_resize(self, src, count, birth, del_index, axis, new_vals=None, physical_transposed=False):
        """np.concatenate / np.delete along the ragged axis of a (C, p, n_rep) tensor for every chain.
        axis = 0: rows are ragged (beta (k, 1)); axis = 1: columns (theta (1, k), basis (n, k))."""
        torch = _torch()
        Cn, p, n_rep = src.shape
        dst = torch.empty_like(src)
        if axis == 1:
            rows, kmax = p, n_rep
        else:
            rows, kmax = n_rep, p
        st = src.stride()
        rs, js = (st[1], st[2]) if axis == 1 else (st[2], st[1])
        if dst.stride() != st:
            raise ValueError("unsupported memory layout")
        check(lib.omc_ragged_resize(self._ctx, rows, kmax, self._chain_scalar(count), self._i32(birth),
                                    self._i64(del_index), self._p(new_vals), C.c_void_p(src.data_ptr()),
                                    C.c_void_p(dst.data_ptr()), st[0], rs, js))
        return dst

    def gaussian_basis(self, X, knots, out, count=None, scales=None, scale=1.0, column=None, prev_count=None):
        """Gaussian-kernel basis on per-chain knots written into out (C, kmax, n); column: rewrite only that column;
        prev_count (C,): out already holds zeros in the columns >= prev_count[c] -- those beyond count[c] too are skipped."""
        Cn, kmax, n = out.shape
        if not out.is_contiguous():
            raise ValueError("out must be a contiguous (C, kmax, n) tensor")
        check(lib.omc_gaussian_basis(self._ctx, n, kmax, self._vec(X, n), self._p(knots, Cn, kmax), self._p(scales),
                                     float(scale), self._chain_scalar(count), self._chain_scalar(prev_count),
                                     -1 if column is None else int(column),
                                     self._p(out.view(Cn, -1))))
        return out

    def knot_loop(self, X, scale, y, B, beta, theta, count, step, lower, upper, add_shared=None, add_chain=None, w=None,
                  tau=None, inject_z=None, inject_u=None, draw_index=0, accept_count=None, proposal_count=None,
                  accept_out=None, log_alpha_out=None):
        """RandomWalkLoop over the knots of a Gaussian-kernel basis under a regression likelihood in one launch
        (omc_knot_loop): theta (C, kmax) and B (C, kmax, n) are updated in place."""
        Cn, kmax, n = B.shape
        if not B.is_contiguous() or not theta.is_contiguous():
            raise ValueError("B (C, kmax, n) and theta (C, kmax) must be contiguous")
        check(lib.omc_knot_loop(self._ctx, n, kmax, self._vec(X, n), float(scale), self._vec(y, n), self._vec(add_shared, n),
                                self._p(add_chain, Cn, n), self._vec(w, n), self._chain_scalar(tau), self._p(beta, Cn, kmax),
                                self._p(theta, Cn, kmax), self._chain_scalar(count), self._p(B.view(Cn, -1)), float(step),
                                float(lower), float(upper), self._p(inject_z), self._p(inject_u), int(draw_index),
                                None if accept_count is None else self._i64(accept_count),
                                None if proposal_count is None else self._i64(proposal_count),
                                self._i32_block(accept_out, kmax), self._p(log_alpha_out)))

    def design_predict_batched(self, B, coef, add_chain=None, add_shared=None, alpha=1.0, chain_scale=None, out=None):
        """out[c] = chain_scale[c] * (alpha * B_c coef_c + add_chain[c] + add_shared);
        B: (C, kmax, n) contiguous (column j of chain c contiguous)."""
        Cn, kmax, n = B.shape
        if not B.is_contiguous():
            raise ValueError("B must be a contiguous (C, kmax, n) tensor")
        out = self.empty(Cn, n) if out is None else out
        check(lib.omc_design_predict_batched(self._ctx, n, kmax, self._p(B.view(Cn, -1)), self._p(coef, Cn, kmax),
                                             self._p(add_chain), self._vec(add_shared, n), float(alpha),
                                             self._chain_scalar(chain_scale), self._p(out)))
        return out

    def design_resid_sq_batched(self, B, coef, y, add_chain=None, add_shared=None, w=None, out=None):
        """out[c] = sum_i w_i (y_i - (B_c coef_c + add_chain[c] + add_shared)_i)^2 without the fitted values in memory."""
        Cn, kmax, n = B.shape
        if not B.is_contiguous():
            raise ValueError("B must be a contiguous (C, kmax, n) tensor")
        out = self.empty(Cn) if out is None else out
        check(lib.omc_design_resid_sq_batched(self._ctx, n, kmax, self._p(B.view(Cn, -1)), self._p(coef, Cn, kmax),
                                              self._p(add_chain), self._vec(add_shared, n), self._vec(y, n),
                                              self._vec(w, n), self._chain_scalar(out)))
        return out

    def design_gram_batched(self, B, w=None, resid_shared=None, resid_chain=None, count=None):
        """(gram (C, kmax, kmax), rhs (C, kmax) or None) = (B_c' W B_c, B_c' W (resid_shared - resid_chain[c]));
        count (C,): only the leading count[c] columns of chain c are live (the rest of the outputs is 0)."""
        Cn, kmax, n = B.shape
        if not B.is_contiguous():
            raise ValueError("B must be a contiguous (C, kmax, n) tensor")
        gram = self.empty(Cn, kmax, kmax)
        want_rhs = resid_shared is not None or resid_chain is not None
        rhs = self.empty(Cn, kmax) if want_rhs else None
        check(lib.omc_design_gram_batched(self._ctx, n, kmax, self._p(B.view(Cn, -1)), self._vec(w, n),
                                          self._vec(resid_shared, n), self._p(resid_chain), self._chain_scalar(count),
                                          self._p(gram.view(Cn, -1)), self._p(rhs)))
        return gram, rhs

    def design_gram_select(self, B, count, B_alt, count_alt, select, w=None):
        """gram (C, kmax, kmax) of B_alt[c] (live columns count_alt[c]) where select[c] != 0, of B[c] otherwise."""
        Cn, kmax, n = B.shape
        if not B.is_contiguous() or not B_alt.is_contiguous() or B_alt.shape != B.shape:
            raise ValueError("B and B_alt must be contiguous (C, kmax, n) tensors of one shape")
        gram = self.empty(Cn, kmax, kmax)
        check(lib.omc_design_gram_select(self._ctx, n, kmax, self._p(B.view(Cn, -1)), self._chain_scalar(count),
                                         self._p(B_alt.view(Cn, -1)), self._chain_scalar(count_alt), self._i32(select),
                                         self._vec(w, n), self._p(gram.view(Cn, -1))))
        return gram

    def small_sample_canonical(self, gram, gram_rhs, prior_prec, lik_scale=None, prior_mean=None, count=None, z=None,
                               draw_index=0, mean_out=None):
        Cn, kmax, _ = gram.shape
        x = self.empty(Cn, kmax)
        check(lib.omc_small_sample_canonical(self._ctx, kmax, self._p(gram.view(Cn, -1)), self._p(gram_rhs),
                                             self._chain_scalar(lik_scale), self._p(prior_prec), self._p(prior_mean),
                                             self._chain_scalar(count), self._p(z), int(draw_index), self._p(x),
                                             self._p(mean_out)))
        return x

    def small_spd_ops(self, A, v=None, want_Av=False, want_quad=False, want_logdet=False):
        """(Av, quad, logdet) for per-chain small SPD matrices A (C, k, k) and vectors v (C, k) (omc_small_spd_ops)."""
        Cn, k, _ = A.shape
        Av = self.empty(Cn, k) if want_Av else None
        quad = self.empty(Cn) if want_quad else None
        logdet = self.empty(Cn) if want_logdet else None
        check(lib.omc_small_spd_ops(self._ctx, k, self._p(A.contiguous().view(Cn, -1)), self._p(v), self._p(Av), self._chain_scalar(quad),
                                    self._chain_scalar(logdet)))
        return Av, quad, logdet

    # per-chain SPD matrices beyond one wave's 64 columns (the generic ManifoldMALA route with a parameter-dependent Hessian):
    # batched dense factorisations as tensor expressions on the context's stream -- a rarely taken generic branch, natural-order
    # Cholesky like everywhere else (the factor is unique: the same draw for the same z as the small-matrix kernels)
    def chain_spd_ops(self, A, v=None, want_Av=False, want_quad=False, want_logdet=False):
        torch = _torch()
        Av = torch.bmm(A, v.unsqueeze(2)).squeeze(2) if (want_Av or want_quad) else None
        quad = (v * Av).sum(dim=1) if want_quad else None
        logdet = None
        if want_logdet:
            L, info = torch.linalg.cholesky_ex(A)
            if bool((info != 0).any().item()):
                raise np.linalg.LinAlgError(f"Matrix is not positive definite (chain {int(torch.nonzero(info)[0].item())})")
            logdet = 2.0 * torch.log(torch.diagonal(L, dim1=1, dim2=2)).sum(dim=1)
        return (Av if want_Av else None), quad, logdet

    def chain_sample_canonical(self, A, b, z=None, draw_index=0, mean_out=None):
        """x = A^-1 b + L^-T z per chain, L = chol(A) (gmrf.py:167-198 for a dense per-chain precision of any order)."""
        torch = _torch()
        Cn, k, _ = A.shape
        L, info = torch.linalg.cholesky_ex(A)
        if bool((info != 0).any().item()):
            raise np.linalg.LinAlgError(f"Matrix is not positive definite (chain {int(torch.nonzero(info)[0].item())})")
        w = torch.linalg.solve_triangular(L, b.unsqueeze(2), upper=False)
        if mean_out is not None:
            mean_out.copy_(torch.linalg.solve_triangular(L.transpose(1, 2), w, upper=True).squeeze(2))
        zz = self.fill_normal(k, draw_index) if z is None else z
        return torch.linalg.solve_triangular(L.transpose(1, 2), w + zz.unsqueeze(2), upper=True).squeeze(2).contiguous()

    def rj_matched_transition(self, gram_cur, gram_prop, count, birth, del_index, coef_cur, scale, limits, lq_fwd,
                              lq_rev, inject=None, draw_index=0, sub=0):
        """coef_prop (C, kmax); lq_fwd / lq_rev (C,) are added to in place."""
        Cn, kmax, _ = gram_cur.shape
        out = self.empty(Cn, kmax)
        lo, hi = (0.0, 0.0) if limits is None else (float(limits[0]), float(limits[1]))
        check(lib.omc_rj_matched_transition(self._ctx, kmax, self._p(gram_cur.view(Cn, -1)), self._p(gram_prop.view(Cn, -1)),
                                            self._chain_scalar(count), self._i32(birth), self._i64(del_index),
                                            self._p(coef_cur, Cn, kmax), float(scale), int(limits is not None), lo, hi,
                                            self._chain_scalar(inject), int(draw_index), int(sub), self._p(out),
                                            self._chain_scalar(lq_fwd), self._chain_scalar(lq_rev)))
        return out

    def log_transform(self, x):
        """(log x (C, n), sum_i log x (C,)) of a per-chain vector."""
        Cn, n = x.shape
        out, sumlog = self.empty(Cn, n), self.empty(Cn)
        check(lib.omc_log_transform(self._ctx, n, self._p(x, Cn, n), x.stride(0), self._p(out), n, self._p(sumlog)))
        return out, sumlog

    def dense_quadform(self, M, x, center=None, M_center=None):
        """(C,) tensor (x - m)' M (x - m) for a dense shared M (device (n, n)): one GEMM + one reduction kernel."""
        Cn, n = x.shape
        y = self.design_predict(M, x)  # M symmetric: rows of x times M
        out = self.empty(Cn)
        check(lib.omc_centered_rowdot(self._ctx, n, self._p(x, Cn, n), x.stride(0), self._vec(center, n), self._p(y, Cn, n),
                                      y.stride(0), self._vec(M_center, n), self._p(out)))
        return out

    def uniform_draw(self, lower, rng, inject=None, draw_index=0, sub=0):
        """(C, p) tensor lower + range * U(0, 1]."""
        p = lower.numel()
        out = self.empty(self.n_chains, p)
        check(lib.omc_uniform_draw(self._ctx, p, self._p(lower), self._p(rng), self._p(inject), int(draw_index), int(sub),
                                   self._p(out)))
        return out

    def diag_gauss_logpdf(self, x, prec, out, mean=None, count=None, accumulate=False):
        Cn, kmax = x.shape
        check(lib.omc_diag_gauss_logpdf(self._ctx, kmax, self._p(x), self._p(mean), self._p(prec),
                                        self._chain_scalar(count), self._chain_scalar(out), int(accumulate)))

    def gamma_logpdf_ragged(self, x, shape, rate, out, count=None, last_only=False, accumulate=False):
        Cn, kmax = x.shape
        check(lib.omc_gamma_logpdf_ragged(self._ctx, kmax, self._p(x), self._chain_scalar(count), float(shape), float(rate),
                                          int(last_only), self._chain_scalar(out), int(accumulate)))
        return out

    def diag_gauss_grad(self, x, prec, mean=None, count=None):
        Cn, kmax = x.shape
        grad = self.empty(Cn, kmax)
        check(lib.omc_diag_gauss_grad(self._ctx, kmax, self._p(x), self._p(mean), self._p(prec), self._chain_scalar(count),
                                      self._p(grad)))
        return grad

    def mala_diag(self, x, grad, hdiag, step, count=None, x_other=None, z=None, draw_index=0, sub=0):
        """ManifoldMALA with a diagonal Hessian.  x_other None: propose -> (x_proposed (C, kmax), log q(x'|x));
        x_other given: -> log q(x_other | x) of the reverse move."""
        Cn, kmax = x.shape
        propose = x_other is None
        out = self.empty(Cn, kmax) if propose else x_other
        lq = self.empty(Cn)
        check(lib.omc_mala_diag(self._ctx, kmax, self._p(x), self._p(grad), self._p(hdiag), self._chain_scalar(count),
                                float(step), int(propose), self._p(z), int(draw_index), int(sub), self._p(out), self._p(lq)))
        return (out, lq) if propose else lq

    def poisson_draw(self, rate, u=None, draw_index=0):
        """One Poisson(rate) count per chain as a (C,) float64 tensor (omc_poisson_draw); rate: float or (C,) tensor;
        u: (C, K) injected uniforms, NaN-padded."""
        shared = not hasattr(rate, "data_ptr")
        r = self.full((1,), float(rate)) if shared else rate
        out = self.empty(self.n_chains)
        check(lib.omc_poisson_draw(self._ctx, self._p(r), 0 if shared else 1, self._p(u, self.n_chains, 1) if u is not None else None,
                                   0 if u is None else u.stride(0), int(draw_index), self._p(out)))
        return out

    def poisson_logpmf(self, x, rate, out, accumulate=False):
        check(lib.omc_poisson_logpmf(self._ctx, self._chain_scalar(x), float(rate), self._chain_scalar(out), int(accumulate)))

    def count_logpdf(self, count, per_element, out, accumulate=False):
        check(lib.omc_count_logpdf(self._ctx, self._chain_scalar(count), float(per_element), self._chain_scalar(out),
                                   int(accumulate)))

    def mixture_gather(self, param, alloc, count=None, fill=0.0):
        """out[c][j] = param[alloc[c][j]]; param: (m,) shared or (C, m) per chain."""
        Cn, kmax = alloc.shape
        out = self.empty(Cn, kmax)
        per_chain = param.dim() == 2
        m = param.shape[-1]
        check(lib.omc_mixture_gather(self._ctx, kmax, m, self._p(param), m if per_chain else 0, self._p(alloc),
                                     self._chain_scalar(count), float(fill), self._p(out)))
        return out

    def mixture_gather2(self, alloc, count, param_a, fill_a, param_b, fill_b):
        """(param_a[alloc], param_b[alloc]) with the fills beyond each chain's live length, in one launch; both tables
        (m,) shared or (C, m) per chain, of one length m."""
        Cn, kmax = alloc.shape
        out_a, out_b = self.empty(Cn, kmax), self.empty(Cn, kmax)
        m = param_a.shape[-1]
        if param_b.shape[-1] != m:
            raise ValueError("the two tables must have one length")
        check(lib.omc_mixture_gather2(self._ctx, kmax, m, self._p(alloc), self._chain_scalar(count), self._p(param_a),
                                      m if param_a.dim() == 2 else 0, float(fill_a), self._p(out_a), self._p(param_b),
                                      m if param_b.dim() == 2 else 0, float(fill_b), self._p(out_b)))
        return out_a, out_b

    def mixture_allocation(self, y, prior, mean, prec, u=None, draw_index=0):
        """MixtureAllocation.sample: y (C, p), prior (1 or p, K) shared, mean / prec (K,) shared or (C, K) -> alloc (C, p)."""
        Cn, p = y.shape
        K = prior.shape[-1]
        out = self.empty(Cn, p)
        stride = lambda t: K if t.dim() == 2 else 0  # noqa: E731
        check(lib.omc_mixture_allocation(self._ctx, p, K, self._p(y, Cn, p), self._p(prior), prior.shape[0], self._p(mean),
                                         stride(mean), self._p(prec), stride(prec), self._p(u), int(draw_index), self._p(out)))
        return out

    def categorical_logpmf(self, alloc, prob, out, accumulate=False):
        Cn, p = alloc.shape
        check(lib.omc_categorical_logpmf(self._ctx, p, prob.shape[-1], self._p(alloc, Cn, p), self._p(prob), prob.shape[0],
                                         self._chain_scalar(out), int(accumulate)))
        return out

    def mixture_normal_gamma(self, resid, alloc, a0, b0, g=None, draw_index=0):
        Cn, p = resid.shape
        K = a0.numel()
        out = self.empty(Cn, K)
        check(lib.omc_mixture_normal_gamma(self._ctx, p, K, self._p(resid, Cn, p), self._p(alloc, Cn, p), self._p(a0),
                                           self._p(b0), self._p(g), int(draw_index), self._p(out)))
        return out

    def gamma_logpdf_vec(self, x, shape, rate, out, accumulate=False):
        Cn, K = x.shape
        check(lib.omc_gamma_logpdf_vec(self._ctx, K, self._p(x, Cn, K), self._p(shape), self._p(rate),
                                       self._chain_scalar(out), int(accumulate)))
        return out

    # ------------------------------------------------------------------ posterior summaries
    def store_moments(self, store, pooled=False):
        """(mean, var) of a device store (n_iter, C, size): per chain (C, size) or pooled (size,)."""
        if store.dim() != 3 or store.shape[1] != self.n_chains or not store.is_contiguous():
            raise ValueError("store must be a contiguous (n_iter, C, size) tensor")
        n_iter, _, size = store.shape
        shape = (size,) if pooled else (self.n_chains, size)
        mean, var = self.empty(*shape), self.empty(*shape)
        check(lib.omc_store_moments(self._ctx, n_iter, size, self._p(store), int(pooled), self._p(mean), self._p(var)))
        return mean, var

    def store_quantiles(self, store, q, pooled=False, omit_nan=True):
        """np.quantile (omit_nan=False) / np.nanquantile (True) of a device store (n_iter, C, size) along the iterations,
        default "linear" method: (len(q), C, size) per chain or (len(q), size) pooled over chains; computed on the device."""
        if store.dim() != 3 or store.shape[1] != self.n_chains or not store.is_contiguous():
            raise ValueError("store must be a contiguous (n_iter, C, size) tensor")
        qs = np.ascontiguousarray(np.atleast_1d(np.asarray(q, dtype=np.float64)))
        if qs.ndim != 1 or qs.size < 1:
            raise ValueError("q must be a scalar or a one-dimensional sequence")
        if not np.all((qs >= 0) & (qs <= 1)):
            raise ValueError("Quantiles must be in the range [0, 1]")  # np.quantile's own message
        n_iter, _, size = store.shape
        out = self.empty(*((qs.size, size) if pooled else (qs.size, self.n_chains, size)))
        check(lib.omc_store_quantiles(self._ctx, n_iter, size, self._p(store), int(pooled), qs.size,
                                      qs.ctypes.data_as(C.POINTER(C.c_double)), int(bool(omit_nan)), self._p(out)))
        return out

    def store_thin(self, store, every, first=0):
        """store[first::every] of a device store (n_iter, C, ...) as a packed device tensor (one launch)."""
        if store.dim() < 2 or store.shape[1] != self.n_chains or not store.is_contiguous():
            raise ValueError("store must be a contiguous (n_iter, C, ...) tensor")
        n_iter = store.shape[0]
        every, first = int(every), int(first)
        if every < 1 or not 0 <= first < n_iter:
            raise ValueError("every must be >= 1 and first inside the store")
        n_out = (n_iter - first + every - 1) // every
        size = int(np.prod(store.shape[2:])) if store.dim() > 2 else 1
        out = self.empty(*((n_out,) + tuple(store.shape[1:])))
        check(lib.omc_store_thin(self._ctx, n_iter, size, self._p(store), first, every, self._p(out), None))
        return out

    # ------------------------------------------------------------------ the gather of the stores (RCCL)
    def communicator(self, world, rank, unique_id):
        """RCCL communicator of this rank (omc_comm_create); collective over all ranks.  `unique_id` is the bytes
        object one rank obtained from `new_unique_id()` and shipped to the others."""
        return Communicator(self, world, rank, unique_id)

    # ------------------------------------------------------------------ random fills
    def fill_normal(self, n, draw_index=0):
        out = self.empty(self.n_chains, n)
        check(lib.omc_fill_normal(self._ctx, n, int(draw_index), self._p(out), n))
        return out

    def fill_philox_u32(self, n_words, draw_index=0):
        torch = _torch()
        out = torch.empty(self.n_chains, n_words, dtype=torch.int32, device=self.device)
        check(lib.omc_fill_philox_u32(self._ctx, n_words, int(draw_index), C.c_void_p(out.data_ptr()), n_words))
        return out


# Engine methods that write into tensors the caller hands in (parameter names): the calls are recorded (Engine.note_write) so that
# a cache derived from a state tensor can tell whether the library has touched that tensor since.  (Methods that return fresh
# tensors need no entry.  The whitened Metropolis-Hastings steps write x in place behind torch's version counter like the others
# and are recorded like them: a quadratic form cached for a Normal-Gamma block further down the sweep must not survive them; the
# samplers take `_white_serial` AFTER the call, so their own whitened image stays valid.)
_WRITES = {
    "tridiag_sample_canonical": ("x_out", "mean_out"), "gmrf_sweep": ("x_out",), "dense_sample_canonical": ("x_out", "mean_out"),
    "dense_spectral_sample": ("x_out", "mean_out"), "band_sample_canonical": ("x_out", "mean_out"),
    "band_gibbs_truncated": ("x",), "tridiag_gibbs_truncated": ("x",), "dense_gibbs_truncated": ("x",),
    "chain_lincomb": ("out",), "chain_copy": ("dst",), "chain_select": ("dst",), "mala_step": ("x",), "rw_step": ("x",),
    "tridiag_matvec_chain": ("out",), "band_matvec_chain": ("out",), "design_predict": ("fitted",),
    "design_predict_batched": ("out",), "knot_loop": ("B", "beta", "theta"), "gaussian_basis": ("out",), "mala_diag": ("x",),
    "mala_step_white": ("x",), "rw_step_white": ("x",), "mala_run_white": ("x", "x_store"),
}


def _recording(fn, written):
    import inspect

    names = list(inspect.signature(fn).parameters)  # includes self
    spec = tuple((names.index(w), w) for w in written)

    def wrapper(self, *args, **kwargs):
        out = fn(self, *args, **kwargs)
        self._write_serial += 1
        for pos, name in spec:
            t = args[pos - 1] if pos - 1 < len(args) else kwargs.get(name)
            if t is not None:
                self._written[t.untyped_storage().data_ptr()] = self._write_serial
        return out

    wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
    return wrapper


for _name, _written in _WRITES.items():
    setattr(Engine, _name, _recording(getattr(Engine, _name), _written))


def gather_local(engine, blocks, root=0, staging_limit_bytes=0):
    """omc_gather_samples_local: the blocks of several shards that live on THIS GPU, blocks[r] = (n_outer, counts[r], ...),
    joined into (n_outer, sum(counts), ...) in rank order -- the root's side of omc_gather_samples with device-to-device
    copies where the RCCL receives stand."""
    blocks = [b.contiguous() for b in blocks]
    world = len(blocks)
    n_outer = int(blocks[0].shape[0])
    tail = tuple(blocks[0].shape[2:])
    if any(b.dim() < 2 or int(b.shape[0]) != n_outer or tuple(b.shape[2:]) != tail for b in blocks):
        raise ValueError("blocks must be (n_outer, counts[r], ...) with equal outer and trailing dimensions")
    counts = [int(b.shape[1]) for b in blocks]
    row = int(np.prod(tail)) if tail else 1
    out = engine.empty(n_outer, sum(counts), *tail)
    ptrs = (C.c_void_p * world)(*[(engine._p(b) if b.numel() else None) for b in blocks])
    arr = (C.c_int64 * world)(*counts)
    check(lib.omc_gather_samples_local(engine._ctx, world, ptrs, n_outer, row, arr, engine._p(out) if out.numel() else None,
                                       int(root), int(staging_limit_bytes)))
    return out


def new_unique_id():
    """Bootstrap id of an RCCL communicator (omc_comm_unique_id); call on ONE rank and ship the bytes to the others."""
    n = int(lib.omc_comm_unique_id_bytes())
    buf = C.create_string_buffer(n)
    check(lib.omc_comm_unique_id(buf, n))
    return buf.raw


class Communicator:
    """omc_comm of one rank: the library's own RCCL communicator, used by exactly one collective of the path, the
    gather of the per-rank stores on a root (omc_gather_samples)."""

    def __init__(self, engine, world, rank, unique_id):
        self.engine, self.world, self.rank = engine, int(world), int(rank)
        comm = C.c_void_p()
        check(lib.omc_comm_create(engine._ctx, self.world, self.rank, unique_id, len(unique_id), C.byref(comm)))
        self._comm = comm

    def gather(self, block, counts, root=0, staging_limit_bytes=0):
        """block (n_outer, counts[rank], ...) float64 device tensor of this rank -> on `root` the tensor
        (n_outer, sum(counts), ...) with the chains in rank order; None on the other ranks.  Asynchronous on the
        engine's stream, like every other call."""
        eng = self.engine
        counts = [int(c) for c in counts]
        if len(counts) != self.world:
            raise ValueError("counts must have one entry per rank")
        if block.dim() < 2 or block.shape[1] != counts[self.rank]:
            raise ValueError("block must be (n_outer, counts[rank], ...)")
        block = block.contiguous()
        n_outer = int(block.shape[0])
        row = int(np.prod(block.shape[2:])) if block.dim() > 2 else 1
        out = eng.empty(n_outer, sum(counts), *block.shape[2:]) if self.rank == root else None
        arr = (C.c_int64 * self.world)(*counts)
        check(lib.omc_gather_samples(eng._ctx, self._comm, eng._p(block) if block.numel() else None, n_outer, row, arr,
                                     eng._p(out) if out is not None and out.numel() else None, int(root), int(staging_limit_bytes)))
        return out

    def close(self):
        if self._comm is not None:
            lib.omc_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
