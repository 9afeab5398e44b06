"""Normal distribution in mean/precision form (reference distribution/location_scale.py:126-272).

On the GPU path a Normal is consumed structurally: precision = ScaledMatrix(shared matrix, per-chain
scalar) or Identity(shared matrix); exactly one of {response, mean} is per chain.  `structure()`
exposes that view to the samplers; log_p evaluates it with omc_tridiag_quadform +
omc_scaled_gauss_logpdf.
"""

from dataclasses import dataclass
from typing import Union

import numpy as np
from scipy import sparse

from openmcmc_amd.chains import is_chain
from openmcmc_amd.distribution.distribution import Distribution
from openmcmc_amd.parameter import (Identity, LinearCombination, MixtureParameterMatrix, MixtureParameterVector,
                                     ScaledMatrix, _is_identity)


def tridiagonal_bands(M, n):
    """(diag, off) of a symmetric matrix with bandwidth <= 1, or None if it is wider.
    diag is None for an exact identity (the kernels then skip the load)."""
    if sparse.issparse(M):
        M = M.tocsr()
        coo = M.tocoo()
        if coo.nnz and np.max(np.abs(coo.row - coo.col)) > 1:
            return None
        diag = np.asarray(M.diagonal(), dtype=np.float64)
        off = np.asarray(M.diagonal(1), dtype=np.float64) if n > 1 else np.zeros(0)
        low = np.asarray(M.diagonal(-1), dtype=np.float64) if n > 1 else np.zeros(0)
    else:
        M = np.asarray(M, dtype=np.float64)
        if np.any(np.triu(M, 2)) or np.any(np.tril(M, -2)):
            return None
        diag, off, low = np.diag(M).copy(), np.diag(M, 1).copy(), np.diag(M, -1).copy()
    if not np.array_equal(off, low):
        raise ValueError("precision matrix is not symmetric")
    if not off.any():
        off = None
    if off is None and np.all(diag == 1.0):
        diag = None
    return diag, off


from collections import namedtuple

# Hessian of branch (ii) of Normal.grad_log_p: scale[c] * matrix, the per-chain scalar kept apart from the shared matrix
ScaledHessian = namedtuple("ScaledHessian", ["scale", "matrix"])

BAND_MAX = 128  # widest band the band kernel takes (omc_band_sample_canonical)


def band_storage(M, n):
    """(w+1, n) array with row d = the d-th sub-diagonal of the symmetric matrix M padded with zeros (the layout of
    omc_band_terms), or None when the bandwidth exceeds BAND_MAX or a band factorisation would not pay (w+1 >= n/2)."""
    if sparse.issparse(M):
        coo = M.tocoo()
        w = int(np.max(np.abs(coo.row - coo.col))) if coo.nnz else 0
    else:
        A = np.asarray(M)
        nz = np.nonzero(A)
        w = int(np.max(np.abs(nz[0] - nz[1]))) if nz[0].size else 0
    if w > BAND_MAX or (w + 1) * 2 > n:
        return None
    out = np.zeros((w + 1, n))
    for d in range(w + 1):
        lower = M.diagonal(-d) if sparse.issparse(M) else np.diag(np.asarray(M), -d)
        upper = M.diagonal(d) if sparse.issparse(M) else np.diag(np.asarray(M), d)
        if not np.array_equal(lower, upper):
            raise ValueError("precision matrix is not symmetric")
        out[d, : n - d] = lower
    return out


@dataclass
class NormalStructure:
    """What the samplers need to know about one Normal: precision = scale * M."""

    n: int
    matrix: object          # host matrix M (n x n)
    scale_key: object       # state label of the scalar, or None (scale 1)
    diag: object            # tridiagonal bands of M (None, None) if M is the identity; False = wider than tridiagonal
    off: object
    n_pos: int              # #{diag(M) > 0}  (sampler.py:283)
    band: object = None     # (w+1, n) band storage when M is banded but wider than tridiagonal, else None

    def band_rows(self):
        """Band storage of M whatever its structure class (identity -> None = the kernels' identity)."""
        if self.diag is False:
            return self.band
        if self.diag is None and self.off is None:
            return None
        diag = np.ones(self.n) if self.diag is None else self.diag
        if self.off is None:
            return diag.reshape(1, -1).copy()
        return np.stack([diag, np.concatenate([self.off, [0.0]])])


@dataclass
class Normal(Distribution):
    """Multivariate normal in mean/precision form.  Truncation limits on the response make NormalNormal sample
    its conditional by single-site truncated updates and log_p return -inf outside the domain, as in the reference
    (sampler.py:199-205, location_scale.py:162-188)."""

    mean: Union[str, Identity, LinearCombination, MixtureParameterVector]
    precision: Union[str, Identity, ScaledMatrix, MixtureParameterMatrix]
    domain_response_lower: np.ndarray = None
    domain_response_upper: np.ndarray = None

    def __post_init__(self):
        if isinstance(self.mean, str):
            self.mean = Identity(self.mean)
        if not isinstance(self.mean, (Identity, LinearCombination, MixtureParameterVector)):
            raise TypeError("mean expected to be one of [Identity, LinearCombination, MixtureParameterVector]")
        if isinstance(self.precision, str):
            self.precision = Identity(self.precision)
        if not isinstance(self.precision, (Identity, ScaledMatrix, MixtureParameterMatrix)):
            raise TypeError("precision expected to be one of [Identity, ScaledMatrix, MixtureParameterMatrix]")

    @property
    def _dist_params(self) -> list:
        return self.mean.get_param_list() + self.precision.get_param_list()

    # ------------------------------------------------------------------ mixture (diagonal, per-chain, ragged) form
    @property
    def is_mixture(self) -> bool:
        return isinstance(self.precision, MixtureParameterMatrix)

    def mixture_pieces(self, state, engine):
        """(x, mean, prec, count): (C, kmax) tensors of a Normal whose mean / precision are picked per element by an
        allocation (parameter.py:420-538) -- the prior of basis coefficients whose number changes under reversible
        jump.  prec is 1 and mean 0 beyond each chain's live length `count`."""
        x = state[self.response]
        if not is_chain(x) or x.shape[1] != 1:
            raise NotImplementedError("mixture Normal needs a per-chain (k, 1) response")
        if not isinstance(self.mean, MixtureParameterVector):
            raise NotImplementedError("mixture precision with a non-mixture mean")
        if self.mean.allocation == self.precision.allocation:
            # one allocation picks the mean and the precision of every element: both tables in one launch
            alloc = state[self.mean.allocation]
            if not is_chain(alloc):
                raise NotImplementedError("mixture parameters need a per-chain allocation")

            def table(par):
                return par.vector() if is_chain(par) else engine.shared(par).reshape(-1)

            mean, prec = engine.mixture_gather2(alloc.vector(), alloc.count(state), table(state[self.mean.param]), 0.0,
                                                table(state[self.precision.param]), 1.0)
            return x.vector(), mean, prec, x.count(state)
        return (x.vector(), self.mean.gather_device(state, engine, 0.0), self.precision.gather_device(state, engine, 1.0),
                x.count(state))

    # ------------------------------------------------------------------ structure
    def structure(self, state) -> NormalStructure:
        if self.is_mixture:
            raise NotImplementedError("mixture precision has no shared-matrix structure (use mixture_pieces)")
        if isinstance(self.precision, ScaledMatrix):
            M, scale_key = state[self.precision.matrix], self.precision.scalar
        else:
            M, scale_key = state[self.precision.form], None
        if is_chain(M):
            raise NotImplementedError("per-chain precision matrices")
        memo = self.__dict__.setdefault("_structure_memo", {})
        hit = memo.get(id(M))
        if hit is not None and hit.matrix is M and hit.scale_key == scale_key:
            return hit  # shared matrices are immutable: the band extraction is done once, not every log_p
        if M.shape[0] != M.shape[1]:
            raise ValueError("Matrix is not square")
        n = M.shape[0]
        bands = tridiagonal_bands(M, n)
        diag, off = bands if bands is not None else (False, False)  # False = not tridiagonal
        d = M.diagonal() if sparse.issparse(M) else np.diag(np.asarray(M))
        st = NormalStructure(n=n, matrix=M, scale_key=scale_key, diag=diag, off=off, n_pos=int(np.sum(d > 0)),
                             band=band_storage(M, n) if bands is None else None)
        memo[id(M)] = st
        return st

    def chain_and_center(self, state):
        """(x, m): the per-chain vector and the shared vector such that the residual of the Gaussian
        is +-(x - m).  Exactly one side of {response, mean} must be per chain."""
        resp = state[self.response]
        if isinstance(self.mean, Identity):
            mean = state[self.mean.form]
        else:
            mean = self.mean.predictor(state)
        if is_chain(resp) and not is_chain(mean):
            return resp, np.asarray(mean, dtype=np.float64)
        if is_chain(mean) and not is_chain(resp):
            return mean, np.asarray(resp, dtype=np.float64)
        if is_chain(mean) and is_chain(resp):
            return resp, mean  # hierarchical: both sides sampled; the callers form the residual on the device
        raise NotImplementedError("Normal with response and mean both shared on the GPU path")

    @staticmethod
    def _replicate_split(st, m):
        """sum_r (y_r - x)'M(y_r - x) = n_rep (x - ybar)'M(x - ybar) + sum_r (y_r - ybar)'M(y_r - ybar): (ybar, the constant)."""
        ybar = m.mean(axis=1, keepdims=True)
        dev = m - ybar
        return ybar, float(np.sum(dev * np.asarray(st.matrix @ dev)))

    def residual_inputs(self, state):
        """Device tensors of the per-chain state entries the residual response - mean reads."""
        keys = [self.response] + [k for k in self.mean.get_param_list() if k != self.response]
        return [state[k].data for k in keys if k in state and is_chain(state[k])]

    def residual_quad(self, state, engine, st=None, replicates=False):
        """(C,) tensor r' M r with r = response - mean, M the unscaled precision matrix: the sufficient
        statistic of NormalGamma.sample (sampler.py:276,284) and of log_p (gmrf.py:343-344)."""
        st = self.structure(state) if st is None else st
        hit = engine.quad_cache_get(self, self.residual_inputs(state))
        if hit is not None:  # the draw that produced this state computed it (NormalNormal.sample)
            return hit
        resp = state[self.response]
        dense_design = (isinstance(self.mean, LinearCombination) and not is_chain(resp)
                        and any(is_chain(state[k]) and not _is_identity(state[a], state[k].shape[0])
                                for k, a in self.mean.form.items()))
        if dense_design and (st.diag is False or st.off is not None):
            # a correlated response under a design matrix: r_c = y - fitted_c on the device, then r'Wr on W's own route
            if resp.shape[1] != 1:
                raise NotImplementedError("replicated responses")
            frozen = getattr(self.mean, "_frozen", None)
            fitted = self.mean.predictor_device(state, engine)
            r = engine.chain_lincomb(-1.0, fitted, 1.0, engine.shared(resp).reshape(-1))
            if st.diag is False and st.band is None:
                return engine.dense_quadform(engine.shared(st.matrix), r)
            quad = engine.empty(engine.n_chains)
            if st.diag is False:
                engine.band_quadform(st.n, engine.band_cache(self, st, np.zeros((st.n, 1)))["band"], r, quad)
                return quad
            cache = engine.model_cache(self, state, st, np.zeros((st.n, 1)))
            q2 = engine.empty(1, engine.n_chains)
            engine.tridiag_quadform(st.n, cache["terms_unit"], r, q2)
            return q2[0]
        if dense_design:
            if resp.shape[1] != 1:
                raise NotImplementedError("replicated responses")
            w = None if st.diag is None else engine.shared(st.diag)
            frozen = getattr(self.mean, "_frozen", None)
            if not (frozen and "fitted" in frozen):  # (fitted values at hand for this state: the residual is one pass over them)
                quad = self.mean.resid_sq_device(state, engine, engine.shared(resp).reshape(-1), w)
                if quad is not None:
                    return quad
            fitted = self.mean.predictor_device(state, engine)
            quad = engine.empty(engine.n_chains)
            engine.weighted_resid_sq(engine.shared(resp).reshape(-1), fitted, quad, w=w)
            return quad
        if st.diag is False and st.band is None:
            # dense shared precision: r'Mr through one GEMM over all chains
            x, m = self.chain_and_center(state)
            if x.shape[1] != 1:
                raise NotImplementedError("replicated per-chain side of a Normal")
            if is_chain(m):  # both sides sampled: the residual is formed on the device
                if m.shape[1] != 1:
                    raise NotImplementedError("replicated per-chain side of a Normal")
                r = engine.chain_lincomb(1.0, x.vector(), -1.0, m.vector())
                return engine.dense_quadform(engine.shared(st.matrix), r)
            n_rep, const = m.shape[1], 0.0
            if n_rep != 1:
                if not replicates:
                    raise NotImplementedError("replicated responses")
                m, const = self._replicate_split(st, m)
            memo = self.__dict__.setdefault("_dense_memo", {})
            key = id(st.matrix)
            c_host = np.ascontiguousarray(m, dtype=np.float64).reshape(-1)
            hit = memo.get(key)
            if hit is None or not np.array_equal(hit[0], c_host):
                Mm = np.asarray(st.matrix @ c_host).reshape(-1)
                hit = memo[key] = (c_host, engine.to_device(c_host) if c_host.any() else None,
                                   engine.to_device(Mm) if c_host.any() else None)
            quad = engine.dense_quadform(engine.shared(st.matrix), x.vector(), center=hit[1], M_center=hit[2])
            return quad if n_rep == 1 else quad * float(n_rep) + const
        if st.diag is False:
            x, m = self.chain_and_center(state)
            if x.shape[1] != 1:
                raise NotImplementedError("replicated per-chain side of a Normal")
            if is_chain(m):  # both sides sampled
                if m.shape[1] != 1:
                    raise NotImplementedError("replicated per-chain side of a Normal")
                r = engine.chain_lincomb(1.0, x.vector(), -1.0, m.vector())
                cache = engine.band_cache(self, st, np.zeros((st.n, 1)))
                quad = engine.empty(engine.n_chains)
                engine.band_quadform(st.n, cache["band"], r, quad)
                return quad
            n_rep, const = m.shape[1], 0.0
            if n_rep != 1:
                if not replicates:
                    raise NotImplementedError("replicated responses")
                m, const = self._replicate_split(st, m)
            cache = engine.band_cache(self, st, m)
            quad = engine.empty(engine.n_chains)
            engine.band_quadform(st.n, cache["band"], x.vector(), quad, center=cache["center"])
            return quad if n_rep == 1 else quad * float(n_rep) + const
        x, m = self.chain_and_center(state)
        if x.shape[1] != 1:
            raise NotImplementedError("replicated per-chain side of a Normal")
        if is_chain(m):
            # both sides per chain (a sampled mean under a sampled response): r_c = x_c - m_c on the device, then r'Mr
            if m.shape[1] != 1:
                raise NotImplementedError("replicated per-chain side of a Normal")
            cache = engine.model_cache(self, state, st, np.zeros((st.n, 1)))
            quad = engine.empty(1, engine.n_chains)
            mv = m.vector()
            if mv.stride(1) == 1:  # one launch: the quadratic form around the chain's own centre
                engine.set_center_chain(cache["terms_unit"], [mv], st.n)
                try:
                    engine.tridiag_quadform(st.n, cache["terms_unit"], x.vector(), quad)
                finally:
                    engine.set_center_chain(cache["terms_unit"], [None], st.n)
            else:
                r = engine.chain_lincomb(1.0, x.vector(), -1.0, mv)
                engine.tridiag_quadform(st.n, cache["terms_unit"], r, quad)
            return quad[0]
        n_rep = m.shape[1]
        if n_rep != 1 and not replicates:
            # NormalGamma's b* = r'Pr is a scalar only for one replicate (the reference's .item() raises, sampler.py:284)
            raise NotImplementedError("replicated responses")
        const = 0.0
        if n_rep != 1:
            m, const = self._replicate_split(st, m)
        cache = engine.model_cache(self, state, st, m)
        quad = engine.empty(1, engine.n_chains)
        engine.tridiag_quadform(st.n, cache["terms_unit"], x.vector(), quad)
        return quad[0] if n_rep == 1 else quad[0] * float(n_rep) + const

    # ------------------------------------------------------------------ log density
    def log_p(self, state: dict, by_observation: bool = False, engine=None, out=None, accumulate=False):
        """location_scale.py:145-167 -> gmrf.py:321-348, one value per chain."""
        if engine is None:
            raise RuntimeError("Normal.log_p needs the engine (use Model.log_p)")
        if by_observation:
            if self._column_replicates(state):
                return self._columns_log_p(state, engine)[1]  # (C, kmax): one value per live column, 0 beyond
            return self._log_p_by_observation(state, engine)   # fixed-size response: (C, n_rep)
        if self.is_mixture:
            x, mean, prec, count = self.mixture_pieces(state, engine)
            out = engine.empty(engine.n_chains) if out is None else out
            engine.diag_gauss_logpdf(x, prec, out, mean=mean, count=count, accumulate=accumulate)
            return out
        if self._column_replicates(state):
            lp = self._columns_log_p(state, engine)[0]
            if out is None:
                return lp
            out.add_(lp) if accumulate else out.copy_(lp)
            return out
        st = self.structure(state)
        quad = self.residual_quad(state, engine, st, replicates=True)
        if st.scale_key is not None and not is_chain(state[st.scale_key]):
            raise NotImplementedError("shared precision scalar")
        scale = state[st.scale_key].scalar() if st.scale_key is not None else None
        logdet = engine.matrix_logdet(st)
        resp = state[self.response]
        n_rep = 1 if is_chain(resp) else resp.shape[1]  # the log-density is summed over replicate columns (gmrf.py:346-348)
        out = engine.empty(engine.n_chains) if out is None else out
        engine.scaled_gauss_logpdf(st.n * n_rep, scale, logdet if n_rep == 1 else logdet * float(n_rep), quad, out,
                                   accumulate=accumulate)
        if (self.domain_response_lower is not None or self.domain_response_upper is not None) and is_chain(resp):
            # location_scale.py:162-164: -inf for a response outside its domain (the un-normalised density otherwise)
            lo, hi = self._domain_device(engine, st.n)
            engine.domain_penalty(resp.vector(), out, lower=lo, upper=hi)
        return out

    def log_p_piece(self, state, engine, dry=False):
        """This distribution's term of Model.log_p as a piece of omc_log_post_sum, or None where log_p does more than the
        scaled Gaussian formula (mixtures, replicate columns, domain limits on a per-chain response).  dry: only say whether."""
        if type(self) is not Normal or self.is_mixture or self._column_replicates(state):
            return None
        resp = state[self.response]
        if (self.domain_response_lower is not None or self.domain_response_upper is not None) and is_chain(resp):
            return None
        st = self.structure(state)
        if st.scale_key is not None and not is_chain(state[st.scale_key]):
            return None
        if dry:
            return True
        quad = self.residual_quad(state, engine, st, replicates=True)
        scale = state[st.scale_key].scalar() if st.scale_key is not None else None
        n_rep = 1 if is_chain(resp) else resp.shape[1]
        return ("gauss", st.n * n_rep, scale, engine.matrix_logdet(st), float(n_rep), quad)

    def _domain_device(self, engine, n):
        memo = self.__dict__.setdefault("_domain_memo", {})
        if n not in memo:
            def vec(v):
                if v is None:
                    return None
                return engine.to_device(np.broadcast_to(np.asarray(v, dtype=np.float64).reshape(-1, 1), (n, 1)).reshape(-1).copy())
            memo[n] = (vec(self.domain_response_lower), vec(self.domain_response_upper))
        return memo[n]

    # ------------------------------------------------------------------ a variable number of replicate columns
    # The associated parameter of a reversible jump (theta (d, k): one column per knot, k per chain) under a Normal prior with
    # shared mean and precision: the density is summed over the live columns (gmrf.py:346-348), a birth draws one new column
    # from the prior and scores the LAST current one (reversible_jump.py:130-132,143).  d is small: plain tensor algebra.
    def _log_p_by_observation(self, state, engine, values=None):
        """log_p(state, by_observation=True) for a fixed-size response (location_scale.py:145-167 -> gmrf.py:321-348 with
        by_observation): one log density per replicate column and chain, (C, n_rep).  Either side may be per chain; the
        precision is the distribution's shared matrix times its (per-chain or shared) scalar.  A diagnostic read-out, not a step of
        the sampler loop: dense tensor expressions on the device."""
        import torch

        if self.is_mixture:
            raise NotImplementedError("by_observation of a mixture Normal")
        st = self.structure(state)
        memo = self.__dict__.setdefault("_byobs_memo", {})
        hit = memo.get(id(st.matrix))
        if hit is None or hit[0] is not st.matrix:
            Md = st.matrix.toarray() if sparse.issparse(st.matrix) else np.array(st.matrix, dtype=np.float64, ndmin=2)
            sign, logdet = np.linalg.slogdet(Md)
            if sign <= 0:
                raise np.linalg.LinAlgError("Matrix is not positive definite")  # gmrf.py:518 through the Cholesky factor
            hit = memo[id(st.matrix)] = (st.matrix, engine.to_device(Md), float(logdet))
        Md, logdetM = hit[1], hit[2]
        resp = state[self.response] if values is None else values
        d = st.n
        X = resp.data if is_chain(resp) else engine.to_device(np.asarray(resp, dtype=np.float64).reshape(d, -1)).unsqueeze(0)
        if isinstance(self.mean, Identity):
            mv = state[self.mean.form]
            m = mv.data.reshape(mv.data.shape[0], d, -1) if is_chain(mv) else engine.to_device(np.asarray(mv, dtype=np.float64).reshape(1, d, -1))
        elif any(is_chain(state[k]) for k in self.mean.get_param_list() if k in state):
            m = self.mean.predictor_device(state, engine).reshape(engine.n_chains, d, 1)
        else:
            m = engine.to_device(np.asarray(self.mean.predictor(state), dtype=np.float64).reshape(1, d, -1))
        r = X - m                                                   # (C or 1, d, n_rep)
        q = (r * torch.matmul(Md.unsqueeze(0), r)).sum(dim=1)       # (C or 1, n_rep)
        if st.scale_key is not None:
            sv = state[st.scale_key]
            sc = sv.scalar().reshape(-1, 1) if is_chain(sv) else engine.full((1, 1), float(np.asarray(sv).item()))
        else:
            sc = engine.full((1, 1), 1.0)
        per = 0.5 * (d * torch.log(sc) + logdetM - d * float(np.log(2.0 * np.pi))) - 0.5 * sc * q
        return per.expand(engine.n_chains, per.shape[1]).contiguous()

    def _column_replicates(self, state) -> bool:
        x = state.get(self.response)  # (a prior draw is asked for when the state has no value yet)
        return is_chain(x) and x.ragged is not None and x.ragged[1] == 1 and not self.is_mixture

    def _columns_pieces(self, state, engine):
        """(mu (d,) device, Q (d, d) device, log det Q, host Cholesky factor) of the shared prior; memoised on the matrix."""
        if not isinstance(self.precision, Identity) or is_chain(state[self.precision.form]):
            raise NotImplementedError("replicate columns need a shared matrix precision")
        mean = self.mean.predictor(state)
        if is_chain(mean):
            raise NotImplementedError("replicate columns need a shared mean")
        Q = state[self.precision.form]
        memo = self.__dict__.setdefault("_columns_memo", {})
        hit = memo.get(id(Q))
        if hit is None or hit[0] is not Q:
            Qd = Q.toarray() if sparse.issparse(Q) else np.array(Q, dtype=np.float64, ndmin=2)
            L = np.linalg.cholesky(Qd)
            hit = memo[id(Q)] = (Q, engine.to_device(Qd), 2.0 * float(np.sum(np.log(np.diag(L)))),
                                 engine.to_device(np.linalg.inv(L)))
        mu = engine.to_device(np.asarray(mean, dtype=np.float64).reshape(-1).copy())
        return mu, hit[1], hit[2], hit[3]

    def _columns_values(self, state, engine):
        """(C, d, kmax) values the Gaussian density is evaluated at, the live-column mask and what is subtracted per live column
        (0 here; sum of logs for the log-normal)."""
        import torch

        x = state[self.response]
        k = torch.arange(x.data.shape[2], device=x.data.device).reshape(1, -1)
        live = k < x.count(state).reshape(-1, 1)
        return x.data, live, None

    def _columns_log_p(self, state, engine):
        """(log_p summed over the live columns (C,), per-column log densities (C, kmax), live mask)."""
        import torch

        mu, Qd, logdet, _ = self._columns_pieces(state, engine)
        v, live, minus = self._columns_values(state, engine)
        d = v.shape[1]
        r = v - mu.reshape(1, d, 1)
        q = (r * torch.einsum("ij,cjk->cik", Qd, r)).sum(dim=1)
        per = 0.5 * logdet - 0.5 * d * float(np.log(2.0 * np.pi)) - 0.5 * q
        if minus is not None:
            per = per - minus
        per = torch.where(live, per, torch.zeros_like(per))
        return per.sum(dim=1), per, live

    def log_p_last(self, state: dict, engine):
        """log_p(state, by_observation=True)[-1] of every chain (reversible_jump.py:132,143): (C,) tensor."""
        _, per, _ = self._columns_log_p(state, engine)
        last = (state[self.response].count(state) - 1).clamp(min=0).to(per.device).long().reshape(-1, 1)
        return per.gather(1, last).reshape(-1)

    def _column_draw(self, state, engine, draw_index, sub, inject):
        """mu + L^-T z for every chain (gmrf.py:29-61): (C, d) tensor; `inject` (C, d) standard normals."""
        mu, _, _, Linv = self._columns_pieces(state, engine)
        d = mu.numel()
        if inject is None:
            z = engine.fill_normal(d, draw_index=int(draw_index) + ((int(sub) & 0xF) << 44))
        else:
            z = inject.reshape(engine.n_chains, d)
        return mu.reshape(1, d) + z @ Linv  # rows: (L^-T z)' = z' L^-1

    def rvs(self, state: dict, n: int = 1, engine=None, draw_index=0, sub=0, inject=None):
        """One draw per chain from N(mean, (scale * M)^-1) (location_scale.py:252-272 -> gmrf.py:29-61),
        used by MCMC when the state has no initial value for a sampled parameter (mcmc.py:78-80) and by a birth move for the
        new column of an associated parameter (reversible_jump.py:130).  With domain limits: gmrf.sample_truncated_normal for
        one replicate (gmrf.py:64-164), i.e. rejection on the whole vector -- every chain redraws until all of its elements lie
        inside; `inject` is then (attempts, C, d): the standard normals of attempt 0, 1, ... of every chain."""
        if engine is None:
            raise RuntimeError("Normal.rvs needs the engine")
        if n != 1:
            raise NotImplementedError("replicated prior draws")
        limited = self.domain_response_lower is not None or self.domain_response_upper is not None
        columns = self._column_replicates(state)
        if columns:
            d = self._columns_pieces(state, engine)[0].numel()

            def draw(attempt, z):
                return self._column_draw(state, engine, int(draw_index) + (attempt << 24), sub, z)
        else:
            if sub:
                raise NotImplementedError("sub-streams for a fixed-size prior draw")
            d = self.structure(state).n

            def draw(attempt, z):
                return self._fixed_draw(state, engine, int(draw_index) + (attempt << 24), z)
        from openmcmc_amd.chains import ChainArray

        if not limited:
            x = draw(0, None if inject is None else engine.to_device(inject).reshape(engine.n_chains, d))
            return ChainArray(x.unsqueeze(2)) if columns else ChainArray(x)
        import torch

        lo = np.full(d, -np.inf) if self.domain_response_lower is None else np.broadcast_to(
            np.asarray(self.domain_response_lower, dtype=np.float64).reshape(-1), (d,))
        hi = np.full(d, np.inf) if self.domain_response_upper is None else np.broadcast_to(
            np.asarray(self.domain_response_upper, dtype=np.float64).reshape(-1), (d,))
        if np.any(lo >= hi):
            raise ValueError("Error lower bound must be strictly less than upper bound")  # gmrf.py:149-150
        lo_d, hi_d = engine.to_device(lo.copy()), engine.to_device(hi.copy())
        tape = None if inject is None else engine.to_device(inject).reshape(-1, engine.n_chains, d)
        x = engine.empty(engine.n_chains, d)
        bad = torch.ones(engine.n_chains, dtype=torch.bool, device=x.device)
        pen = engine.empty(engine.n_chains)
        # attempt a draws from stream draw_index + (a << 24): 16 bits of attempts below bit 40, where MCMC's prior-draw field
        # starts ((1 << 40) + position; the sub-stream field of _column_draw sits at bit 44) -- one more bit and a late
        # attempt of one draw would replay another draw's early attempts
        attempts = 65536 if tape is None else tape.shape[0]
        for a in range(attempts):
            cand = draw(a, None if tape is None else tape[a].contiguous()).contiguous()
            pen.zero_()
            engine.domain_penalty(cand, pen, lo_d, hi_d)  # -inf for a chain with an element outside
            ok = (pen == 0) & bad                          # every chain keeps the first attempt that lies inside
            engine.chain_select(ok.to(torch.int32), cand, x)
            bad &= ~ok
            if not bool(bad.any().item()):
                break
        else:
            raise RuntimeError("truncated prior draw: some chains found no vector inside the domain limits"
                               + ("" if tape is None else " on the injected tape"))
        return ChainArray(x.unsqueeze(2)) if columns else ChainArray(x)

    def _fixed_draw(self, state, engine, draw_index, z):
        """mu + L^-T z for every chain on the structure's own route (gmrf.py:29-61); z: (C, n) injected standard normals."""
        st = self.structure(state)
        mean = self.mean.predictor(state)
        if is_chain(mean):
            raise NotImplementedError("per-chain mean in a prior draw")
        mean = np.asarray(mean, dtype=np.float64).reshape(-1)
        scale = None
        if st.scale_key is not None:
            sv = state[st.scale_key]
            scale = sv.scalar() if is_chain(sv) else engine.full((engine.n_chains,), float(np.asarray(sv).item()))
        x = engine.empty(engine.n_chains, st.n)
        if st.diag is False:  # dense precision: x = Q^-1 (Q mean) + L^-T z
            M = engine.shared(st.matrix)
            rhs = engine.design_rhs(M, engine.to_device(mean)) if mean.any() else None
            engine.dense_sample_canonical(st.n, [{"mat": M, "rhs": rhs, "scale": scale}], x, z=z, draw_index=draw_index)
        else:
            cache = engine.model_cache(self, state, st, mean)
            engine.tridiag_sample_canonical(st.n, [{"diag": cache["diag"], "off": cache["off"], "rhs": cache["rhs"],
                                                    "scale": scale}], x, z=z, draw_index=draw_index)
        return x

    def constant_hessian(self, param: str) -> bool:
        """Is this distribution Gaussian in `param` (Hessian independent of it: branches (i) and (ii) of
        location_scale.py:190-250)?  Then ManifoldMALA may take the dense route built on grad_terms."""
        return param == self.response or (param in self.mean.get_grad_param_list() and param not in self.precision.get_grad_param_list())

    def grad_terms(self, state: dict, param: str, engine):
        """This distribution's share of the gradient and Hessian w.r.t. a per-chain (p, 1) parameter, in the form the
        dense route consumes: (grad (C, p) tensor, [{"mat": shared host matrix or None = identity, "scale": (C,) tensor or
        None}], diag (C, p) per-chain diagonal or None).  None when the distribution does not involve `param`.  Covers
        branches (i) and (ii) of location_scale.py:190-250 for constant Hessians (Gaussian in `param`)."""
        if param not in self.param_list:
            return None
        x = state[param]
        if not is_chain(x) or x.shape[1] != 1:
            raise NotImplementedError("grad_terms needs a per-chain (p, 1) parameter")
        if param == self.response:
            if self.is_mixture:
                xv, mean, prec, count = self.mixture_pieces(state, engine)
                if count is not None:
                    raise NotImplementedError("variable-size parameter on the dense route")
                return engine.diag_gauss_grad(xv, prec, mean=mean), [], prec
            st = self.structure(state)
            mu = self.mean.predictor(state)
            if is_chain(mu):
                raise NotImplementedError("per-chain prior mean")
            Md = engine.shared(st.matrix)
            r = x.vector() - engine.to_device(np.asarray(mu, dtype=np.float64).reshape(1, -1))
            g = -engine.design_predict(Md, r.contiguous())           # M symmetric: rows of r times M
            scale = state[st.scale_key].scalar() if st.scale_key is not None else None
            if scale is not None:
                g = g * scale.unsqueeze(1)
            return g, [{"mat": st.matrix, "scale": scale}], None
        grad, hess = self.grad_log_p(state, param, hessian_required=True, engine=engine)  # branch (ii)
        if isinstance(hess, ScaledHessian):
            return grad.vector(), [{"mat": hess.matrix, "scale": hess.scale}], None
        if hasattr(hess, "shape") and not hasattr(hess, "data_ptr"):
            return grad.vector(), [{"mat": hess, "scale": None}], None
        raise NotImplementedError("parameter-dependent Hessian on the dense route")

    def grad_log_p_diag(self, state: dict, param: str, engine):
        """(grad, hdiag), each (C, kmax), when the Hessian w.r.t. `param` is diagonal per chain: the response of a
        mixture Normal (grad = -prec (x - mean), Hessian diag(prec); location_scale.py:222-226 with parameter.py:501).
        None when this distribution does not depend on `param`."""
        if param != self.response:
            if param in self.param_list:
                raise NotImplementedError("diagonal-Hessian gradient w.r.t. a mean / precision parameter")
            return None
        if not self.is_mixture:
            raise NotImplementedError("diagonal-Hessian gradient of a Normal with a matrix precision")
        x, mean, prec, count = self.mixture_pieces(state, engine)
        return engine.diag_gauss_grad(x, prec, mean=mean, count=count), prec

    def grad_log_p(self, state: dict, param: str, hessian_required: bool = True, engine=None):
        """Gradient of +log p and Hessian of -log p w.r.t. a per-chain parameter (location_scale.py:190-250), the
        reference's three branches:
          (i)   `param` is the response and the precision a shared matrix: grad_c = -Q (x_c - mu) as one GEMM over all
                chains, Hessian = Q (shared host matrix);
          (ii)  `param` enters the mean only (Identity, or LinearCombination with a shared design A):
                grad_c = s_c A' W sum_rep (y - fitted_c) as one GEMM, Hessian = n_rep A' W A times the per-chain scalar
                s_c, returned as ScaledHessian(scale (C,), matrix);
          (iii) anything else (e.g. the precision's scalar): central differences of log_p (distribution.py:124-198)."""
        if engine is None:
            raise RuntimeError("Normal.grad_log_p needs the engine")
        from openmcmc_amd.chains import ChainArray

        x = state.get(param)
        if param == self.response and isinstance(self.precision, Identity) and is_chain(x):
            Q, mu = state[self.precision.form], self.mean.predictor(state)
            if is_chain(Q) or is_chain(mu):
                raise NotImplementedError("grad_log_p of the response needs a shared mean and precision")
            dQ = engine.shared(Q)
            r = x.vector() - engine.to_device(np.asarray(mu, dtype=np.float64).reshape(1, -1))
            grad = ChainArray(-engine.design_predict(dQ, r))  # Q symmetric: Q r for every chain
            return (grad, Q) if hessian_required else grad
        in_mean = param in self.mean.get_grad_param_list() and param not in self.precision.get_grad_param_list()
        if in_mean and param != self.response and is_chain(x) and not self.is_mixture:
            st = self.structure(state)
            general_w = st.diag is False or st.off is not None  # a correlated response: W r through (W A)' (location_scale.py:234-242 with any Q)
            resp = state[self.response]
            if is_chain(resp) or x.shape[1] != 1:
                raise NotImplementedError("gradient through the mean needs a shared response and a (p, 1) parameter")
            A = None if isinstance(self.mean, Identity) else state[self.mean.form[param]]
            if is_chain(A):
                raise NotImplementedError("gradient through a per-chain design matrix")
            n_rep = resp.shape[1]
            w = np.ones(st.n) if (st.diag is None or general_w) else np.asarray(st.diag, dtype=np.float64)
            fitted = x.vector() if isinstance(self.mean, Identity) else self.mean.predictor_device(state, engine)
            ysum = engine.to_device(np.asarray(resp, dtype=np.float64).sum(axis=1).reshape(1, -1))
            r = (ysum - float(n_rep) * fitted) * engine.to_device(w.reshape(1, -1))   # W sum_rep (y - fitted) (W applied below if general)
            memo = self.__dict__.setdefault("_grad_memo", {})
            if general_w:
                key = ("general", id(A), id(st.matrix))
                hit = memo.get(key)
                if hit is None or hit[0] is not A:
                    Wh = st.matrix
                    if A is None:
                        WA = Wh.toarray() if sparse.issparse(Wh) else np.asarray(Wh, dtype=np.float64)  # G = I: g = W r, H = n_rep W
                        Hh = float(n_rep) * WA
                    else:
                        Ad = A.toarray() if sparse.issparse(A) else np.asarray(A, dtype=np.float64)
                        WA = np.asarray(Wh @ Ad, dtype=np.float64)
                        Hh = float(n_rep) * (Ad.T @ WA)
                    hit = memo[key] = (A, engine.to_device(np.ascontiguousarray(WA.T)), Hh)
                g = engine.design_predict(hit[1], r.contiguous())  # (W A)' r_c for every chain: (C, p)
                H = hit[2]
            elif A is None:
                g, H = r, (sparse.diags(w) * float(n_rep)).tocsc()
            else:
                hit = memo.get(id(A))
                if hit is None or hit[0] is not A:
                    Ad = A.toarray() if sparse.issparse(A) else np.asarray(A, dtype=np.float64)
                    hit = memo[id(A)] = (A, engine.to_device(np.ascontiguousarray(Ad.T)), float(n_rep) * (Ad.T * w) @ Ad)
                g = engine.design_predict(hit[1], r.contiguous())  # rows of r times A: (C, p)
                H = hit[2]
            scale = state[st.scale_key].scalar() if st.scale_key is not None else None
            if scale is not None:
                g = g * scale.unsqueeze(1)
            grad = ChainArray(g)
            if not hessian_required:
                return grad
            return grad, (H if scale is None else ScaledHessian(scale=scale, matrix=H))
        return Distribution.grad_log_p(self, state, param, hessian_required=hessian_required, engine=engine)


@dataclass
class NullDistribution(Normal):
    """Null distribution of the reference's reversible-jump prior-recovery tests (location_scale.py:63-124): log-density
    0, zero gradient and Hessian, no draws.  Mean / precision parameters are kept so that `param_list` and
    `model.response` predictors behave as for a Normal."""

    def log_p(self, state: dict, by_observation: bool = False, engine=None, out=None, accumulate=False):
        if out is None:
            return 0.0
        if not accumulate:
            out.zero_()
        return out

    def grad_log_p_diag(self, state: dict, param: str, engine):
        return None  # contributes nothing

    def grad_terms(self, state: dict, param: str, engine):
        return None

    def constant_hessian(self, param: str) -> bool:
        return True  # zero

    def grad_log_p(self, state: dict, param: str, hessian_required: bool = True, engine=None):
        raise NotImplementedError("NullDistribution.grad_log_p: use grad_log_p_diag / grad_terms (zero contribution)")

    def rvs(self, state: dict, n: int = 1, engine=None, draw_index=0):
        return None


@dataclass
class LogNormal(Normal):
    """Multivariate log-normal (location_scale.py:275-418): log(response) ~ Normal(mean, precision): log_p, rvs and
    the analytic gradient / Hessian."""

    def log_p(self, state: dict, by_observation: bool = False, engine=None, out=None, accumulate=False):
        """location_scale.py:279-300: the Normal log-density at log(response) minus sum log(response)."""
        if engine is None:
            raise RuntimeError("LogNormal.log_p needs the engine (use Model.log_p)")
        if by_observation:
            if self._column_replicates(state):
                return self._columns_log_p(state, engine)[1]
            # fixed-size response (location_scale.py:293-299): the Normal's per-replicate log density at log(response) minus the
            # column sums of log(response)
            import torch

            from openmcmc_amd.chains import ChainArray as _CA

            rv = state[self.response]
            d = self.structure(state).n
            if is_chain(rv):
                lg = torch.log(rv.data)
                per = self._log_p_by_observation(state, engine, values=_CA(lg))
                return per - lg.sum(dim=1)
            lgh = np.log(np.asarray(rv, dtype=np.float64).reshape(d, -1))
            per = self._log_p_by_observation(state, engine, values=lgh)
            return per - engine.to_device(lgh.sum(axis=0).reshape(1, -1))
        from openmcmc_amd.chains import ChainArray

        resp = state[self.response]
        if self._column_replicates(state):
            lp = self._columns_log_p(state, engine)[0]
            if out is None:
                return lp
            out.add_(lp) if accumulate else out.copy_(lp)
            return out
        logged = dict(state)
        if is_chain(resp):
            if resp.shape[1] != 1:
                raise NotImplementedError("replicated per-chain log-normal response")
            lx, sumlog = engine.log_transform(resp.vector())
            logged[self.response] = ChainArray(lx)
        else:
            memo = self.__dict__.setdefault("_log_memo", {})
            hit = memo.get(id(resp))
            if hit is None or hit[0] is not resp:
                arr = np.asarray(resp, dtype=np.float64)
                hit = memo[id(resp)] = (resp, np.log(arr), float(np.sum(np.log(arr))))
            logged[self.response], sumlog = hit[1], hit[2]
        out = Normal.log_p(self, logged, engine=engine, out=out, accumulate=accumulate)
        out -= sumlog
        return out

    def rvs(self, state: dict, n: int = 1, engine=None, draw_index=0, sub=0, inject=None):
        """location_scale.py:404-418: exp of the Normal draw."""
        from openmcmc_amd.chains import ChainArray

        draw = Normal.rvs(self, state, n=n, engine=engine, draw_index=draw_index, sub=sub, inject=inject)
        return ChainArray(draw.data.exp())

    def _columns_values(self, state, engine):
        """The Gaussian part is evaluated at log(x) and every live column pays its sum of logs (location_scale.py:296-299)."""
        import torch

        x, live, _ = Normal._columns_values(self, state, engine)
        lx = torch.log(torch.where(live.unsqueeze(1), x, torch.ones_like(x)))  # padding is not data
        return lx, live, lx.sum(dim=1)

    def constant_hessian(self, param: str) -> bool:
        return param != self.response and Normal.constant_hessian(self, param)  # as a response: H depends on x

    def grad_terms(self, state: dict, param: str, engine):
        if param == self.response:
            raise NotImplementedError("log-normal response: parameter-dependent Hessian (generic ManifoldMALA route)")
        return Normal.grad_terms(self, state, param, engine)

    def grad_log_p(self, state: dict, param: str, hessian_required: bool = True, engine=None):
        """location_scale.py:302-402, the reference's three branches for a per-chain parameter:
          (i)   `param` is the response x (shared mean mu, shared matrix precision Q): with r = log x - mu,
                grad = -(1 + Q r) / x and (negative second derivatives) H = diag(1/x) Q diag(1/x) - diag((1 + Q r) / x^2),
                one (d, d) matrix per chain;
          (ii)  `param` enters the mean only: the Normal's branch (ii) at log(response);
          (iii) anything else: central differences of log_p."""
        if engine is None:
            raise RuntimeError("LogNormal.grad_log_p needs the engine")
        import torch

        from openmcmc_amd.chains import ChainArray

        x = state.get(param)
        if param == self.response and isinstance(self.precision, Identity) and is_chain(x):
            Q, mu = state[self.precision.form], self.mean.predictor(state)
            if is_chain(Q) or is_chain(mu) or x.shape[1] != 1:
                raise NotImplementedError("grad_log_p of a log-normal response needs a shared mean and precision and a (d, 1) response")
            dQ = engine.shared(Q)
            xv = x.vector()
            lx, _ = engine.log_transform(xv)
            r = lx - engine.to_device(np.asarray(mu, dtype=np.float64).reshape(1, -1))
            inv = 1.0 / xv
            t = inv * (1.0 + engine.design_predict(dQ, r))  # Q symmetric: Q r for every chain
            grad = ChainArray(-t)
            if not hessian_required:
                return grad
            return grad, inv.unsqueeze(2) * dQ.unsqueeze(0) * inv.unsqueeze(1) - torch.diag_embed(inv * t)
        in_mean = param in self.mean.get_grad_param_list() and param not in self.precision.get_grad_param_list()
        resp = state[self.response]
        if in_mean and param != self.response and is_chain(x) and not self.is_mixture and not is_chain(resp):
            logged = dict(state)
            logged[self.response] = np.log(np.asarray(resp, dtype=np.float64))
            return Normal.grad_log_p(self, logged, param, hessian_required=hessian_required, engine=engine)
        return Distribution.grad_log_p(self, state, param, hessian_required=hessian_required, engine=engine)
