"""Conjugate Gibbs samplers on chain-batched state (reference sampler/sampler.py:37-288).

Same constructor signature and plugin surface as the reference (`param`, `model`,
`max_variable_size`, `sample`, `init_store`, `store`); `sample(state)` replaces state[param] for
ALL chains with one launch of the HIP library.  A sampler is attached to the chains it serves with
`bind(engine, position, n_samplers)` (MCMC does this), which also fixes its random stream:
sweep t of sampler j uses draw index t * n_samplers + j, so results do not depend on fusion or on
how chains are sharded over GPUs.
"""

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Union

import numpy as np

from openmcmc_amd.chains import ChainArray, is_chain
from openmcmc_amd.distribution.location_scale import Normal
from openmcmc_amd.model import Model
from openmcmc_amd.parameter import (Identity, LinearCombination, MixtureParameterMatrix, MixtureParameterVector,
                                     ScaledMatrix, _is_identity)


@dataclass
class MCMCSampler(ABC):
    """Base class (sampler.py:37-118)."""

    param: str
    model: Model
    max_variable_size: Union[int, tuple, None] = None

    def __post_init__(self):
        self.model = self.model.conditional(self.param)
        self._init_runtime()

    def _init_runtime(self):
        self.engine = None
        self._position, self._n_samplers, self._sweep = 0, 1, 0
        self._plan = None
        self.inject = None  # test hook: callable(sampler, sweep) -> injected draws (device tensor)

    def bind(self, engine, position=0, n_samplers=1):
        self.engine, self._position, self._n_samplers = engine, position, n_samplers
        self._plan = None
        return self

    def _draw_index(self):
        return self._sweep * self._n_samplers + self._position

    def _need_engine(self):
        if self.engine is None:
            raise RuntimeError(f"{type(self).__name__}('{self.param}') is not bound to an Engine: "
                               "construct it through MCMC or call sampler.bind(engine)")
        return self.engine

    @abstractmethod
    def sample(self, current_state: dict) -> dict:
        """Replace state[param] by a new draw for every chain."""

    def init_store(self, current_state: dict, store: dict, n_iterations: int) -> dict:
        """store[param]: (n_iterations, C, size) device tensor, NaN-filled (sampler.py:69-87; the
        reference's (size, n_iterations) per chain, iteration-major so a draw can be written into its
        slab directly).  A variable-size parameter gets max_variable_size rows, NaN beyond its live length."""
        eng = self._need_engine()
        if isinstance(self.max_variable_size, tuple):
            # matrix-valued variable-size parameter (sampler.py:81-82): (max_rows, max_cols, n_iterations) per chain
            if len(self.max_variable_size) != 2:
                raise ValueError("max_variable_size as a tuple wants (rows, columns)")
            store[self.param] = eng.full((n_iterations, eng.n_chains) + tuple(int(v) for v in self.max_variable_size), float("nan"))
            return store
        size = current_state[self.param].size if self.max_variable_size is None else int(self.max_variable_size)
        store[self.param] = eng.full((n_iterations, eng.n_chains, size), float("nan"))
        return store

    def store(self, current_state: dict, store: dict, iteration: int) -> dict:
        """sampler.py:89-118."""
        value = current_state[self.param]
        if isinstance(self.max_variable_size, tuple):
            # sampler.py:105-111: the current (p, q) block goes to the top-left corner, the NaN fill stays elsewhere
            import torch

            block = value.data
            slab = store[self.param][iteration]
            if block.shape[1] > slab.shape[1] or block.shape[2] > slab.shape[2]:
                raise ValueError("parameter larger than max_variable_size")
            if value.ragged is not None:
                axis = value.ragged[1]
                live = torch.arange(block.shape[1 + axis], device=block.device).reshape((1, -1, 1) if axis == 0 else (1, 1, -1)) \
                    < value.count(current_state).reshape(-1, 1, 1)
                block = torch.where(live, block, torch.full_like(block, float("nan")))
            slab[:, : block.shape[1], : block.shape[2]] = block
            return store
        flat = value.data.reshape(self.engine.n_chains, -1)
        slab = store[self.param][iteration]
        if flat.data_ptr() == slab.data_ptr() and flat.shape == slab.shape and flat.stride() == slab.stride():
            return store  # the draw was written into its slab directly (MCMC.run_mcmc hands the slab to NormalNormal.sample)
        if value.ragged is None:
            # big states through the library's copy kernel (the runtime's device-to-device memcpy moves 82 MB at 0.26 TB/s);
            # small ones stay with the tensor copy, whose host-side cost is a third of a ctypes call (cfg5's loop is bound
            # by the host: 0.83 against 0.87 ms per sweep)
            if flat.numel() >= (1 << 23) and flat.stride(1) == 1 and slab.stride(1) == 1 and flat.shape == slab.shape:
                self.engine.chain_copy(flat, slab)
            else:
                slab.copy_(flat)
        else:  # live entries, NaN beyond (the reference leaves its NaN fill there, sampler.py:116)
            self.engine.store_ragged(flat if flat.stride(1) == 1 else flat.contiguous(), value.count(current_state), slab)
        return store


def _as_chain_scalar(engine, state, key):
    """state[key] as a per-chain scalar; a shared host scalar is broadcast once and written back."""
    v = state[key]
    if not is_chain(v):
        v = ChainArray(engine.full((engine.n_chains, 1, 1), float(np.asarray(v).item())))
        state[key] = v
    return v


@dataclass
class NormalNormal(MCMCSampler):
    """Normal-Normal conjugate update (sampler.py:121-207): x ~ N(Q^{-1} b, Q^{-1}),
    Q = P + sum_k A_k' W_k A_k, b = P m + sum_k A_k' W_k (y_k - d_k).

    GPU routes, chosen from the structure of the terms (per-chain scalar x shared matrix each):
      tridiagonal  every matrix tridiagonal (example 4 as written and its sparse route; omc_tridiag_sample_canonical,
                   fused with the NormalGamma updates by MCMC when the sampler list allows);
      band         banded but wider (RW2, seasonal, lattice GMRFs; omc_band_sample_canonical);
      dense        anything else, incl. a regression likelihood through its Gram matrix (example 3; blocked batched Cholesky);
      ragged       a variable-size parameter with a mixture prior and a per-chain design matrix (reversible jump);
    a prior with domain limits switches the tridiagonal and dense routes to the truncated single-site scan."""

    def __post_init__(self):
        super().__post_init__()
        self._is_response = {key: key == self.param for key in self.model.keys()}

    # -- plan: the device-side description of Q and b, built once --
    def _build_plan(self, state):
        eng = self._need_engine()
        prior = self.model[self.param]
        if not isinstance(prior, Normal):
            raise TypeError("NormalNormal needs a Normal prior on the parameter")
        n = state[self.param].shape[0]
        mixture_prior = prior.is_mixture
        if mixture_prior:
            if prior.domain_response_lower is not None or prior.domain_response_upper is not None:
                raise NotImplementedError("truncated mixture prior")
            per_chain_design = any(k != self.param and isinstance(d.mean, LinearCombination)
                                   and is_chain(state[d.mean.form[self.param]]) for k, d in self.model.items())
            if per_chain_design or state[self.param].ragged is not None:
                return self._ragged_plan(state, n)
        pieces = []  # one per distribution: what Q and b receive from it
        for key, dist in self.model.items():
            if mixture_prior and key == self.param:
                continue  # its diagonal precision and rhs are per chain: added at sample time (diag_chain, rhs_chain)
            if not isinstance(dist, Normal):
                raise TypeError("NormalNormal handles Normal distributions only")
            st = dist.structure(state)
            piece = {"key": key, "dist": dist, "st": st, "design": None, "offset": False}
            if self._is_response[key]:
                if _mean_is_chain(dist.mean, state):
                    # a prior mean that is itself sampled (hierarchical model): b_c += s_c M m_c every sweep on the
                    # device (sampler.py:181-183); nothing shared goes into the plan for this term
                    piece["center"], piece["chain_vec"] = np.zeros((n, 1)), ("mean", dist)
                else:
                    mean = dist.mean.predictor(state)  # sampler.py:183: b += Q_rsp @ mean
                    piece["center"] = np.asarray(mean, dtype=np.float64)
            else:
                # likelihood: the parameter enters the mean linearly (sampler.py:185-192)
                if isinstance(dist.mean, Identity):
                    if dist.mean.form != self.param:
                        raise TypeError("mean of the response does not depend on the parameter")
                    rest = 0.0
                else:
                    A = state[dist.mean.form[self.param]]
                    if not _is_identity(A, n):
                        piece["design"] = A
                    if dist.mean.has_chain_terms(state, exclude=self.param):
                        # per-chain offset d_c in y - d_c (e.g. a basis expansion B_c beta_c next to the GMRF):
                        # evaluated on the device every sweep and fed to the solve as a per-chain rhs
                        piece["offset"], rest = True, 0.0
                    else:
                        rest = dist.mean.predictor_conditional(state, term_to_exclude=self.param)
                y = state[key]
                if is_chain(y):
                    # a response that is itself sampled (the parameter is the mean of another sampled vector):
                    # b_c += s_c W (y_c - d) with y_c per chain (sampler.py:185-192)
                    if piece["design"] is not None or piece["offset"] or y.shape[1] != 1:
                        raise NotImplementedError("per-chain response under a design matrix, with per-chain offsets, or replicated")
                    piece["center"] = -np.asarray(rest, dtype=np.float64) * np.ones((n, 1))
                    piece["chain_vec"] = ("response", key)
                    pieces.append(piece)
                    continue
                y = np.asarray(y, dtype=np.float64)
                n_rep = y.shape[1]
                if n_rep != 1:
                    # replicated draws of the response (sampler.py:165-167, 187-188): b += W sum_r y_r and
                    # Q += n_rep * W  ==  one observation ybar under the precision n_rep * W
                    if not isinstance(dist.mean, Identity):
                        # (the reference fails here too: b (p, 1) += A'W(y - d) of shape (p, n_rep), sampler.py:192)
                        raise NotImplementedError("replicated response under a LinearCombination mean")
                    y = y.mean(axis=1, keepdims=True)
                    piece["st"] = st = self._replicated_structure(key, st, n_rep)
                    piece["replicated"] = True
                piece["center"] = y - rest
            pieces.append(piece)
        tridiagonal = all(pc["design"] is None and pc["st"].diag is not False and pc["st"].n == n for pc in pieces)
        banded = all(pc["design"] is None and pc["st"].n == n and (pc["st"].diag is not False or pc["st"].band is not None)
                     for pc in pieces)
        if mixture_prior:
            plan = self._dense_plan(state, n, pieces)
            plan["mixture_prior"] = prior
        elif tridiagonal and n <= self.TRIDIAG_WG_MAX:
            plan = self._tridiag_plan(state, n, pieces)
        elif banded:  # (long tridiagonal chains too: beyond one workgroup per chain the segmented lane kernels of the band
            #            route, w = 1, are a hundred times faster than the one-lane-per-chain fallback of the tridiagonal one)
            plan = self._band_plan(state, n, pieces)
        else:
            plan = self._dense_plan(state, n, pieces)
        plan["limits"] = self._domain_limits(prior, n)
        return plan

    def _domain_limits(self, prior, n):
        """(lower, upper) device vectors of the prior's truncation, or None when there is none: then the conditional
        is sampled exactly (sampler.py:196-197); otherwise by one scan of single-site truncated updates
        (sampler.py:199-205 -> gmrf.gibbs_canonical_truncated_normal, which itself falls back to the exact draw when
        both limits are infinite, gmrf.py:231-232)."""
        lo, hi = prior.domain_response_lower, prior.domain_response_upper
        if lo is None and hi is None:
            return None
        lo = np.full(n, -np.inf) if lo is None else np.broadcast_to(np.asarray(lo, dtype=np.float64).reshape(-1, 1), (n, 1)).reshape(-1)
        hi = np.full(n, np.inf) if hi is None else np.broadcast_to(np.asarray(hi, dtype=np.float64).reshape(-1, 1), (n, 1)).reshape(-1)
        if np.any(lo >= hi):
            raise ValueError("Error lower bound must be strictly less than upper bound")  # gmrf.py:149-150
        if np.all(np.isneginf(lo)) and np.all(np.isposinf(hi)):
            return None
        return self.engine.to_device(lo.copy()), self.engine.to_device(hi.copy())

    def _replicated_structure(self, key, st, n_rep):
        """The structure of n_rep * M (kept on the sampler so the device cache sees one stable matrix object)."""
        from openmcmc_amd.distribution.location_scale import NormalStructure

        memo = self.__dict__.setdefault("_rep_structs", {})
        hit = memo.get(key)
        if hit is None or hit[0] is not st.matrix or hit[1].n_pos != st.n_pos:
            if st.diag is False:  # dense or banded wider than tridiagonal
                rep = NormalStructure(n=st.n, matrix=st.matrix * float(n_rep), scale_key=st.scale_key, diag=False, off=None,
                                      n_pos=st.n_pos, band=None if st.band is None else st.band * float(n_rep))
            else:
                diag = np.full(st.n, float(n_rep)) if st.diag is None else st.diag * float(n_rep)
                off = None if st.off is None else st.off * float(n_rep)
                rep = NormalStructure(n=st.n, matrix=st.matrix * float(n_rep), scale_key=st.scale_key, diag=diag, off=off,
                                      n_pos=st.n_pos)
            hit = memo[key] = (st.matrix, rep)
        return hit[1]

    def _tridiag_plan(self, state, n, pieces):
        eng = self.engine
        terms, keys, offsets = [], [], []
        for pc in pieces:
            st = pc["st"]
            if pc["offset"]:
                if st.diag is not None or st.off is not None:
                    raise NotImplementedError("per-chain offset under a non-identity response precision")
                offsets.append((pc["dist"], st.scale_key))
            cache = eng.model_cache(pc["dist"], state, st, pc["center"])
            scale = _as_chain_scalar(eng, state, st.scale_key) if st.scale_key is not None else None
            terms.append({"diag": cache["diag"], "off": cache["off"], "rhs": cache["rhs"], "center": cache["center"],
                          "scale": None if scale is None else scale.scalar()})
            keys.append(pc["key"])
        # A term centred at a per-chain vector (sampled prior mean, sampled response): where the kernel takes it
        # (omc_tridiag_terms.center_chain: the workgroup-per-chain form) the launch forms s_c M v_c itself and its fused
        # quadratic form is the term's residual statistic; elsewhere the product vector is made by a launch of its own
        # and fed in as a per-chain right-hand side.
        in_launch = eng.tridiag_takes_center_chain(n) and sum(pc.get("chain_vec") is not None for pc in pieces) == 1  # (one per launch)
        chain_rhs = [(pc["chain_vec"], pc["st"].scale_key, eng.model_cache(pc["dist"], state, pc["st"], pc["center"]))
                     for pc in pieces if pc.get("chain_vec") is not None and not in_launch]
        center_chain = [(k, pc["chain_vec"]) for k, pc in enumerate(pieces) if pc.get("chain_vec") is not None and in_launch]
        # terms whose fused quadratic form equals the distribution's residual statistic for the state the draw leaves
        quad_ok = [not pc["offset"] and not pc.get("replicated") and (pc.get("chain_vec") is None or in_launch) for pc in pieces]
        # ... and of those, the ones a LATER block of the same sweep invalidates before anybody could read them (the other
        # side of the term is sampled after this block: MCMC tells the sampler which parameters follow it): not computed
        later = getattr(self, "_later_params", frozenset())
        for k, pc in enumerate(pieces):
            cv = pc.get("chain_vec")
            if quad_ok[k] and cv is not None:
                other = cv[1].mean.get_param_list() if cv[0] == "mean" else [cv[1]]
                if any(key in later for key in other):
                    quad_ok[k] = False
        # With a per-chain centre on the tridiagonal term the library has a specialised kernel for "that term + a scaled
        # identity" (the shifted smoother, omc_tridiag.hip SIG 3).  An unscaled term with a CONSTANT diagonal d0 (a Normal prior
        # N(m0, (d0 I)^-1), say) is that identity with the scalar d0: handed over in that form -- same precision, same right-hand
        # side; its fused quadratic form then lacks the factor d0, which is put back where the form is cached.
        quad_factor = [1.0] * len(pieces)
        if center_chain and not offsets:
            for k, pc in enumerate(pieces):
                st = pc["st"]
                if st.scale_key is None and st.off is None and st.diag is not None and pc.get("chain_vec") is None \
                        and np.ptp(st.diag) == 0.0 and st.diag[0] > 0.0:
                    d0 = float(st.diag[0])
                    cvec = eng.model_cache(pc["dist"], state, st, pc["center"])["center"]
                    terms[k] = {"diag": None, "off": None, "rhs": cvec, "center": cvec, "scale": eng.full((eng.n_chains,), d0)}
                    quad_factor[k] = d0
        return {"kind": "tridiag", "n": n, "terms_list": terms, "terms": eng.tridiag_terms(terms, n), "keys": keys,
                "offsets": offsets, "replicated": any(pc.get("replicated") for pc in pieces), "chain_rhs": chain_rhs,
                "center_chain": center_chain, "quad_ok": quad_ok, "quad_factor": quad_factor}

    def _ragged_plan(self, state, n_max):
        """Small variable-size parameter with a mixture prior (diagonal precision picked by an allocation) and
        regression likelihoods whose design matrix is per chain: Q_c = diag(d_c) + tau_c B_c' W B_c on the live
        block (sampler.py:176-192 with parameter.py:501, location_scale.py:238-241)."""
        likes = []
        for key, dist in self.model.items():
            if key == self.param:
                continue
            if not isinstance(dist, Normal) or not isinstance(dist.mean, LinearCombination):
                raise NotImplementedError("ragged NormalNormal needs LinearCombination likelihood means")
            st = dist.structure(state)
            if st.diag is False or st.off is not None:
                raise NotImplementedError("regression likelihood needs a diagonal response precision")
            if not is_chain(state[dist.mean.form[self.param]]):
                raise NotImplementedError("ragged parameter with a shared design matrix")
            if is_chain(state[key]) or state[key].shape[1] != 1:
                raise NotImplementedError("per-chain or replicated response")
            likes.append((key, dist, st))
        if len(likes) != 1:
            raise NotImplementedError("ragged NormalNormal: exactly one likelihood term")
        return {"kind": "ragged", "n": n_max, "like": likes[0], "terms_list": [], "keys": [], "limits": None}

    def _band_plan(self, state, n, pieces):
        """Q_c = sum_k s_k[c] M_k with banded M_k wider than tridiagonal (RW2, seasonal, lattice GMRFs): natural-order
        band Cholesky per chain (omc_band_sample_canonical)."""
        eng = self.engine
        terms, keys, chain_terms = [], [], []
        for pc in pieces:
            st = pc["st"]
            cache = eng.band_cache(pc["dist"], st, pc["center"])
            scale = _as_chain_scalar(eng, state, st.scale_key) if st.scale_key is not None else None
            rhs = cache["rhs"]
            if rhs is None and cache["center"] is not None:
                rhs = cache["center"]  # identity matrix: M m = m
            terms.append({"band": cache["band"], "rhs": rhs, "scale": None if scale is None else scale.scalar()})
            keys.append(pc["key"])
            op = None if cache["band"] is None else ("band", cache["band"])  # per-chain vectors go through M as a band product
            if pc["offset"]:
                chain_terms.append((("offset", pc["dist"]), st.scale_key, op))
            if pc.get("chain_vec") is not None:
                chain_terms.append((pc["chain_vec"], st.scale_key, op))
        plan = {"kind": "band", "n": n, "terms_list": terms, "terms": eng.band_terms(terms, n), "keys": keys,
                "chain_terms": chain_terms}
        # Long tridiagonal chains arrive here too (beyond one workgroup per chain the band route's segmented kernels, w = 1).  Their
        # quadratic forms (x - m_k)' M_k (x - m_k) are the residual statistics NormalGamma and log_p ask for right after the
        # draw: ONE launch over x for all terms (omc_tridiag_quadform) behind the draw, cached like the fused forms of the
        # workgroup-per-chain kernel -- instead of one pass per distribution and caller (3.3 per sweep at n = 20 000).
        if all(pc["st"].diag is not False and not pc["offset"] and pc.get("chain_vec") is None and not pc.get("replicated") for pc in pieces):
            caches = [eng.model_cache(pc["dist"], state, pc["st"], pc["center"]) for pc in pieces]
            plan["quad_terms"] = eng.tridiag_terms([{"diag": c["diag"], "off": c["off"], "center": c["center"]} for c in caches], n)
        return plan

    def _dense_plan(self, state, n, pieces):
        """Q_c = sum_k s_k[c] M_k with dense M_k: prior precision as is, a regression likelihood as the
        Gram matrix A' W A (one fp64 GEMM at plan time) -- sampler.py:185-192, location_scale.py:238-241."""
        eng = self.engine
        terms, keys, chain_terms = [], [], []
        for pc in pieces:
            st, A = pc["st"], pc["design"]
            center = eng.to_device(pc["center"].reshape(-1))
            per_chain = pc["offset"] or pc.get("chain_vec") is not None
            if A is None:
                if st.n != n:
                    raise ValueError("precision / parameter size mismatch")
                mat = None if (st.diag is None and st.off is None) else eng.shared(st.matrix)
                rhs = center if mat is None else eng.design_rhs(mat, center)  # M' m = M m, once per model
                op = mat
            elif st.diag is False or st.off is not None:
                # a correlated response (tridiagonal, banded or dense W, any symmetric matrix the reference accepts:
                # sampler.py:185-192 forms A'QA and A'Q(y - d) with whatever Q is): W A once per model on the host (a band or
                # sparse product), then the same contraction as the diagonal case, A'(W A), on the device
                Ad = A.toarray() if hasattr(A, "toarray") else np.asarray(A, dtype=np.float64)
                WA = eng.to_device(np.ascontiguousarray(np.asarray(st.matrix @ Ad, dtype=np.float64)))  # (n_obs, p)
                dA = eng.shared(A)
                mat = dA.t().contiguous() @ WA
                mat = (0.5 * (mat + mat.t())).contiguous()
                rhs = (WA.t().contiguous() @ center.reshape(-1, 1)).reshape(-1).contiguous()
                op = WA.t().contiguous() if per_chain else None  # A' W as a (p, n_obs) operator on per-chain vectors
            else:
                dA = eng.shared(A)
                w = None if st.diag is None else eng.to_device(st.diag)
                mat, rhs = eng.gram(dA, w), eng.design_rhs(dA, center, w)
                op = None
                if per_chain:  # A' W as a (p, n_obs) operator on per-chain vectors of the observation space
                    Ad = A.toarray() if hasattr(A, "toarray") else np.asarray(A, dtype=np.float64)
                    op = eng.to_device(np.ascontiguousarray((Ad if st.diag is None else Ad * np.asarray(st.diag).reshape(-1, 1)).T))
            # what the right-hand side receives per chain and per sweep (sampler.py:181-192): the part of the mean carried by
            # other sampled parameters (b_c -= s_c A'W d_c), a sampled prior mean (b_c += s_c M m_c), a sampled response
            # (b_c += s_c W y_c) -- each one GEMM over all chains with the term's shared operator
            if pc["offset"]:
                chain_terms.append((("offset", pc["dist"]), st.scale_key, op))
            if pc.get("chain_vec") is not None:
                chain_terms.append((pc["chain_vec"], st.scale_key, op))
            scale = _as_chain_scalar(eng, state, st.scale_key) if st.scale_key is not None else None
            terms.append({"mat": mat, "rhs": rhs, "scale": None if scale is None else scale.scalar()})
            keys.append(pc["key"])
        return {"kind": "dense", "n": n, "terms_list": terms, "terms": eng.dense_terms(terms, n), "keys": keys,
                "chain_terms": chain_terms}

    spectral = True  # class-level switch: False keeps every dense draw on the per-chain factorisation
    TRIDIAG_WG_MAX = 16384  # largest n the workgroup-per-chain tridiagonal kernel takes (omc_tridiag.hip: seg_max_n(32))

    def _spectral_plan(self, p):
        """(k_mat, V, ev) if the dense plan is `scaled identities + one shared matrix` (order >= 64), else None; the
        eigendecomposition is made once per plan."""
        if "spectral" not in p:
            mats = [k for k, t in enumerate(p["terms_list"]) if t["mat"] is not None]
            p["spectral"] = None
            if len(mats) == 1 and p["n"] >= 64 and p.get("mixture_prior") is None and p.get("limits") is None:
                V, ev = self.engine.dense_spectral_prepare(p["terms_list"][mats[0]]["mat"])
                p["spectral"] = (mats[0], V, ev)
        return p["spectral"]

    def plan(self, state):
        p = self._plan
        if p is not None:
            # the per-chain scalars must still be the tensors the plan points at
            for t, key in zip(p["terms_list"], p["keys"]):
                st_key = self.model[key].precision.scalar if isinstance(self.model[key].precision, ScaledMatrix) else None
                if st_key is not None and state[st_key].scalar().data_ptr() != t["scale"].data_ptr():
                    p = None
                    break
        if p is None:
            p = self._plan = self._build_plan(state)
        return p

    def sample(self, current_state: dict, out=None) -> dict:
        """sampler.py:154-207.  `out`: optional (C, n) destination (e.g. a store slab)."""
        eng = self._need_engine()
        p = self.plan(current_state)
        n = p["n"]
        z = self.inject(self, self._sweep) if self.inject is not None else None
        if p["kind"] == "ragged":
            return self._sample_ragged(current_state, p, z)
        x = eng.empty(eng.n_chains, n) if out is None else out
        rhs_chain = None
        for dist, scale_key in p.get("offsets", ()):  # b_c -= tau_c * d_c  (sampler.py:190-192)
            scale = current_state[scale_key].scalar() if scale_key is not None else None
            t = dist.mean.predictor_device(current_state, eng, exclude=self.param, alpha=-1.0, chain_scale=scale)
            rhs_chain = t if rhs_chain is None else eng.chain_lincomb(1.0, rhs_chain, 1.0, t)
        for (kind, what), scale_key, cache in p.get("chain_rhs", ()):  # per-chain prior mean / per-chain response
            v = what.mean.predictor_device(current_state, eng) if kind == "mean" and not isinstance(what.mean, Identity) else \
                current_state[what.mean.form if kind == "mean" else what].vector()
            scale = current_state[scale_key].scalar() if scale_key is not None else None
            if rhs_chain is None:
                rhs_chain = eng.tridiag_matvec_chain(n, cache["diag"], cache["off"], v, scale=scale)
            else:
                eng.tridiag_matvec_chain(n, cache["diag"], cache["off"], v, scale=scale, out=rhs_chain, accumulate=True)
        for (kind, what), scale_key, op in p.get("chain_terms", ()):  # dense and band routes: the same three kinds, through `op`
            scale = current_state[scale_key].scalar() if scale_key is not None else None
            if kind == "offset":
                v = what.mean.predictor_device(current_state, eng, exclude=self.param, alpha=-1.0, chain_scale=scale)
            else:
                v = what.mean.predictor_device(current_state, eng) if kind == "mean" and not isinstance(what.mean, Identity) else \
                    current_state[what.mean.form if kind == "mean" else what].vector()
                if scale is not None and op is not None and not isinstance(op, tuple):
                    v = eng.tridiag_matvec_chain(v.shape[1], None, None, v, scale=scale)  # s_c v_c (identity matrix)
            if op is None:
                t = v if (scale is None or kind == "offset") else eng.tridiag_matvec_chain(n, None, None, v, scale=scale)
            elif isinstance(op, tuple):
                t = eng.band_matvec_chain(n, op[1], v, scale=None if kind == "offset" else scale)
            else:
                t = eng.design_predict(op, v)
            rhs_chain = t if rhs_chain is None else eng.chain_lincomb(1.0, rhs_chain, 1.0, t)
        if p["limits"] is not None:
            # truncated prior: the scan starts from the current value and `z` carries the injected UNIFORMS
            lower, upper = p["limits"]
            eng.chain_copy(current_state[self.param].vector(), x)
            gibbs = {"tridiag": eng.tridiag_gibbs_truncated, "band": eng.band_gibbs_truncated}.get(p["kind"], eng.dense_gibbs_truncated)
            gibbs(n, p["terms"], x, lower=lower, upper=upper, u=z, rhs_chain=rhs_chain, draw_index=self._draw_index())
        elif p["kind"] == "tridiag":
            vecs = [None] * len(p["keys"])
            for k, (kind, what) in p.get("center_chain", ()):
                vecs[k] = what.mean.predictor_device(current_state, eng) if kind == "mean" and not isinstance(what.mean, Identity) else \
                    current_state[what.mean.form if kind == "mean" else what].vector()
            if p.get("center_chain"):
                eng.set_center_chain(p["terms"], vecs, n)
            quad = eng.empty(len(p["keys"]), eng.n_chains) if any(p.get("quad_ok", ())) else None
            skip = sum(1 << k for k, ok in enumerate(p.get("quad_ok", ())) if not ok) if quad is not None else 0
            if skip:
                eng.set_option("tridiag_quad_skip", skip)
            try:
                eng.tridiag_sample_canonical(n, p["terms"], x, z=z, rhs_chain=rhs_chain, draw_index=self._draw_index(), quad_out=quad)
            finally:
                if skip:
                    eng.set_option("tridiag_quad_skip", 0)
            if quad is not None:
                # the quadratic forms of the launch are the residual statistics of these distributions for the new state:
                # NormalGamma and log_p take them from here instead of a pass of their own over the state
                new_state = dict(current_state)
                new_state[self.param] = ChainArray(x)
                for k, key in enumerate(p["keys"]):
                    if p["quad_ok"][k]:
                        f = p.get("quad_factor", (1.0,) * len(p["keys"]))[k]
                        eng.quad_cache_put(self.model[key], quad[k] if f == 1.0 else quad[k] * f, self.model[key].residual_inputs(new_state))
        elif p["kind"] == "band":
            eng.band_sample_canonical(n, p["terms"], x, z=z, rhs_chain=rhs_chain, draw_index=self._draw_index())
            if p.get("quad_terms") is not None:
                quad = eng.empty(len(p["keys"]), eng.n_chains)
                eng.tridiag_quadform(n, p["quad_terms"], x, quad)
                new_state = dict(current_state)
                new_state[self.param] = ChainArray(x)
                for k, key in enumerate(p["keys"]):
                    eng.quad_cache_put(self.model[key], quad[k], self.model[key].residual_inputs(new_state))
        elif p.get("mixture_prior") is not None:
            # prior N(mean[alloc], diag(prec[alloc])^-1) (parameter.py:447,501): a per-chain diagonal on Q and
            # prec * mean on b (sampler.py:181-183), next to the shared likelihood terms
            _, pmean, pprec, _ = p["mixture_prior"].mixture_pieces(current_state, eng)
            prior_rhs = pprec * pmean
            eng.dense_sample_canonical(n, p["terms"], x, z=z, diag_chain=pprec, draw_index=self._draw_index(),
                                       rhs_chain=prior_rhs if rhs_chain is None else eng.chain_lincomb(1.0, prior_rhs, 1.0, rhs_chain))
        elif z is None and self.spectral and self._spectral_plan(p) is not None:
            # Q_c = a_c I + b_c M with one shared M: the draw in M's eigenbasis -- two GEMMs over all chains instead of
            # one factorisation per chain.  Same conditional law and the reference's mean and log det; not its path-wise
            # image of the draws, so a replay with injected draws (z given) takes the factorisation below.
            k_mat, V, ev = self._spectral_plan(p)
            eng.dense_spectral_sample(n, p["terms"], k_mat, V, ev, x, rhs_chain=rhs_chain, draw_index=self._draw_index())
        else:
            eng.dense_sample_canonical(n, p["terms"], x, z=z, rhs_chain=rhs_chain, draw_index=self._draw_index())
        current_state[self.param] = ChainArray(x)
        self._sweep += 1
        return current_state

    def _sample_ragged(self, state, p, z):
        eng = self.engine
        key, dist, st = p["like"]
        prior = self.model[self.param]
        cur = state[self.param]
        _, pmean, pprec, count = prior.mixture_pieces(state, eng)
        B = state[dist.mean.form[self.param]]
        w = None if st.diag is None else eng.shared(st.diag)
        y = eng.shared(state[key]).reshape(-1)  # cached by the identity of the state's own array
        rest = None
        if dist.mean.has_chain_terms(state, exclude=self.param):
            rest = dist.mean.predictor_device(state, eng, exclude=self.param)
        else:
            host = dist.mean.predictor_conditional(state, term_to_exclude=self.param)
            if not isinstance(host, int):
                y = eng.to_device(np.asarray(state[key], dtype=np.float64).reshape(-1) - np.asarray(host).reshape(-1))
        gram, rhs = eng.design_gram_batched(B.columns(), w=w, resid_shared=y, resid_chain=rest, count=B.count(state))
        scale = _as_chain_scalar(eng, state, st.scale_key).scalar() if st.scale_key is not None else None
        x = eng.small_sample_canonical(gram, rhs, pprec, lik_scale=scale, prior_mean=pmean, count=count, z=z,
                                       draw_index=self._draw_index())
        state[self.param] = cur.like(x.unsqueeze(2))
        self._sweep += 1
        return state


def _mean_is_chain(mean, state):
    """Does the mean parameter evaluate to a per-chain vector in this state?"""
    if isinstance(mean, Identity):
        return is_chain(state[mean.form])
    form = getattr(mean, "form", None)
    if isinstance(form, dict):
        return any(is_chain(state[k]) for k in form)
    return False


@dataclass
class NormalGamma(MCMCSampler):
    """Normal-Gamma conjugate update of a scalar precision (sampler.py:210-288)."""

    def __post_init__(self):
        super().__post_init__()
        others = [k for k in self.model.keys() if k != self.param]
        self.normal_param = others[0]
        precision = self.model[self.normal_param].precision
        if not isinstance(precision, (Identity, ScaledMatrix, MixtureParameterMatrix)):
            raise TypeError("precision must be either Identity, ScaledMatrix or MixtureParameterMatrix")

    def prior_shape_rate(self, state):
        return self.model[self.param].host_shape_rate(state)

    def sample(self, current_state: dict) -> dict:
        """a = a0 + #{diag(P)>0}/2, b = b0 + r'Pr/2, lambda ~ Gamma(a, scale 1/b)  (sampler.py:252-288)."""
        eng = self._need_engine()
        dist = self.model[self.normal_param]
        if dist.is_mixture:
            return self._sample_mixture(current_state, dist)
        st = dist.structure(current_state)
        quad = dist.residual_quad(current_state, eng, st)
        a0, b0 = self.prior_shape_rate(current_state)
        target = _as_chain_scalar(eng, current_state, self.param)
        g = self.inject(self, self._sweep) if self.inject is not None else None
        eng.normal_gamma_update(a0, b0, st.n_pos, quad, target.scalar(), g=g, draw_index=self._draw_index())
        self._sweep += 1
        return current_state

    def _sample_mixture(self, state, dist):
        """One precision per mixture component (sampler.py:276-287 with MixtureParameterMatrix.precision_unscaled,
        parameter.py:525-538): a_k = a0_k + #{alloc == k}/2, b_k = b0_k + sum_{alloc == k} r_i^2 / 2."""
        eng = self.engine
        target = state[self.param]
        if not is_chain(target) or target.shape[1] != 1 or target.ragged is not None:
            raise NotImplementedError("mixture precision vector must be a per-chain (K, 1) parameter")
        K = target.shape[0]
        x, pmean, _, count = dist.mixture_pieces(state, eng)
        if count is not None:
            raise NotImplementedError("NormalGamma on a variable-size mixture")
        alloc = state[dist.precision.allocation].vector()
        a0, b0 = self.model[self.param].host_shape_rate_vec(state, K)
        consts = self.__dict__.setdefault("_ab_dev", {})
        key = (a0.tobytes(), b0.tobytes())
        if key not in consts:
            consts[key] = (eng.to_device(a0), eng.to_device(b0))
        g = self.inject(self, self._sweep) if self.inject is not None else None
        out = eng.mixture_normal_gamma(eng.chain_lincomb(1.0, x, -1.0, pmean), alloc, consts[key][0], consts[key][1], g=g, draw_index=self._draw_index())
        state[self.param] = ChainArray(out.unsqueeze(2))
        self._sweep += 1
        return state


@dataclass
class MixtureAllocation(MCMCSampler):
    """Conditional draw of the allocation of a mixture (sampler.py:291-355): for every element of the response
    parameter, category k with probability proportional to prior_k * N(y_i; mean_k, 1/prec_k)."""

    response_param: Union[str, None] = None

    def __post_init__(self):
        self.model = Model([self.model[self.param], self.model[self.response_param]])
        self._init_runtime()
        resp = self.model[self.response_param]
        if not isinstance(resp, Normal):
            raise TypeError("Mixture model currently only implemented for Normal case")
        if not isinstance(resp.mean, MixtureParameterVector):
            raise TypeError("Mean must be of type MixtureParameterVector")
        if not isinstance(resp.precision, MixtureParameterMatrix):
            raise TypeError("Mean must be of type MixtureParameterMatrix")

    def sample(self, current_state: dict) -> dict:
        eng = self._need_engine()
        resp = self.model[self.response_param]
        prior = self.model[self.param].prob.predictor(current_state)
        if is_chain(prior):
            raise NotImplementedError("per-chain allocation probabilities")
        y = current_state[self.response_param]
        if not is_chain(y) or y.shape[1] != 1:
            raise NotImplementedError("MixtureAllocation needs a per-chain (p, 1) response parameter")

        def table(key):  # component means / precisions: shared (K, 1) host array or per-chain (K, 1)
            v = current_state[key]
            return v.vector() if is_chain(v) else eng.shared(v).reshape(-1)

        u = self.inject(self, self._sweep) if self.inject is not None else None
        alloc = eng.mixture_allocation(y.vector(), eng.shared(prior), table(resp.mean.param),
                                       table(resp.precision.param), u=u, draw_index=self._draw_index())
        current_state[self.param] = ChainArray(alloc.unsqueeze(2))
        self._sweep += 1
        return current_state
