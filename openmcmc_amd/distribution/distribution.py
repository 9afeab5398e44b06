"""Distribution base class and the Gamma prior (reference distribution/distribution.py:28-278).

log_p on the GPU path returns one value per chain: a (C,) device tensor.
"""

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Union

import numpy as np
from math import lgamma as _lgamma

from openmcmc_amd.chains import ChainArray, is_chain
from openmcmc_amd.parameter import Identity, LinearCombination, MixtureParameterVector


@dataclass
class Distribution(ABC):
    """A distribution for state[response] (distribution.py:28-122)."""

    response: str

    @abstractmethod
    def log_p(self, state: dict, by_observation: bool = False):
        """Log-density per chain."""

    @abstractmethod
    def rvs(self, state: dict, n: int = 1):
        """Random draws, one set per chain."""

    @property
    @abstractmethod
    def _dist_params(self) -> list:
        """State labels used by the distribution's parameters (excluding the response)."""

    @property
    def param_list(self) -> list:
        return [self.response] + self._dist_params

    # ------------------------------------------------------------------ finite-difference derivatives
    def grad_log_p(self, state: dict, param: str, hessian_required: bool = True, engine=None):
        """Default of the base class (distribution.py:90-122): central differences of log_p."""
        grad = self.grad_log_p_diff(state, param, engine=engine)
        if hessian_required:
            return grad, self.hessian_log_p_diff(state, param, engine=engine)
        return grad

    def _perturbed(self, state, param, k, delta):
        """state with element k (row-major over (p, n_rep)) of the per-chain parameter shifted by delta on every chain."""
        x = state[param]
        data = x.data.clone()
        data.view(x.n_chains, -1)[:, k] += delta
        out = dict(state)
        out[param] = x.like(data)
        return out

    def grad_log_p_diff(self, state: dict, param: str, step_size: float = 1e-4, engine=None):
        """distribution.py:124-158 for every chain: (log_p(x + h/2 e_k) - log_p(x - h/2 e_k)) / h, k over the elements of
        the per-chain parameter.  Each log_p is one batched device evaluation; returns a ChainArray shaped like the
        parameter."""
        if engine is None:
            raise RuntimeError("grad_log_p_diff needs the engine")
        x = state[param]
        if not is_chain(x) or x.ragged is not None:
            raise NotImplementedError("finite differences need a fixed-size per-chain parameter")
        grad = engine.empty(x.n_chains, x.size)
        for k in range(x.size):
            lp_plus = self.log_p(self._perturbed(state, param, k, step_size / 2), engine=engine)
            lp_minus = self.log_p(self._perturbed(state, param, k, -step_size / 2), engine=engine)
            grad[:, k] = (lp_plus - lp_minus) / step_size
        return ChainArray(grad.reshape(x.data.shape))

    def hessian_log_p_diff(self, state: dict, param: str, step_size: float = 1e-4, engine=None):
        """distribution.py:160-198: column k = (grad(x - h/2 e_k) - grad(x + h/2 e_k)) / h (NEGATIVE second derivatives);
        (C, d, d) tensor."""
        if engine is None:
            raise RuntimeError("hessian_log_p_diff needs the engine")
        x = state[param]
        d = x.size
        hess = engine.empty(x.n_chains, d, d)
        for k in range(d):
            g_plus = self.grad_log_p(self._perturbed(state, param, k, step_size / 2), param, hessian_required=False, engine=engine)
            g_minus = self.grad_log_p(self._perturbed(state, param, k, -step_size / 2), param, hessian_required=False, engine=engine)
            hess[:, :, k] = (g_minus.data - g_plus.data).reshape(x.n_chains, d) / step_size
        return hess


@dataclass
class Gamma(Distribution):
    """Gamma(shape, rate) (distribution.py:201-278)."""

    shape: Union[str, Identity, LinearCombination]
    rate: Union[str, Identity, LinearCombination]

    def __post_init__(self):
        if isinstance(self.shape, str):
            self.shape = Identity(self.shape)
        if not isinstance(self.shape, (Identity, LinearCombination)):
            raise TypeError("shape expected to be one of [Identity, LinearCombination, MixtureParameterVector]")
        if isinstance(self.rate, str):
            self.rate = Identity(self.rate)
        if not isinstance(self.rate, (Identity, LinearCombination)):
            raise TypeError("rate expected to be one of [Identity, LinearCombination, MixtureParameterVector]")

    @property
    def _dist_params(self) -> list:
        return self.shape.get_param_list() + self.rate.get_param_list()

    def host_shape_rate(self, state):
        """Scalar prior parameters; they are host constants in every supported model.  A vector of identical entries
        (one per mixture component, e.g. gamma_shape = 1e-3 * ones(n_cat)) counts as that scalar."""
        a, b = self.shape.predictor(state), self.rate.predictor(state)
        if is_chain(a) or is_chain(b):
            raise NotImplementedError("Gamma prior parameters must be shared on the GPU path")
        a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
        if np.any(a != a[0]) or np.any(b != b[0]):
            raise NotImplementedError("element-wise different Gamma shape / rate: use host_shape_rate_vec")
        return float(a[0]), float(b[0])

    def host_shape_rate_vec(self, state, K):
        """(K,) host vectors of the shape and rate (scalars are broadcast)."""
        a, b = self.shape.predictor(state), self.rate.predictor(state)
        if is_chain(a) or is_chain(b):
            raise NotImplementedError("Gamma prior parameters must be shared on the GPU path")
        bc = lambda v: np.broadcast_to(np.asarray(v, dtype=np.float64).reshape(-1), (K,)).copy()  # noqa: E731
        return bc(a), bc(b)

    def log_p_piece(self, state, engine, dry=False):
        """This distribution's term of Model.log_p as a piece of omc_log_post_sum (a per-chain scalar response), else None."""
        x = state[self.response]
        if type(self) is not Gamma or not is_chain(x) or x.size != 1 or x.ragged is not None:
            return None
        if dry:
            return True
        a, b = self.host_shape_rate(state)
        return ("gamma", x.scalar(), float(a), float(b))

    def log_p(self, state: dict, by_observation: bool = False, engine=None, out=None, accumulate=False):
        """distribution.py:241-261.  Per-chain response -> (C,) tensor via omc_gamma_logpdf; a (1, k) response (ragged or
        not: e.g. the kernel widths of a reversible-jump basis) sums over its live entries."""
        x = state[self.response]
        a, b = self.host_shape_rate(state)
        if not is_chain(x):
            from scipy import stats

            return float(np.sum(stats.gamma.logpdf(np.asarray(x, dtype=np.float64), a, scale=1.0 / b)))
        if by_observation:
            # distribution.py:255-259: the sum over the p rows stays, the replicate columns are kept: (C, n_rep); NaN padding of
            # a ragged response stays out of the sum and reads 0 beyond the live length
            import torch

            v = x.data  # (C, p, n_rep)
            lp = (a * float(np.log(b)) - float(_lgamma(a))) + (a - 1.0) * torch.log(v) - b * v
            lp = torch.where(v > 0, lp, torch.full_like(lp, float("-inf")))
            if x.ragged is not None:
                axis = x.ragged[1]
                live = torch.arange(v.shape[1 + axis], device=v.device).reshape((1, -1, 1) if axis == 0 else (1, 1, -1)) \
                    < x.count(state).reshape(-1, 1, 1)
                lp = torch.where(live, lp, torch.zeros_like(lp))
            return lp.sum(dim=1)
        if engine is None:
            raise RuntimeError("Gamma.log_p on a per-chain response needs the engine (use Model.log_p)")
        out = engine.empty(engine.n_chains) if out is None else out
        if x.size == 1 and x.ragged is None:
            engine.gamma_logpdf(x.scalar(), a, b, out, accumulate=accumulate)
        elif x.shape[1] == 1 and x.ragged is None:
            # (K, 1) response, e.g. the per-component precisions of a mixture: the same scalar shape / rate for all
            K = x.shape[0]
            engine.gamma_logpdf_vec(x.vector(), engine.full((K,), a), engine.full((K,), b), out, accumulate=accumulate)
        else:
            if x.shape[0] != 1:
                raise NotImplementedError("vector-valued Gamma response with replicates")
            engine.gamma_logpdf_ragged(x.data[:, 0, :], a, b, out, count=x.count(state), accumulate=accumulate)
        return out

    def log_p_last(self, state: dict, engine):
        """log-density of the LAST live replicate of a per-chain (1, k) response: what ReversibleJump reads from
        log_p(current_state, by_observation=True)[-1] (reversible_jump.py:132,143).  (C,) tensor."""
        x = state[self.response]
        a, b = self.host_shape_rate(state)
        out = engine.empty(engine.n_chains)
        engine.gamma_logpdf_ragged(x.data[:, 0, :], a, b, out, count=x.count(state), last_only=True)
        return out

    def rvs(self, state, n: int = 1, engine=None, draw_index=0, sub=0, inject=None):
        """distribution.py:263-278: one Gamma(shape, rate) draw per chain (prior draw when the state has no initial
        value, mcmc.py:78-80; new element of an associated parameter in a birth move, reversible_jump.py:130).
        `inject`: (C,) standard-gamma draws Gamma(shape, 1), as scipy's gamma.rvs(a, scale=s) == standard_gamma(a)*s."""
        if engine is None:
            raise RuntimeError("Gamma.rvs needs the engine")
        if n != 1:
            raise NotImplementedError("replicated prior draws")
        a, b = self.host_shape_rate(state)
        out = engine.empty(engine.n_chains)
        # Gamma(a, rate b) = the conjugate update with no data: n_pos = 0, quad = 0
        # `sub` separates several Gamma draws made under one draw index (bits 44-47 of the 48-bit stream index)
        engine.normal_gamma_update(a, b, 0, engine.zeros(engine.n_chains), out, g=inject,
                                   draw_index=int(draw_index) + ((int(sub) & 0xF) << 44))
        return ChainArray(out.reshape(-1, 1, 1))


@dataclass
class Uniform(Distribution):
    """Uniform on a p-dimensional box (distribution.py:377-458); limits are shared (p, 1) arrays."""

    domain_response_lower: Union[float, np.ndarray] = 0.0
    domain_response_upper: Union[float, np.ndarray] = 1.0

    def __post_init__(self):
        self.domain_response_lower = np.array(self.domain_response_lower, ndmin=2, dtype=np.float64)
        if self.domain_response_lower.shape[0] == 1:
            self.domain_response_lower = self.domain_response_lower.T
        self.domain_response_upper = np.array(self.domain_response_upper, ndmin=2, dtype=np.float64)
        if self.domain_response_upper.shape[0] == 1:
            self.domain_response_upper = self.domain_response_upper.T

    @property
    def _dist_params(self) -> list:
        return []

    def domain_range(self, state) -> np.ndarray:
        d = state[self.response].shape[0]
        rng = self.domain_response_upper - self.domain_response_lower
        return np.ones((d, 1)) * rng if rng.size == 1 else rng

    def log_p_per_replicate(self, state) -> float:
        """-sum_p log(range): what every replicate contributes (distribution.py:437)."""
        return float(-np.sum(np.log(self.domain_range(state))))

    def log_p_last(self, state: dict, engine=None) -> float:
        """log_p(state, by_observation=True)[-1] (reversible_jump.py:132,143): the same constant for every replicate."""
        return self.log_p_per_replicate(state)

    def log_p(self, state: dict, by_observation: bool = False, engine=None, out=None, accumulate=False):
        """distribution.py:422-442: n_rep * (-sum log range); per chain n_rep is the live length of a ragged response."""
        x = state[self.response]
        per = self.log_p_per_replicate(state)
        if by_observation:
            if is_chain(x):  # distribution.py:436-440: the same constant for every replicate, 0 beyond a ragged response's live length
                import torch

                k = x.data.shape[2]
                out_ = torch.full((x.n_chains, k), float(per), dtype=torch.float64, device=x.data.device)
                if x.ragged is not None and x.ragged[1] == 1:
                    live = torch.arange(k, device=x.data.device).reshape(1, -1) < x.count(state).reshape(-1, 1)
                    out_ = torch.where(live, out_, torch.zeros_like(out_))
                return out_
            return np.ones(x.shape[1]) * per
        if not is_chain(x):
            return x.shape[1] * per
        if engine is None:
            raise RuntimeError("Uniform.log_p on a per-chain response needs the engine (use Model.log_p)")
        out = engine.empty(engine.n_chains) if out is None else out
        if x.ragged is not None and x.ragged[1] == 1:
            engine.count_logpdf(x.count(state), per, out, accumulate=accumulate)
        elif accumulate:
            out += x.shape[1] * per
        else:
            out.fill_(x.shape[1] * per)
        return out

    def rvs(self, state, n: int = 1, engine=None, draw_index=0, sub=0, inject=None):
        """lower + range * U(0,1)^(p x n) for every chain (distribution.py:444-458); `inject` (C, p*n) uniforms."""
        if engine is None:
            raise RuntimeError("Uniform.rvs needs the engine")
        p = state[self.response].shape[0]
        lo = np.broadcast_to(self.domain_response_lower, (p, 1)).reshape(-1)
        rng = self.domain_range(state).reshape(-1)
        u = None if inject is None else inject.reshape(engine.n_chains, p * n).contiguous()
        consts = self.__dict__.setdefault("_dev_consts", {})
        if (p, n) not in consts:  # uploaded once: a host-to-device copy per call would stall the launch queue
            consts[(p, n)] = (engine.to_device(np.repeat(lo, n)), engine.to_device(np.repeat(rng, n)))
        lo_d, rng_d = consts[(p, n)]
        draw = engine.uniform_draw(lo_d, rng_d, inject=u, draw_index=draw_index, sub=sub)
        return ChainArray(draw.reshape(engine.n_chains, p, n))


@dataclass
class Poisson(Distribution):
    """Poisson count (distribution.py:461-523); the rate is a shared scalar on the GPU path."""

    rate: Union[str, Identity, LinearCombination, MixtureParameterVector]

    def __post_init__(self):
        if isinstance(self.rate, str):
            self.rate = Identity(self.rate)
        if not isinstance(self.rate, (Identity, LinearCombination, MixtureParameterVector)):
            raise TypeError("rate expected to be one of [Identity, LinearCombination, MixtureParameterVector]")

    @property
    def _dist_params(self) -> list:
        return self.rate.get_param_list()

    def log_p(self, state: dict, by_observation: bool = False, engine=None, out=None, accumulate=False):
        """distribution.py:490-508: k log(rate) - rate - lgamma(k + 1)."""
        x = state[self.response]
        rate = self.rate.predictor(state)
        if is_chain(rate) or np.size(rate) != 1:
            raise NotImplementedError("Poisson rate must be a shared scalar on the GPU path")
        rate = float(np.asarray(rate).item())
        if not is_chain(x):
            from scipy import stats

            return float(np.sum(stats.poisson.logpmf(x, rate)))
        if x.size != 1:
            raise NotImplementedError("vector-valued Poisson response")
        if engine is None:
            raise RuntimeError("Poisson.log_p on a per-chain response needs the engine (use Model.log_p)")
        out = engine.empty(engine.n_chains) if out is None else out
        engine.poisson_logpmf(x.scalar(), rate, out, accumulate=accumulate)
        return out

    def rvs(self, state: dict, n: int = 1, engine=None, draw_index=0, inject=None):
        """One Poisson count per chain (distribution.py:508-523 -> scipy.stats.poisson.rvs, i.e. NumPy's legacy generator;
        omc_poisson_draw follows its algorithm draw for draw): the prior draw of a jump parameter without an initial value
        (mcmc.py:78-85).  `inject` (C, K): the uniforms each chain's draw consumes, NaN-padded (parity tests)."""
        if engine is None:
            raise RuntimeError("Poisson.rvs needs the engine")
        if n != 1:
            raise NotImplementedError("replicated prior draws")
        rate = self.rate.predictor(state)
        if is_chain(rate):
            if rate.size != 1:
                raise NotImplementedError("vector-valued Poisson response")
            rate = rate.scalar()
        elif np.size(rate) != 1:
            raise NotImplementedError("vector-valued Poisson response")
        else:
            rate = float(np.asarray(rate).item())
        u = None if inject is None else engine.to_device(inject).reshape(engine.n_chains, -1).contiguous()
        return ChainArray(engine.poisson_draw(rate, u=u, draw_index=draw_index).reshape(engine.n_chains, 1, 1))


@dataclass
class Categorical(Distribution):
    """Categorical distribution (distribution.py:281-375): the response holds category indices 0..n_cat-1, one trial
    per element; `prob` is a shared (1 or p, n_cat) array."""

    prob: Union[str, Identity]

    def __post_init__(self):
        if isinstance(self.prob, str):
            self.prob = Identity(self.prob)
        if not isinstance(self.prob, Identity):
            raise TypeError("prob expected to be Identity")

    @property
    def _dist_params(self) -> list:
        return self.prob.get_param_list()

    def _prob(self, state):
        prob = self.prob.predictor(state)
        if is_chain(prob):
            raise NotImplementedError("per-chain allocation probabilities")
        return np.asarray(prob, dtype=np.float64)

    def log_p(self, state: dict, by_observation: bool = False, engine=None, out=None, accumulate=False):
        """distribution.py:318-352 for one replicate: sum_i log prob[i, z_i] (multinomial.logpmf with one trial)."""
        x, prob = state[self.response], self._prob(state)
        if by_observation:
            raise NotImplementedError("by_observation")
        if not is_chain(x):
            z = np.asarray(x).astype(int).reshape(-1)
            rows = np.zeros_like(z) if prob.shape[0] == 1 else np.arange(z.size)
            return float(np.sum(np.log(prob[rows, z])))
        if x.shape[1] != 1:
            raise NotImplementedError("replicated categorical response")
        if engine is None:
            raise RuntimeError("Categorical.log_p on a per-chain response needs the engine (use Model.log_p)")
        out = engine.empty(engine.n_chains) if out is None else out
        engine.categorical_logpmf(x.vector(), engine.shared(prob), out, accumulate=accumulate)
        return out

    def rvs(self, state, n: int = 1, engine=None, draw_index=0):
        """One category per element and chain by inverse CDF of a uniform (distribution-equivalent to the reference's
        multinomial draw, distribution.py:354-375; used for missing initial values only)."""
        if engine is None:
            raise RuntimeError("Categorical.rvs needs the engine")
        if n != 1:
            raise NotImplementedError("replicated prior draws")
        prob = self._prob(state)
        p, K = prob.shape[0], prob.shape[1]
        flat = engine.zeros(engine.n_chains, p)  # the same likelihood for every component: the draw is from the prior
        alloc = engine.mixture_allocation(flat, engine.shared(prob), engine.zeros(K), engine.full((K,), 1.0),
                                          draw_index=draw_index)
        return ChainArray(alloc.unsqueeze(2))
