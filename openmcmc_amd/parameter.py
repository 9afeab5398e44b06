"""Parameter objects: how a distribution parameter is built from named state entries.

Same names, fields and meaning as the reference (parameter.py:74-141 Identity, :144-228
LinearCombination, :300-373 ScaledMatrix).  They are declarative: samplers read `form` /
`matrix` / `scalar` to lay out GPU work; `predictor` evaluates on host constants and, for
per-chain entries, returns ChainArray results for the cases the hot path needs.
"""

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Union

import numpy as np

from openmcmc_amd.chains import ChainArray, is_chain


@dataclass
class Parameter(ABC):
    """Abstract parameter (reference parameter.py:26-71)."""

    @abstractmethod
    def predictor(self, state: dict):
        """Value of the parameter for the given state."""

    @abstractmethod
    def get_param_list(self) -> list:
        """All state labels the parameter reads."""

    @abstractmethod
    def get_grad_param_list(self) -> list:
        """State labels the gradient is defined for."""


@dataclass
class Identity(Parameter):
    """f = state[form]  (parameter.py:74-141)."""

    form: str

    def predictor(self, state: dict):
        return state[self.form]

    def get_param_list(self) -> list:
        return [self.form]

    def get_grad_param_list(self) -> list:
        return [self.form]


@dataclass
class LinearCombination(Parameter):
    """f = sum_i state[prefactor_i] @ state[param_i], form = {param: prefactor}  (parameter.py:144-228)."""

    form: dict

    def predictor(self, state: dict):
        return self.predictor_conditional(state)

    def predictor_conditional(self, state: dict, term_to_exclude: Union[str, list] = None):
        """Sum of the terms not excluded (parameter.py:174-197).  Host terms are summed on the host;
        a per-chain term is supported when its prefactor is a (sparse) identity."""
        skip = [] if term_to_exclude is None else ([term_to_exclude] if isinstance(term_to_exclude, str) else term_to_exclude)
        host_sum, chain_sum = 0, None
        for prm, prefactor in self.form.items():
            if prm in skip:
                continue
            A, v = state[prefactor], state[prm]
            if is_chain(v):
                if not _is_identity(A, v.shape[0]):
                    raise NotImplementedError("per-chain term with a non-identity design matrix: use predictor_device (needs the engine)")
                chain_sum = v if chain_sum is None else ChainArray(chain_sum.data + v.data)
            else:
                host_sum = host_sum + A @ v
        if chain_sum is None:
            return host_sum
        if isinstance(host_sum, int):
            return chain_sum
        import torch

        return ChainArray(chain_sum.data + torch.as_tensor(np.asarray(host_sum), device=chain_sum.data.device))

    def predictor_device(self, state: dict, engine, out=None, exclude=None, alpha=1.0, chain_scale=None):
        """(C, n) tensor chain_scale[c] * alpha * sum_i A_i x_i over the terms not excluded, with per-chain terms
        evaluated on the GPU and shared terms added once (parameter.py:162-197).  A design matrix may be shared
        (one GEMM over all chains), an identity, or itself per chain -- a ChainArray (n, k) basis that depends on
        per-chain knots, possibly ragged -- which goes to omc_design_predict_batched."""
        # While the state is frozen (MCMC's bookkeeping at the end of a sweep: fitted-value store, then log_post) the full
        # predictor is evaluated once: `_frozen` is a dict the MCMC loop hangs on the parameter for that stretch.
        frozen = getattr(self, "_frozen", None)
        plain = exclude is None and alpha == 1.0 and chain_scale is None
        if frozen is not None and plain and "fitted" in frozen:
            hit = frozen["fitted"]
            if out is None or out.data_ptr() == hit.data_ptr():
                return hit
            out.copy_(hit)
            return out
        if frozen is not None and plain:
            frozen["fitted"] = self._predictor_device(state, engine, out, None, 1.0, None)
            return frozen["fitted"]
        return self._predictor_device(state, engine, out, exclude, alpha, chain_scale)

    def _predictor_device(self, state: dict, engine, out, exclude, alpha, chain_scale):
        skip = [] if exclude is None else ([exclude] if isinstance(exclude, str) else list(exclude))
        host_sum, ident, batched, dense = 0, [], [], []
        for prm, prefactor in self.form.items():
            if prm in skip:
                continue
            A, v = state[prefactor], state[prm]
            if is_chain(A):
                if not is_chain(v) or v.shape[1] != 1:
                    raise NotImplementedError("a per-chain design matrix needs a per-chain (k, 1) coefficient vector")
                batched.append((A, v))
            elif not is_chain(v):
                host_sum = host_sum + A @ v
            elif v.shape[1] != 1:
                raise NotImplementedError("replicated parameters")
            elif _is_identity(A, v.shape[0]):
                ident.append(v.vector())
            else:
                dense.append((A, v))
        if not (ident or batched or dense):
            raise ValueError("no per-chain term: use predictor()")
        shared = None if isinstance(host_sum, int) else engine.to_device(np.asarray(host_sum, dtype=np.float64).reshape(-1))
        if len(batched) == 1 and len(ident) <= 1 and not dense:
            # the whole expression in one launch
            B, v = batched[0]
            cs = None if (alpha == 1.0 and chain_scale is None) else _scaled(engine, chain_scale, alpha)
            return engine.design_predict_batched(B.columns(), v.vector(), add_chain=ident[0] if ident else None,
                                                 add_shared=shared, chain_scale=cs, out=out)
        # several terms: summed with the library's a x + b y (omc_chain_lincomb; the state's own vectors are never written)
        fitted, own = None, False  # own: `fitted` is a buffer of this call, free to be overwritten
        for t in ident:
            if fitted is None:
                fitted = t
            else:
                fitted, own = engine.chain_lincomb(1.0, fitted, 1.0, t, out=fitted if own else None), True
        for A, v in dense:
            term = engine.design_predict(engine.shared(A), v.vector(), out if fitted is None else None)
            if fitted is None:
                fitted, own = term, True
            else:
                fitted, own = engine.chain_lincomb(1.0, term, 1.0, fitted, out=term), True
        for B, v in batched:
            fitted, own = engine.design_predict_batched(B.columns(), v.vector(), add_chain=fitted), True
        if shared is not None:
            fitted, own = engine.chain_lincomb(1.0, fitted, 1.0, shared, out=fitted if own else None), True
        if alpha != 1.0 or chain_scale is not None:
            # s_c * fitted_c: the identity-matrix case of the per-chain scaled product
            fitted, own = engine.tridiag_matvec_chain(fitted.shape[1], None, None, fitted, scale=_scaled(engine, chain_scale, alpha)), True
        if out is not None and fitted.data_ptr() != out.data_ptr():
            engine.chain_copy(fitted, out) if (fitted.stride(1) == 1 and out.stride(1) == 1) else out.copy_(fitted)
            fitted = out
        return fitted

    def resid_sq_device(self, state: dict, engine, y, w=None):
        """(C,) tensor sum_i w_i (y_i - f_i)^2 for f = predictor, in one pass over the basis without the fitted
        values in memory -- what the acceptance ratio of a knot move needs of a regression likelihood.  Only for the
        form predictor_device evaluates in one launch (one per-chain design matrix, at most one per-chain offset,
        shared terms); None otherwise."""
        host_sum, ident, batched = 0, [], []
        for prm, prefactor in self.form.items():
            A, v = state[prefactor], state[prm]
            if is_chain(A):
                if not is_chain(v) or v.shape[1] != 1:
                    return None
                batched.append((A, v))
            elif not is_chain(v):
                host_sum = host_sum + A @ v
            elif v.shape[1] == 1 and _is_identity(A, v.shape[0]):
                ident.append(v.vector())
            else:
                return None
        if len(batched) != 1 or len(ident) > 1:
            return None
        shared = None if isinstance(host_sum, int) else engine.to_device(np.asarray(host_sum, dtype=np.float64).reshape(-1))
        B, v = batched[0]
        return engine.design_resid_sq_batched(B.columns(), v.vector(), y, add_chain=ident[0] if ident else None,
                                              add_shared=shared, w=w)

    def has_chain_terms(self, state: dict, exclude=None) -> bool:
        """Is any term other than `exclude` per chain (a per-chain offset in the response mean)?"""
        skip = [] if exclude is None else ([exclude] if isinstance(exclude, str) else list(exclude))
        return any(prm not in skip and (is_chain(state[prm]) or is_chain(state[pre])) for prm, pre in self.form.items())

    def get_param_list(self) -> list:
        return list(self.form.keys()) + list(self.form.values())

    def get_grad_param_list(self) -> list:
        return list(self.form.keys())


@dataclass
class ScaledMatrix(Parameter):
    """f = state[scalar] * state[matrix]  (parameter.py:300-373).  On the GPU path the matrix is a
    shared host constant and the scalar is per chain; the product is never formed."""

    matrix: str
    scalar: str

    def predictor(self, state: dict):
        s = state[self.scalar]
        if is_chain(s):
            raise NotImplementedError("ScaledMatrix.predictor with a per-chain scalar is consumed structurally "
                                      "(matrix, scalar) by the GPU samplers; it is not materialised")
        return float(np.asarray(s).item()) * state[self.matrix]

    def get_param_list(self) -> list:
        return [self.scalar, self.matrix]

    def get_grad_param_list(self) -> list:
        return [self.scalar]

    def precision_unscaled(self, state: dict, _) -> np.ndarray:
        return state[self.matrix]


_IDENTITY_MEMO = {}


def _is_identity(A, n):
    """Is the (shared, immutable) matrix A the n x n identity?  Memoised per object: samplers ask
    every sweep."""
    from scipy import sparse

    if is_chain(A) or getattr(A, "shape", None) != (n, n):
        return False
    hit = _IDENTITY_MEMO.get(id(A))
    if hit is not None and hit[0] is A:
        return hit[1]
    if sparse.issparse(A):
        ans = (A - sparse.identity(n)).nnz == 0
    else:
        ans = bool(np.array_equal(np.asarray(A), np.eye(n)))
    _IDENTITY_MEMO[id(A)] = (A, ans)
    return ans


def _scaled(engine, chain_scale, alpha):
    """(C,) tensor alpha * chain_scale (chain_scale None = ones)."""
    if chain_scale is None:
        return engine.full((engine.n_chains,), alpha)
    if alpha == 1.0:
        return chain_scale
    cs = chain_scale.reshape(-1, 1)  # the library's own a x + b y on the (C, 1) column: alpha * s + 0 * s
    return engine.chain_lincomb(alpha, cs, 0.0, cs).reshape(-1)


@dataclass
class MixtureParameter(Parameter, ABC):
    """Parameter whose elements are picked from a short vector by an allocation (parameter.py:376-417).
    On the GPU path `param` is a shared host vector (m, 1) and `allocation` a per-chain, usually ragged,
    ChainArray of integer values (k, 1): the reference grows/shrinks it with the jump parameter
    (tests/test_reversible_jump.py:86-87, 113-114)."""

    param: str
    allocation: str

    def get_param_list(self) -> list:
        return [self.param, self.allocation]

    def gather_device(self, state: dict, engine, fill: float):
        """(C, kmax) tensor param[allocation] with `fill` beyond each chain's live length."""
        alloc, par = state[self.allocation], state[self.param]
        if not is_chain(alloc):
            raise NotImplementedError("mixture parameters need a per-chain allocation")
        table = par.vector() if is_chain(par) else engine.shared(par).reshape(-1)  # (C, m) per chain or (m,) shared
        return engine.mixture_gather(table, alloc.vector(), count=alloc.count(state), fill=fill)


@dataclass
class MixtureParameterVector(MixtureParameter):
    """predictor = param[allocation]  (parameter.py:420-471)."""

    def predictor(self, state: dict):
        alloc = state[self.allocation]
        if is_chain(alloc):
            raise NotImplementedError("per-chain allocation: use gather_device (consumed structurally by the samplers)")
        return state[self.param][np.asarray(alloc).astype(int).flatten()]

    def get_grad_param_list(self) -> list:
        return [self.param]


@dataclass
class MixtureParameterMatrix(MixtureParameter):
    """predictor = diag(param[allocation])  (parameter.py:474-538); never formed on the GPU path."""

    def predictor(self, state: dict):
        from scipy import sparse

        alloc = state[self.allocation]
        if is_chain(alloc):
            raise NotImplementedError("per-chain allocation: use gather_device (consumed structurally by the samplers)")
        return sparse.diags(diagonals=state[self.param][np.asarray(alloc).astype(int)].flatten(), offsets=0, format="csc")

    def get_grad_param_list(self) -> list:
        return []
