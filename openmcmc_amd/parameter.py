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
                    raise NotImplementedError("per-chain term with a non-identity design matrix (dense path: next round)")
                chain_sum = v if chain_sum is None else ChainArray(chain_sum.data + v.data)
            else:
                host_sum = host_sum + A @ v
        if chain_sum is None:
            return host_sum
        if isinstance(host_sum, int):
            return chain_sum
        import torch

        return ChainArray(chain_sum.data + torch.as_tensor(np.asarray(host_sum), device=chain_sum.data.device))

    def predictor_device(self, state: dict, engine, out=None):
        """(C, n) fitted values sum_i A_i x_i with per-chain terms evaluated on the GPU (one GEMM per
        dense design matrix) and shared terms added once (parameter.py:174-197)."""
        host_sum, fitted = 0, None
        for prm, prefactor in self.form.items():
            A, v = state[prefactor], state[prm]
            if not is_chain(v):
                host_sum = host_sum + A @ v
                continue
            if v.shape[1] != 1:
                raise NotImplementedError("replicated parameters")
            if _is_identity(A, v.shape[0]):
                term = v.vector()
            else:
                term = engine.design_predict(engine.shared(A), v.vector(), out if fitted is None else None)
            fitted = term if fitted is None else fitted + term
        if fitted is None:
            raise ValueError("no per-chain term: use predictor()")
        if not isinstance(host_sum, int):
            fitted = fitted + engine.to_device(np.asarray(host_sum, dtype=np.float64).reshape(1, -1))
        if out is not None and fitted.data_ptr() != out.data_ptr():
            out.copy_(fitted)
            fitted = out
        return fitted

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

    if getattr(A, "shape", None) != (n, n):
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
