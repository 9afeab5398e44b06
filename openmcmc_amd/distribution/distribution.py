"""Distribution base class and the Gamma prior (reference distribution/distribution.py:28-278).

log_p on the GPU path returns one value per chain: a (C,) device tensor.
"""

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Union

import numpy as np

from openmcmc_amd.chains import ChainArray, is_chain
from openmcmc_amd.parameter import Identity, LinearCombination


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
        """Scalar prior parameters; they are host constants in every supported model."""
        a, b = self.shape.predictor(state), self.rate.predictor(state)
        if is_chain(a) or is_chain(b) or np.size(a) != 1 or np.size(b) != 1:
            raise NotImplementedError("Gamma prior parameters must be shared scalars on the GPU path")
        return float(np.asarray(a).item()), float(np.asarray(b).item())

    def log_p(self, state: dict, by_observation: bool = False, engine=None, out=None, accumulate=False):
        """distribution.py:241-261.  Per-chain response -> (C,) tensor via omc_gamma_logpdf."""
        x = state[self.response]
        a, b = self.host_shape_rate(state)
        if not is_chain(x):
            from math import lgamma, log

            v = float(np.asarray(x).item())
            return a * log(b) - lgamma(a) + (a - 1) * log(v) - b * v
        if engine is None:
            raise RuntimeError("Gamma.log_p on a per-chain response needs the engine (use Model.log_p)")
        out = engine.empty(engine.n_chains) if out is None else out
        engine.gamma_logpdf(x.scalar(), a, b, out, accumulate=accumulate)
        return out

    def rvs(self, state, n: int = 1, engine=None, draw_index=0):
        """distribution.py:263-278: prior draw, used by MCMC when the state has no initial value."""
        if engine is None:
            raise RuntimeError("Gamma.rvs needs the engine")
        if n != 1:
            raise NotImplementedError("replicated prior draws")
        a, b = self.host_shape_rate(state)
        out = engine.empty(engine.n_chains)
        # Gamma(a, rate b) = the conjugate update with no data: n_pos = 0, quad = 0
        engine.normal_gamma_update(a, b, 0, engine.zeros(engine.n_chains), out, draw_index=draw_index)
        return ChainArray(out.reshape(-1, 1, 1))
