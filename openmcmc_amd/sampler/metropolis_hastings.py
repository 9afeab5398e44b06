"""Metropolis-Hastings samplers on chain-batched state (reference sampler/metropolis_hastings.py).

Built this round: RandomWalk (untruncated) and ManifoldMALA for a parameter whose conditional model
is one Normal with the parameter as response, a shared mean and a shared dense precision
(Normal("x", mean="mu", precision="Q") -- BASELINE configs[3]).  Because the Hessian of that target
is constant, chol(H/step^2) is factorised once per sampler instead of five times per update
(SURVEY.md section 3.4); every chain then advances with level-3 BLAS on the d x C state matrix.
"""

from dataclasses import dataclass, field
from typing import Callable

import numpy as np

from openmcmc_amd.chains import ChainArray, is_chain
from openmcmc_amd.distribution.location_scale import Normal
from openmcmc_amd.parameter import Identity
from openmcmc_amd.sampler.sampler import MCMCSampler


class AcceptRate:
    """Acceptance counters (metropolis_hastings.py:25-66), one pair per chain on the device (int64);
    `count` reports the totals over chains, as the reference's dict does for its single chain."""

    def __init__(self):
        self.accept = None
        self.proposal = None

    def attach(self, engine):
        import torch

        self.accept = torch.zeros(engine.n_chains, dtype=torch.int64, device=engine.device)
        self.proposal = torch.zeros(engine.n_chains, dtype=torch.int64, device=engine.device)

    @property
    def count(self):
        if self.accept is None:
            return {"accept": 0, "proposal": 0}
        return {"accept": int(self.accept.sum().item()), "proposal": int(self.proposal.sum().item())}

    @property
    def acceptance_rate(self) -> float:
        c = self.count
        return c["accept"] / c["proposal"] * 100

    def get_acceptance_rate(self) -> str:
        if self.count["proposal"] == 0:
            return "No proposals"
        return f"Acceptance rate {self.acceptance_rate:.0f}%"


@dataclass
class MetropolisHastings(MCMCSampler):
    """Base class (metropolis_hastings.py:69-173): `step`, `accept_rate`."""

    step: np.ndarray = field(default_factory=lambda: np.array([0.2], ndmin=2), init=True)
    accept_rate: AcceptRate = field(default_factory=lambda: AcceptRate(), init=False)

    def __post_init__(self):
        super().__post_init__()
        self.step = np.array(self.step, ndmin=2)
        self.inject_uniform = None  # test hook like `inject`, for the accept/reject uniforms

    def bind(self, engine, position=0, n_samplers=1):
        super().bind(engine, position, n_samplers)
        self.accept_rate.attach(engine)
        return self

    def _target(self, state):
        """(Q device, mu device or None, d) of the Gaussian target, checking the supported structure."""
        eng = self._need_engine()
        if list(self.model.keys()) != [self.param]:
            raise NotImplementedError("MH samplers are built for a parameter whose conditional model is its own Normal")
        dist = self.model[self.param]
        if not isinstance(dist, Normal) or not isinstance(dist.precision, Identity) or not isinstance(dist.mean, Identity):
            raise NotImplementedError("MH target must be Normal(param, mean=<shared>, precision=<shared matrix>)")
        Q, mu = state[dist.precision.form], state[dist.mean.form]
        if is_chain(Q) or is_chain(mu):
            raise NotImplementedError("per-chain target parameters")
        if np.size(self.step) != 1:
            raise NotImplementedError("per-element step sizes")
        d = Q.shape[0]
        mu = np.asarray(mu, dtype=np.float64).reshape(-1)
        return eng.shared(Q), (eng.shared(mu) if mu.any() else None), d

    def _x(self, state):
        v = state[self.param]
        if not is_chain(v) or v.shape[1] != 1:
            raise NotImplementedError("MH samplers need a per-chain (d, 1) parameter")
        return v.vector()


@dataclass
class RandomWalk(MetropolisHastings):
    """Gaussian random-walk proposal (metropolis_hastings.py:176-269), untruncated."""

    domain_limits: np.ndarray = None
    state_update_function: Callable = None

    def sample(self, current_state: dict) -> dict:
        eng = self._need_engine()
        if self.domain_limits is not None or self.state_update_function is not None:
            raise NotImplementedError("truncated proposals / state_update_function: later round")
        Q, mu, d = self._target(current_state)
        if self._plan is None:
            self._plan = eng.dense_cholesky(Q, 1.0)  # chol(Q) for log p (gmrf.py:339)
        LQ, sl = self._plan
        z = self.inject(self, self._sweep) if self.inject is not None else None
        u = self.inject_uniform(self, self._sweep) if self.inject_uniform is not None else None
        eng.rw_step(mu, LQ, sl, float(self.step.item()), self._x(current_state), z=z, u=u,
                    draw_index=self._draw_index(), accept_count=self.accept_rate.accept,
                    proposal_count=self.accept_rate.proposal)
        self._sweep += 1
        return current_state


@dataclass
class ManifoldMALA(MetropolisHastings):
    """Manifold MALA (Girolami & Calderhead 2011; metropolis_hastings.py:292-373)."""

    def sample(self, current_state: dict) -> dict:
        eng = self._need_engine()
        Q, mu, d = self._target(current_state)
        step = float(self.step.item())
        if self._plan is None:
            self._plan = eng.dense_cholesky(Q, 1.0 / step**2)  # chol(H/step^2), H = Q (metropolis_hastings.py:345-346)
        L, sl = self._plan
        z = self.inject(self, self._sweep) if self.inject is not None else None
        u = self.inject_uniform(self, self._sweep) if self.inject_uniform is not None else None
        eng.mala_step(Q, mu, L, sl, step, self._x(current_state), z=z, u=u, draw_index=self._draw_index(),
                      accept_count=self.accept_rate.accept, proposal_count=self.accept_rate.proposal)
        self._sweep += 1
        return current_state
