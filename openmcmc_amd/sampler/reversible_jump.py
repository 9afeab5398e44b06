"""Reversible-jump sampler on chain-batched, ragged state (reference sampler/reversible_jump.py).

Every chain makes its own birth or death move in the same launches:
  omc_rj_move                 move type, p_birth / p_death, deletion index          (reversible_jump.py:310-373, 173)
  Distribution.rvs            the new element of each associated parameter          (:130)
  omc_ragged_resize           np.concatenate / np.delete on the padded arrays       (:131, :175)
  user callbacks              state_birth_function / state_death_function           (:133-134, :178-181)
  omc_design_gram_batched +
  omc_rj_matched_transition   matched coefficient transition and its densities      (:195-308)
  Model.log_p x 2, omc_mh_accept, omc_chain_select                                  (metropolis_hastings.py:127-173)

Batched callback convention (the reference calls ONE of the two callbacks for its single chain):
  * `state_birth_function(current_state, prop_state)` is called every sweep and must bring the dependent entries
    of prop_state in line with the resized associated parameters for EVERY chain -- recomputing a basis matrix
    from the proposed knots does that for births and deaths alike;
  * `state_death_function(current_state, prop_state, deletion_index)`, if given, is called after it with the (C,)
    int64 deletion indices (-1 on chains that make a birth) for models that delete rather than recompute;
  * `prop_state["__rj_move__"]` holds {"birth": int32 (C,), "deletion_index": int64 (C,)} for both.
Allocation vectors of mixture parameters tied to the jump parameter (tests/test_reversible_jump.py:86-87) stay
valid without a callback when they are all-zero: the padding already holds zeros.
"""

from dataclasses import dataclass
from typing import Callable, Union

import numpy as np

from openmcmc_amd.chains import ChainArray, is_chain
from openmcmc_amd.distribution.distribution import Gamma, Uniform
from openmcmc_amd.distribution.location_scale import Normal
from openmcmc_amd.sampler.metropolis_hastings import MetropolisHastings, _add_contribution

# Philox sub-streams within one ReversibleJump.sample call (block numbers; omc_rj_move owns 0..63)
_SUB_ASSOCIATED, _SUB_MATCH, _SUB_ACCEPT = 64, 96, 97


@dataclass
class ReversibleJump(MetropolisHastings):
    """reversible_jump.py:24-373; `param` is a per-chain (1, 1) count, `associated_params` ragged ChainArrays whose
    live length it is."""

    associated_params: Union[list, str, None] = None
    n_max: Union[int, None] = None
    birth_probability: float = 0.5
    state_birth_function: Union[Callable, None] = None
    state_death_function: Union[Callable, None] = None
    matching_params: Union[dict, None] = None

    def __post_init__(self):
        # reversible_jump.py:66-74: the WHOLE model stays attached
        self._init_runtime()
        self.step = np.array(self.step, ndmin=2)
        self.inject_uniform = None
        self.trace = None
        # test hooks: callables (sampler, sweep) -> device tensors replaying the reference's draws
        self.inject_move = None       # (u (C,), deletion index int64 (C,))
        self.inject_associated = None  # {key: uniforms (C, p)}
        self.inject_match = None      # (C,) uniform (limits) or normal draw of the matched coefficient
        if isinstance(self.associated_params, str):
            self.associated_params = [self.associated_params]
        if self.associated_params is None:
            self.associated_params = []

    # ------------------------------------------------------------------ proposal
    def proposal(self, current_state: dict, param_index: int = None):
        eng = self._need_engine()
        n_cur = current_state[self.param]
        if not is_chain(n_cur) or n_cur.size != 1:
            raise NotImplementedError("the jump parameter must be a per-chain (1, 1) count")
        count = n_cur.scalar()
        di, t = self._draw_index(), self._sweep
        u_move, idx = self.inject_move(self, t) if self.inject_move is not None else (None, None)
        # log_p(current_state, by_observation=True)[-1] of every associated parameter (:132, :143): the density of the
        # LAST element of the CURRENT value -- a constant for a Uniform prior, a per-chain number for a Gamma one
        dens_const, dens_chain = 0.0, None
        for key in self.associated_params:
            dist, cur = self.model[key], current_state[key]
            if not is_chain(cur) or cur.ragged is None or cur.ragged[0] != self.param:
                raise NotImplementedError(f"associated parameter '{key}' must be a ragged ChainArray counted by '{self.param}'")
            if not isinstance(dist, (Uniform, Gamma, Normal)) or getattr(dist, "is_mixture", False):
                raise NotImplementedError("associated parameters need a Uniform, Gamma, Normal or LogNormal prior")
            d = dist.log_p_last(current_state, eng)
            if hasattr(d, "data_ptr"):
                dens_chain = d if dens_chain is None else dens_chain + d
            else:
                dens_const += float(d)
        # move type, deletion index, proposed count and the move's part of the two proposal densities
        # (reversible_jump.py:142-144 for a birth, 189-191 for a death) in one launch; the callbacks' and the matched
        # transition's contributions are added to lq_f / lq_r below
        birth, del_index, count_prop, lq_f, lq_r = eng.rj_move_densities(
            count, int(self.n_max), float(self.birth_probability), density_chain=dens_chain, density_const=dens_const,
            u=u_move, idx=idx, draw_index=di)
        prop_state = dict(current_state)
        prop_state[self.param] = ChainArray(count_prop.reshape(-1, 1, 1))
        prop_state["__rj_move__"] = {"birth": birth, "deletion_index": del_index}
        inj_assoc = self.inject_associated(self, t) if self.inject_associated is not None else {}
        for j, key in enumerate(self.associated_params):
            dist, cur = self.model[key], current_state[key]
            new = dist.rvs(current_state, n=1, engine=eng, draw_index=di, sub=_SUB_ASSOCIATED + 2 * j,
                           inject=inj_assoc.get(key))  # reversible_jump.py:130
            prop_state[key] = cur.like(eng.ragged_resize(cur.data, count, birth, del_index, axis=cur.ragged[1],
                                                         new_vals=new.data.reshape(eng.n_chains, -1)))
        if callable(self.state_birth_function):
            prop_state, f_extra, r_extra = self.state_birth_function(current_state, prop_state)
            lq_f, lq_r = _add_contribution(eng, lq_f, f_extra), _add_contribution(eng, lq_r, r_extra)
        if callable(self.state_death_function):
            prop_state, f_extra, r_extra = self.state_death_function(current_state, prop_state, del_index)
            lq_f, lq_r = _add_contribution(eng, lq_f, f_extra), _add_contribution(eng, lq_r, r_extra)
        if self.matching_params is not None:
            prop_state = self._matched_transition(current_state, prop_state, count, birth, del_index, lq_f, lq_r)
        del prop_state["__rj_move__"]
        if self.trace is not None:
            self.trace.update(birth=birth, deletion_index=del_index, lq_fwd=lq_f.clone(), lq_rev=lq_r.clone(),
                              prop={k: prop_state[k].data.clone() for k in self._changed_keys()})
        return prop_state, lq_f, lq_r

    def _changed_keys(self):
        keys = list(self.associated_params)
        if self.matching_params is not None:
            keys.append(self.matching_params["variable"])
        return keys

    def _matched_transition(self, current_state, prop_state, count, birth, del_index, lq_f, lq_r):
        """reversible_jump.py:195-308 for every chain (omc_rj_matched_transition)."""
        eng = self.engine
        vector, matrix = self.matching_params["variable"], self.matching_params["matrix"]
        scale, limits = float(self.matching_params["scale"]), self.matching_params["limits"]
        coef, B_cur, B_prop = current_state[vector], current_state[matrix], prop_state[matrix]
        if not (is_chain(coef) and is_chain(B_cur) and is_chain(B_prop)) or coef.shape[1] != 1:
            raise NotImplementedError("matched transitions need per-chain coefficient vector and basis matrices")
        # X'X of the larger basis only: the proposed one where a chain makes a birth, the current one for a death
        gram = eng.design_gram_select(B_cur.columns(), B_cur.count(current_state), B_prop.columns(), B_prop.count(prop_state), birth)
        inject = self.inject_match(self, self._sweep) if self.inject_match is not None else None
        out = eng.rj_matched_transition(gram, gram, count, birth, del_index, coef.vector(), scale, limits, lq_f, lq_r,
                                        inject=inject, draw_index=self._draw_index(), sub=_SUB_MATCH)
        prop_state[vector] = coef.like(out.unsqueeze(2))
        return prop_state

    # ------------------------------------------------------------------ sample
    def sample(self, current_state: dict) -> dict:
        prop_state, lq_f, lq_r = self.proposal(current_state)
        u = self.inject_uniform(self, self._sweep) if self.inject_uniform is not None else None
        current_state = self._accept_reject_proposal(current_state, prop_state, lq_f, lq_r, u=u, sub=_SUB_ACCEPT,
                                                     trace=self.trace)
        self._sweep += 1
        return current_state
