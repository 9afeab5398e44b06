"""Metropolis-Hastings samplers on chain-batched state (reference sampler/metropolis_hastings.py).

Two routes:
  * fused: RandomWalk (untruncated) and ManifoldMALA for a parameter whose conditional model is one Normal with
    the parameter as response, a shared mean and a shared dense precision (Normal("x", mean="mu", precision="Q")
    -- BASELINE configs[3]).  The Hessian of that target is constant, so chol(H/step^2) is factorised once per
    sampler instead of five times per update and every chain advances with level-3 BLAS
    on the d x C state matrix (omc_mala_step / omc_rw_step).
  * generic: any model whose log_p the device can evaluate.  The proposal (omc_rw_propose: Gaussian or truncated
    Gaussian random walk), the model log-density of the current and the proposed state, the accept test
    (omc_mh_accept) and the per-chain merge of the two states (omc_chain_select) are separate launches over all
    chains; `state_update_function` callbacks receive and return chain-batched state.  This is the route of
    RandomWalkLoop over the knots of a basis and of ReversibleJump (BASELINE configs[4]).
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
        self.trace = None  # test hook: a dict that receives the internals of the last proposal / decision

    def bind(self, engine, position=0, n_samplers=1):
        super().bind(engine, position, n_samplers)
        self.accept_rate.attach(engine)
        self._white_tag, self._white_serial = None, 0  # a sampler object taken into another run starts without a cached state
        return self

    def _white_state_tag(self, state, x, L, mu):
        """What the library's cached whitened state a = L'(x - mu) of the fused steps belongs to: this state entry (the object,
        not only its address -- a fresh tensor can land on a recycled address with the same version), its contents as far as
        torch sees them (version counter), the factor and the mean."""
        return (id(state[self.param]), x.data_ptr(), x._version, L.data_ptr(), L._version,
                None if mu is None else (mu.data_ptr(), mu._version))

    def _white_state_is_current(self, engine, x, tag):
        """... and as far as the library's own writes go, which torch's version counter does not see (Engine.note_write)."""
        return getattr(self, "_white_tag", None) == tag and not engine.written_since(x, getattr(self, "_white_serial", 0))

    def _log_p_buffer(self, engine):
        """(C,) device buffer the fused steps leave the target's log density in."""
        buf = getattr(self, "_log_p_buf", None)
        if buf is None or buf.shape[0] != engine.n_chains:
            buf = self._log_p_buf = engine.empty(engine.n_chains)
        return buf

    def _gaussian_target(self, state):
        """Does the fused route apply?  (one Normal with the parameter as response, shared mean and precision)"""
        if list(self.model.keys()) != [self.param] or np.size(self.step) != 1:
            return False
        dist = self.model[self.param]
        if not isinstance(dist, Normal) or not isinstance(dist.precision, Identity) or not isinstance(dist.mean, Identity):
            return False
        return not (is_chain(state[dist.precision.form]) or is_chain(state[dist.mean.form]))

    def _target(self, state):
        """(Q device, mu device or None, d) of the Gaussian target of the fused route."""
        eng = self._need_engine()
        if not self._gaussian_target(state):
            raise NotImplementedError("fused MH route needs Normal(param, mean=<shared>, precision=<shared matrix>) "
                                      "and a scalar step")
        dist = self.model[self.param]
        Q, mu = state[dist.precision.form], state[dist.mean.form]
        # device copies are cached by the identity of the STATE ENTRY (a reshaped copy would be a new object every step:
        # one upload per step, and a new address for everything the library derives from the mean)
        mu_dev = eng.shared(mu).reshape(-1) if np.any(mu) else None
        return eng.shared(Q), mu_dev, Q.shape[0]

    def _factor_plan(self, eng, state, Q, scale):
        """chol(scale * Q) of the fused routes, kept from step to step (the reference refactorises every step,
        metropolis_hastings.py:345-346) but rebuilt whenever what it was made from changes: another precision array in
        the state, or another step size.  The library keeps matrices derived from the factor (drift matrix, L^-T) keyed
        by device addresses; a rebuilt factor may land on a recycled address, so that cache is dropped as well."""
        key = (float(scale), id(state[self.model[self.param].precision.form]), int(Q.data_ptr()), int(Q._version))
        if self._plan is None or getattr(self, "_plan_key", None) != key:
            eng.mh_invalidate()
            self._plan = eng.dense_cholesky(Q, scale)
            self._plan_key = key
        return self._plan

    def _x(self, state):
        v = state[self.param]
        if not is_chain(v) or v.shape[1] != 1:
            raise NotImplementedError("MH samplers need a per-chain (d, 1) parameter")
        return v.vector()

    # ------------------------------------------------------------------ generic route
    def _accept_reject_proposal(self, current_state: dict, prop_state: dict, logp_pr_g_cr, logp_cr_g_pr, count=None,
                                index=0, u=None, sub=0, trace=None, lp_cur=None) -> dict:
        """metropolis_hastings.py:127-173 for every chain at once: log_alpha = lp' + q_rev - (lp + q_fwd), accept iff
        log U < log_alpha, then current_state takes the proposed value of every entry that differs, chain by chain.
        `count`/`index` gate the chains taking part (a loop over the columns of a ragged parameter).
        `lp_cur`: the model log-density of current_state if the caller already has it (the reference recomputes it
        for every proposal; inside a loop over columns it is the value selected at the previous step); it is
        updated in place to the log-density of the returned state and kept in self._lp_state."""
        eng = self._need_engine()
        # Only distributions that read an entry the proposal replaced can differ between the two states; the others
        # contribute the same number to both log-densities (the reference sums them all and subtracts,
        # metropolis_hastings.py:150-155).  Proposals and callbacks must REPLACE state entries, never mutate them.
        changed = [k for k, v in prop_state.items() if v is not current_state.get(k)]
        members = self.model.affected_by(changed)
        if lp_cur is None or getattr(self, "_lp_members", None) != members:
            lp_cur = self.model.log_p(current_state, engine=eng, members=members)
        self._lp_members = members
        lp_prop = self.model.log_p(prop_state, engine=eng, members=members)
        log_alpha = eng.empty(eng.n_chains) if trace is not None else None
        acc = eng.mh_accept(lp_cur, lp_prop, _as_chain_tensor(eng, logp_pr_g_cr), _as_chain_tensor(eng, logp_cr_g_pr),
                            count=count, index=index, u=u, draw_index=self._draw_index(), sub=sub,
                            accept_count=self.accept_rate.accept, proposal_count=self.accept_rate.proposal,
                            log_alpha=log_alpha)
        if trace is not None:
            trace.update(log_alpha=log_alpha, accept=acc, lp_cur=lp_cur.clone(), lp_prop=lp_prop)
        pairs = []
        for key, value in prop_state.items():
            cur = current_state.get(key)
            if value is cur or not is_chain(value):
                continue
            if not is_chain(cur) or cur.data.shape != value.data.shape:
                raise NotImplementedError(f"proposed state entry '{key}' changes kind or padded shape")
            pairs.append((value.storage(), cur.storage()))
        pairs.append((lp_prop, lp_cur))
        eng.chain_select_many(acc, pairs)  # every changed entry and the log-density in one launch
        self._lp_state = lp_cur
        return current_state


def _as_chain_tensor(engine, value):
    """Proposal log-density contributions may be Python floats (0.0 from a callback) or (C,) tensors."""
    if value is None or hasattr(value, "data_ptr"):
        return value
    v = float(value)
    return None if v == 0.0 else engine.full((engine.n_chains,), v)


def _lin(engine, a, x, b, y):
    """a x + b y per chain by the library's own kernel (omc_chain_lincomb); x, y: (C, n) or (C,) tensors.  With a, b in
    {+-1, +-0.5} both products are exact, so this is the one-rounding sum an element-wise a*x + b*y gives."""
    if x.dim() == 1:
        return engine.chain_lincomb(a, x.reshape(-1, 1), b, y.reshape(-1, 1)).reshape(-1)
    return engine.chain_lincomb(a, x if x.stride(1) == 1 else x.contiguous(), b, y if y.stride(1) == 1 else y.contiguous())


def _add_contribution(engine, total, extra):
    """total (a (C,) tensor) += extra (float or (C,) tensor)."""
    if hasattr(extra, "data_ptr"):
        total += extra
    elif float(extra) != 0.0:
        total += float(extra)
    return total


@dataclass
class RandomWalk(MetropolisHastings):
    """(Truncated) Gaussian random-walk proposal (metropolis_hastings.py:176-269).

    `state_update_function(prop_state, param_index) -> (prop_state, logq_fwd_extra, logq_rev_extra)` is called with
    chain-batched state (ChainArray entries) and may return floats or (C,) tensors for the two extras."""

    domain_limits: np.ndarray = None
    state_update_function: Callable = None

    def __post_init__(self):
        # metropolis_hastings.py:201-210: keep the FULL model when a state_update_function may change other entries
        if self.state_update_function is None:
            self.model = self.model.conditional(self.param)
        self._init_runtime()
        self.step = np.array(self.step, ndmin=2)
        self.inject_uniform = None
        self.trace = None  # test hook: a dict that receives the internals of the last proposal / decision

    def _blocks_per_proposal(self, p):
        return (p + 1) // 2 + 1  # Philox blocks for p proposal draws, plus one for the accept uniform

    def proposal(self, current_state: dict, param_index: int = None, inject=None):
        """metropolis_hastings.py:212-269 for every chain; returns (prop_state, logq_fwd (C,), logq_rev (C,))."""
        eng = self._need_engine()
        x = current_state[self.param]
        if not is_chain(x):
            raise NotImplementedError("RandomWalk needs a per-chain parameter")
        p, n_rep = x.shape
        if param_index is None and n_rep != 1:
            raise NotImplementedError("whole-matrix proposals for a replicated parameter: use RandomWalkLoop")
        step = self.step
        if step.shape[1] != 1:
            if param_index is None:
                raise NotImplementedError("(p, n) step sizes without a column index")
            step = step[:, [param_index]]
        cache = getattr(self, "_dev_consts", None)
        if cache is None:
            cache = self._dev_consts = {}
        skey = ("step", 0 if self.step.shape[1] == 1 else param_index)
        if skey not in cache:
            cache[skey] = eng.to_device(np.ascontiguousarray(step.reshape(-1)))
        if self.domain_limits is not None and "lim" not in cache:
            lim = np.asarray(self.domain_limits, dtype=np.float64).reshape(-1, 2)
            if lim.shape[0] != p:
                raise ValueError("domain_limits must have one (lower, upper) row per element of the parameter")
            cache["lim"] = (eng.to_device(lim[:, 0].copy()), eng.to_device(lim[:, 1].copy()))
        lower, upper = cache.get("lim", (None, None))
        col = 0 if param_index is None else int(param_index)
        ragged_cols = x.ragged is not None and x.ragged[1] == 1
        data = x.data if x.data.is_contiguous() else x.data.contiguous()
        z = data.clone()
        sub = col * self._blocks_per_proposal(p)
        lq_f, lq_r = eng.rw_propose(data, z, cache[skey], lower, upper, column=col if n_rep > 1 or ragged_cols else None,
                                    count=x.count(current_state) if ragged_cols else None, inject=inject,
                                    draw_index=self._draw_index(), sub=sub)
        prop_state = dict(current_state)  # shallow: only the entries a proposal replaces are new objects
        prop_state[self.param] = x.like(z)
        if callable(self.state_update_function):
            prop_state, f_extra, r_extra = self.state_update_function(prop_state, param_index)
            lq_f, lq_r = _add_contribution(eng, lq_f, f_extra), _add_contribution(eng, lq_r, r_extra)
        return prop_state, lq_f, lq_r

    def _generic_step(self, current_state, param_index=None, lp_cur=None):
        x = current_state[self.param]
        p = x.shape[0]
        col = 0 if param_index is None else int(param_index)
        inject = self.inject(self, self._sweep, param_index) if self.inject is not None else None
        u = self.inject_uniform(self, self._sweep, param_index) if self.inject_uniform is not None else None
        prop_state, lq_f, lq_r = self.proposal(current_state, param_index, inject=inject)
        ragged_cols = x.ragged is not None and x.ragged[1] == 1
        trace = None
        if self.trace is not None:
            trace = {"z": prop_state[self.param].data.clone(), "lq_fwd": lq_f.clone(), "lq_rev": lq_r.clone()}
            self.trace.setdefault("steps", []).append(trace)
        return self._accept_reject_proposal(current_state, prop_state, lq_f, lq_r,
                                            count=x.count(current_state) if ragged_cols else None, index=col, u=u,
                                            sub=col * self._blocks_per_proposal(p) + (p + 1) // 2, trace=trace,
                                            lp_cur=lp_cur)

    def sample(self, current_state: dict) -> dict:
        self.last_log_p = None  # set again by the fused whitened steps only
        eng = self._need_engine()
        if self.domain_limits is None and self.state_update_function is None and self._gaussian_target(current_state):
            Q, mu, d = self._target(current_state)
            LQ, sl = self._factor_plan(eng, current_state, Q, 1.0)  # chol(Q) for log p (gmrf.py:339)
            z = self.inject(self, self._sweep) if self.inject is not None else None
            u = self.inject_uniform(self, self._sweep) if self.inject_uniform is not None else None
            x = self._x(current_state)
            tag = self._white_state_tag(current_state, x, LQ, mu)  # see ManifoldMALA.sample: may the library reuse its L_Q'(x - mu)?
            eng.rw_step_white(mu, LQ, sl, float(self.step.item()), x, state_is_current=self._white_state_is_current(eng, x, tag),
                              z=z, u=u, draw_index=self._draw_index(), accept_count=self.accept_rate.accept,
                              proposal_count=self.accept_rate.proposal, log_p_out=self._log_p_buffer(eng))
            self._white_tag, self._white_serial = tag, eng._write_serial
            self.last_log_p = (self._log_p_buf, x)  # the target's log density at the state just left in x (see MCMC.run_mcmc)
        else:
            current_state = self._generic_step(current_state)
        self._sweep += 1
        return current_state


@dataclass
class RandomWalkLoop(RandomWalk):
    """One random-walk step per column of the parameter (metropolis_hastings.py:272-289).  For a ragged parameter
    the loop runs to the largest live length over the chains; chains with fewer columns sit the extra steps out.
    `fused` (default True): let one kernel launch run the whole loop when the model allows it (`_knot_plan`)."""

    fused: bool = True

    def _knot_plan(self, state):
        """Can the whole loop go to omc_knot_loop?  Yes when the sampler moves the knots of a library basis
        (state_update_function is a GaussianKnotBasis on this parameter) and the only distributions a move can change
        are the knots' own Uniform prior (constant over the proposal's domain) and ONE Normal regression likelihood
        whose mean carries the basis: shared response, diagonal precision with a per-chain scalar, mean =
        basis @ coefficients + at most one per-chain offset + shared terms.  Returns the kernel's arguments, or None
        (then the loop runs launch by launch through the callback)."""
        from openmcmc_amd.basis import GaussianKnotBasis
        from openmcmc_amd.distribution.distribution import Uniform
        from openmcmc_amd.parameter import LinearCombination, _is_identity

        basis = self.state_update_function
        if not isinstance(basis, GaussianKnotBasis) or basis.knots != self.param or self.trace is not None:
            return None
        x = state[self.param]
        if (not is_chain(x) or x.ragged is None or x.ragged[1] != 1 or x.shape[0] != 1 or not x.data.is_contiguous()
                or self.step.size != 1 or self.domain_limits is None):
            return None
        lim = np.asarray(self.domain_limits, dtype=np.float64).reshape(-1, 2)
        if lim.shape[0] != 1:
            return None
        lower, upper = float(lim[0, 0]), float(lim[0, 1])
        Bm = state.get(basis.matrix)
        if (not is_chain(Bm) or Bm.shape[1] != x.shape[1] or not Bm.data.transpose(1, 2).is_contiguous()
                or Bm.shape[0] > 10240):  # omc_knot_loop keeps a chain's residual in registers: n <= 10 240
            return None
        lik = None
        for key in self.model.affected_by([self.param, basis.matrix]):
            dist = self.model[key]
            if isinstance(dist, Uniform) and key == self.param:
                if not (np.all(dist.domain_response_lower <= lower) and np.all(dist.domain_response_upper >= upper)):
                    return None
            elif isinstance(dist, Normal) and not dist.is_mixture and lik is None and self.param not in dist.param_list:
                lik = dist
            else:
                return None
        if lik is None or not isinstance(lik.mean, LinearCombination):
            return None
        resp = state[lik.response]
        if is_chain(resp) or resp.shape[1] != 1:
            return None
        host_sum, offset, coef = 0, None, None
        for prm, pre in lik.mean.form.items():
            A, v = state[pre], state[prm]
            if pre == basis.matrix:
                if coef is not None or not is_chain(v) or v.shape[1] != 1:
                    return None
                coef = v.vector()
            elif is_chain(A):
                return None
            elif not is_chain(v):
                host_sum = host_sum + A @ v
            elif v.shape[1] == 1 and _is_identity(A, v.shape[0]) and offset is None:
                offset = v.vector()
            else:
                return None
        if coef is None:
            return None
        st = lik.structure(state)
        if st.diag is False or st.off is not None or (st.scale_key is not None and not is_chain(state[st.scale_key])):
            return None
        eng = self._need_engine()
        return {"basis": basis, "x": x, "B": Bm, "coef": coef, "offset": offset, "lower": lower, "upper": upper,
                "y": eng.shared(resp).reshape(-1), "w": None if st.diag is None else eng.shared(st.diag),
                "shared": None if isinstance(host_sum, int) else eng.to_device(np.asarray(host_sum, dtype=np.float64).reshape(-1)),
                "tau": state[st.scale_key].scalar() if st.scale_key is not None else None}

    def _knot_loop(self, current_state, plan):
        eng = self.engine
        x, Bm = plan["x"], plan["B"]
        kmax = x.shape[1]
        count = x.count(current_state)
        inj = {}
        if self.inject is not None or self.inject_uniform is not None:  # test hooks: one (C,) vector per column visited
            n_cols = int(count.max().item())
            for name, hook in (("inject_z", self.inject), ("inject_u", self.inject_uniform)):
                if hook is not None:
                    rows = eng.zeros(kmax, eng.n_chains)
                    for j in range(n_cols):
                        rows[j] = hook(self, self._sweep, j).reshape(-1)
                    inj[name] = rows
        eng.knot_loop(plan["basis"].X, plan["basis"].scale, plan["y"], Bm.data.transpose(1, 2), plan["coef"], x.data[:, 0, :],
                      count, float(self.step.item()), plan["lower"], plan["upper"], add_shared=plan["shared"],
                      add_chain=plan["offset"], w=plan["w"], tau=plan["tau"], draw_index=self._draw_index(),
                      accept_count=self.accept_rate.accept, proposal_count=self.accept_rate.proposal, **inj)
        # theta and the basis were updated in place; new wrappers, so that anything keyed on the identity of a state
        # entry sees them as replaced (the convention of every other sampler)
        current_state[self.param] = x.like(x.data)
        current_state[plan["basis"].matrix] = Bm.like(Bm.data)
        return current_state

    def sample(self, current_state: dict) -> dict:
        self.last_log_p = None  # set again by the fused whitened steps only
        plan = self._knot_plan(current_state) if self.fused else None
        if plan is not None:
            current_state = self._knot_loop(current_state, plan)
            self._sweep += 1
            return current_state
        x = current_state[self.param]
        n_cols = x.shape[1]
        if x.ragged is not None and x.ragged[1] == 1:
            n_cols = int(x.count(current_state).max().item())
        lp = None  # log-density of the current state, carried from one column to the next
        for param_index in range(n_cols):
            current_state = self._generic_step(current_state, param_index, lp_cur=lp)
            lp = self._lp_state
        self._sweep += 1
        return current_state


@dataclass
class ManifoldMALA(MetropolisHastings):
    """Manifold MALA (Girolami & Calderhead 2011; metropolis_hastings.py:292-373).  `whitened` (default True): the fused
    route for a Gaussian target runs in the coordinates a = L'(x - mu), where the step is element-wise
    (omc_mala_step_white); False keeps it on the products of omc_mala_step."""

    whitened: bool = True

    def _diag_step(self, current_state: dict) -> dict:
        """Generic route when the Hessian is diagonal per chain (e.g. the mixture-Normal prior of a variable-size
        coefficient vector under a null likelihood): metropolis_hastings.py:301-373 with omc_mala_diag for the proposal
        and its two densities, then the generic accept/reject with the sampler's model."""
        eng = self.engine
        x = current_state[self.param]
        if not is_chain(x) or x.shape[1] != 1:
            raise NotImplementedError("ManifoldMALA needs a per-chain (d, 1) parameter")
        step = float(self.step.item())
        count = x.count(current_state) if (x.ragged is not None and x.ragged[1] == 0) else None
        grad, h = self.model.grad_log_p_diag(current_state, self.param, eng)
        z = self.inject(self, self._sweep) if self.inject is not None else None
        xp, lq_f = eng.mala_diag(x.vector(), grad, h, step, count=count, z=z, draw_index=self._draw_index(), sub=0)
        prop_state = dict(current_state)
        prop_state[self.param] = x.like(xp.unsqueeze(2))
        grad_p, h_p = self.model.grad_log_p_diag(prop_state, self.param, eng)
        lq_r = eng.mala_diag(xp, grad_p, h_p, step, count=count, x_other=x.vector().contiguous())
        u = self.inject_uniform(self, self._sweep) if self.inject_uniform is not None else None
        return self._accept_reject_proposal(current_state, prop_state, lq_f, lq_r, u=u, sub=(x.shape[0] + 1) // 2 + 1)

    def _dense_step(self, current_state: dict) -> dict:
        """Generic route for a Hessian that is a per-chain combination of shared matrices (+ a per-chain diagonal):
        regression coefficients under a Gaussian or a mixture prior.  Lambda_c = H_c / step^2 is factorised per chain by
        the dense route (omc_dense_sample_canonical with b = Lambda x + g/2 gives x' = m + L^-T z and m at once, again
        with z = 0 for the reverse mean); log q = (1/2) log det Lambda - (1/2)(. - m)' Lambda (. - m)
        (metropolis_hastings.py:301-373), the quadratic forms term by term as GEMMs."""
        eng = self.engine
        x = current_state[self.param]
        xv = x.vector().contiguous()
        Cn, p = xv.shape
        s2 = float(self.step.item()) ** 2
        grad, terms, diag = self.model.grad_terms(current_state, self.param, eng)
        dterms = []
        for t in terms:
            sc = eng.full((Cn,), 1.0 / s2) if t["scale"] is None else t["scale"] / s2
            dterms.append({"mat": None if t["mat"] is None else eng.shared(t["mat"]), "scale": sc})
        if not dterms:
            dterms.append({"mat": None, "scale": eng.zeros(Cn)})
        dscaled = None if diag is None else diag / s2
        T = eng.dense_terms(dterms, p)

        def lam_times(v):   # Lambda_c v_c for every chain
            out = None if dscaled is None else dscaled * v
            for t in dterms:
                part = (v if t["mat"] is None else eng.design_predict(t["mat"], v.contiguous())) * t["scale"].unsqueeze(1)
                out = part if out is None else out + part
            return out

        def lam_quad(d):    # d_c' Lambda_c d_c
            out = None if dscaled is None else (dscaled * d * d).sum(dim=1)
            for t in dterms:
                q = (d * d).sum(dim=1) if t["mat"] is None else eng.dense_quadform(t["mat"], d.contiguous())
                out = q * t["scale"] if out is None else out + q * t["scale"]
            return out

        z = self.inject(self, self._sweep) if self.inject is not None else None
        xp, mu, logdet = eng.empty(Cn, p), eng.empty(Cn, p), eng.empty(Cn)
        eng.dense_sample_canonical(p, T, xp, z=z, rhs_chain=_lin(eng, 1.0, lam_times(xv), 0.5, grad), draw_index=self._draw_index(),
                                   mean_out=mu, logdet_out=logdet, diag_chain=dscaled)
        lq_f = _lin(eng, 0.5, logdet, -0.5, lam_quad(_lin(eng, 1.0, xp, -1.0, mu)))
        prop_state = dict(current_state)
        prop_state[self.param] = x.like(xp.unsqueeze(2))
        grad_p, _, _ = self.model.grad_terms(prop_state, self.param, eng)   # the Hessian does not depend on the parameter
        mu_p, scratch = eng.empty(Cn, p), eng.empty(Cn, p)
        eng.dense_sample_canonical(p, T, scratch, z=eng.zeros(Cn, p), rhs_chain=_lin(eng, 1.0, lam_times(xp), 0.5, grad_p),
                                   mean_out=mu_p, diag_chain=dscaled)
        lq_r = _lin(eng, 0.5, logdet, -0.5, lam_quad(_lin(eng, 1.0, xv, -1.0, mu_p)))
        u = self.inject_uniform(self, self._sweep) if self.inject_uniform is not None else None
        return self._accept_reject_proposal(current_state, prop_state, lq_f, lq_r, u=u, sub=(p + 1) // 2 + 1)

    def _grad_hess_per_chain(self, state):
        """(grad (C, d), H (C, d, d)) of the whole model w.r.t. the parameter, for every chain, whatever each member
        distribution returns (model.py:72-112): a shared matrix, a per-chain scalar x shared matrix (ScaledHessian) or the
        (C, d, d) tensor of the finite-difference default (distribution.py:124-198)."""
        import torch
        from scipy import sparse

        from openmcmc_amd.distribution.location_scale import ScaledHessian

        eng = self.engine
        x = state[self.param]
        Cn, d = x.n_chains, x.size
        grad, H = None, eng.zeros(Cn, d, d)
        for dist in self.model.values():
            if self.param not in dist.param_list:
                continue
            g, h = dist.grad_log_p(state, self.param, hessian_required=True, engine=eng)
            grad = g.data.reshape(Cn, d) if grad is None else grad + g.data.reshape(Cn, d)
            if isinstance(h, ScaledHessian):
                M = h.matrix.toarray() if sparse.issparse(h.matrix) else np.asarray(h.matrix, dtype=np.float64)
                H = H + h.scale.reshape(Cn, 1, 1) * eng.to_device(M).unsqueeze(0)
            elif isinstance(h, torch.Tensor):
                H = H + h.reshape(Cn, d, d)
            else:
                M = h.toarray() if sparse.issparse(h) else np.asarray(h, dtype=np.float64)
                H = H + eng.to_device(M).unsqueeze(0)
        if grad is None:
            raise ValueError(f"no distribution depends on '{self.param}'")
        return grad.contiguous(), H.contiguous()

    def _general_step(self, current_state: dict) -> dict:
        """Any model whose device log_p exists, for a fixed-size parameter: gradient and Hessian of every
        member as the reference's generic path computes them (analytic where a distribution has one, central differences
        otherwise, distribution.py:90-198), a per-chain Lambda_c = H_c / step^2 factorised per chain in natural order
        (omc_small_sample_canonical: x' = m + L^-T z and m at once; again with z = 0 at the proposed state for the reverse
        mean), log q = (1/2) log det Lambda - (1/2)(. - m)' Lambda (. - m) (metropolis_hastings.py:301-373)."""
        eng = self.engine
        x = current_state[self.param]
        if x.ragged is not None or x.shape[1] != 1:
            raise NotImplementedError("generic ManifoldMALA route: fixed-size (d, 1) parameter")
        xv = x.vector().contiguous()
        Cn, d = xv.shape
        s2 = float(self.step.item()) ** 2
        zero = eng.zeros(Cn, d)
        # up to one wave's 64 columns the library's small-matrix kernels (one wave per chain); beyond, the same operations as
        # batched dense factorisations (Engine.chain_spd_ops / chain_sample_canonical): the reference has no size limit here
        # (metropolis_hastings.py:325-348)
        if d <= 64:
            spd_ops = eng.small_spd_ops

            def draw(Lam_, b_, z_, mean_out):
                return eng.small_sample_canonical(Lam_, b_, zero, z=z_, draw_index=self._draw_index(), mean_out=mean_out)
        else:
            spd_ops = eng.chain_spd_ops

            def draw(Lam_, b_, z_, mean_out):
                return eng.chain_sample_canonical(Lam_, b_, z=z_, draw_index=self._draw_index(), mean_out=mean_out)
        grad, H = self._grad_hess_per_chain(current_state)
        Lam = H / s2
        Lx, _, logdet_f = spd_ops(Lam, xv, want_Av=True, want_logdet=True)
        z = self.inject(self, self._sweep) if self.inject is not None else None
        mu_f = eng.empty(Cn, d)
        xp = draw(Lam, _lin(eng, 1.0, Lx, 0.5, grad), z, mu_f)
        _, quad_f, _ = spd_ops(Lam, _lin(eng, 1.0, xp, -1.0, mu_f), want_quad=True)
        lq_f = _lin(eng, 0.5, logdet_f, -0.5, quad_f)
        prop_state = dict(current_state)
        prop_state[self.param] = x.like(xp.unsqueeze(2))
        grad_p, H_p = self._grad_hess_per_chain(prop_state)
        Lam_p = H_p / s2
        Lxp, _, logdet_r = spd_ops(Lam_p, xp, want_Av=True, want_logdet=True)
        mu_r = eng.empty(Cn, d)
        draw(Lam_p, _lin(eng, 1.0, Lxp, 0.5, grad_p), zero, mu_r)
        _, quad_r, _ = spd_ops(Lam_p, _lin(eng, 1.0, xv, -1.0, mu_r), want_quad=True)
        lq_r = _lin(eng, 0.5, logdet_r, -0.5, quad_r)
        if self.trace is not None:
            self.trace.setdefault("steps", []).append({"prop": xp.clone(), "lq_fwd": lq_f.clone(), "lq_rev": lq_r.clone()})
        u = self.inject_uniform(self, self._sweep) if self.inject_uniform is not None else None
        return self._accept_reject_proposal(current_state, prop_state, lq_f, lq_r, u=u, sub=(d + 1) // 2 + 1)

    def can_run_block(self, state) -> bool:
        """May MCMC.run_mcmc hand this sampler whole blocks of iterations (omc_mala_run_white)?  The fused whitened route, no
        injected draws, no trace, and `sample` not replaced by the caller."""
        return (self.whitened and self.inject is None and self.inject_uniform is None and self.trace is None
                and getattr(self.sample, "__func__", None) is ManifoldMALA.sample and self._gaussian_target(state))

    def run_block(self, current_state: dict, n_steps: int, x_store=None, logp_store=None) -> dict:
        """n_steps calls of `sample` as one library call: blocks of 32 steps per launch, the state of every step into
        x_store (n_steps, C, d) and the target's log density there into logp_store (n_steps, C) when given (what
        sampler.store and mcmc.py:108 keep per iteration).  Same streams, same decisions, same states as the single steps."""
        eng = self._need_engine()
        Q, mu, d = self._target(current_state)
        step = float(self.step.item())
        L, sl = self._factor_plan(eng, current_state, Q, 1.0 / step**2)
        x = self._x(current_state)
        tag = self._white_state_tag(current_state, x, L, mu)
        eng.mala_run_white(mu, L, sl, step, x, int(n_steps), state_is_current=self._white_state_is_current(eng, x, tag),
                           draw_index0=self._draw_index(), draw_stride=self._n_samplers, x_store=x_store, logp_store=logp_store,
                           accept_count=self.accept_rate.accept, proposal_count=self.accept_rate.proposal,
                           log_p_out=self._log_p_buffer(eng))
        self._white_tag, self._white_serial = tag, eng._write_serial
        self.last_log_p = (self._log_p_buf, x)
        self._sweep += int(n_steps)
        return current_state

    def sample(self, current_state: dict) -> dict:
        self.last_log_p = None  # set again by the fused whitened steps only
        eng = self._need_engine()
        if not self._gaussian_target(current_state):
            x = current_state[self.param]
            diag_only = x.ragged is not None or all(
                getattr(d, "is_mixture", False) and k == self.param or type(d).__name__ == "NullDistribution"
                for k, d in self.model.items())
            if diag_only:
                current_state = self._diag_step(current_state)
            else:
                structured = all(hasattr(d, "grad_terms") and d.constant_hessian(self.param)
                                 for d in self.model.values() if self.param in d.param_list)
                current_state = self._dense_step(current_state) if structured else self._general_step(current_state)
            self._sweep += 1
            return current_state
        Q, mu, d = self._target(current_state)
        step = float(self.step.item())
        L, sl = self._factor_plan(eng, current_state, Q, 1.0 / step**2)  # chol(H/step^2), H = Q (metropolis_hastings.py:345-346)
        z = self.inject(self, self._sweep) if self.inject is not None else None
        u = self.inject_uniform(self, self._sweep) if self.inject_uniform is not None else None
        x = self._x(current_state)
        if self.whitened:
            # L = chol(Q / step^2): the step is element-wise in a = L'(x - mu) (omc_mala_step_white).  The library keeps a
            # for the x it wrote last; it may be reused if nobody else has written x since (torch's version counter sees
            # every write but the library's own).
            tag = self._white_state_tag(current_state, x, L, mu)
            eng.mala_step_white(mu, L, sl, step, x, state_is_current=self._white_state_is_current(eng, x, tag), z=z, u=u,
                                draw_index=self._draw_index(), accept_count=self.accept_rate.accept,
                                proposal_count=self.accept_rate.proposal, log_p_out=self._log_p_buffer(eng))
            self._white_tag, self._white_serial = tag, eng._write_serial
            self.last_log_p = (self._log_p_buf, x)  # the target's log density at the state just left in x (see MCMC.run_mcmc)
        else:
            eng.mala_step(Q, mu, L, sl, step, x, z=z, u=u, draw_index=self._draw_index(),
                          accept_count=self.accept_rate.accept, proposal_count=self.accept_rate.proposal)
        self._sweep += 1
        return current_state
