"""MCMC driver on chain-batched state (reference mcmc.py:19-115).

Same fields and loop as the reference plus `n_chains`, `seed`, `device`, `chain_id_offset`: C
independent chains advance together, one HIP launch per sampler per sweep -- or ONE launch per
sweep when the sampler list is the Gaussian-block pattern [NormalNormal(x), NormalGamma(s)...],
which `fuse=True` (default) recognises and hands to omc_gmrf_sweep.  Both routes draw from the
same random streams and give identical results.

store[param] is a device tensor (n_iter, C, size); `collect()` returns host arrays shaped
(C, size, n_iter), i.e. the reference's store per chain.

The store is the step directly behind the sampler loop (mcmc.py:105-111): it stays on the device,
`summary()` / `quantiles()` reduce it there, `collect(every=k)` / `gather(every=k)` move a thinned part only, and with
`store_ring=R` the device keeps a ring of R iteration slabs that a second stream drains -- to pinned host memory, or to the
root rank through the run's one collective (parallel.GatherSink) -- while the chains keep sampling: a run is then not bounded by
what 288 GB hold (about 3 400 stored iterations at cfg3).
"""

from copy import copy
from dataclasses import dataclass, field

import numpy as np
from scipy import sparse

from openmcmc_amd.chains import ChainArray, host_2d, is_chain
from openmcmc_amd.model import Model
from openmcmc_amd.parameter import ScaledMatrix
from openmcmc_amd.sampler.sampler import MCMCSampler, NormalGamma, NormalNormal


@dataclass
class MCMC:
    state: dict
    samplers: list
    model: Model
    n_burn: int = 5000
    n_iter: int = 5000
    n_thin: int = 1
    n_chains: int = 1
    seed: int = 0
    device: int = 0
    chain_id_offset: int = 0
    fuse: bool = True
    engine: object = None  # an existing Engine (e.g. one that user callbacks already hold); default: a new one
    store_ring: int = 0    # > 0: the device store is a ring of this many iteration slabs, drained in halves while sampling goes on
    sink: object = None    # ring mode: callable(key, it0, it1, device_block) run under the drain stream; default: pinned host arrays
    store: dict = field(default_factory=dict, init=False)

    def __post_init__(self):
        from openmcmc_amd.engine import Engine

        self.state = copy(self.state)
        for key, term in self.state.items():  # mcmc.py:65-76
            if sparse.issparse(term) or is_chain(term):
                continue
            self.state[key] = host_2d(term)
        if self.engine is None:
            self.engine = Engine(self.n_chains, seed=self.seed, device=self.device, chain_id_offset=self.chain_id_offset)
        elif self.engine.n_chains != self.n_chains:
            raise ValueError("engine holds a different number of chains")
        eng, C = self.engine, self.n_chains
        ns = len(self.samplers)
        # iteration slabs resident on the device: all of them, or a ring of two halves (one being filled, one being drained)
        self._n_dev, self._half, self._drain = self.n_iter, self.n_iter, None
        if self.store_ring:
            if int(self.store_ring) < 2:
                raise ValueError("store_ring must be at least 2 (two halves)")
            half = int(self.store_ring) // 2
            self._half, self._n_dev = (half, 2 * half) if 2 * half < self.n_iter else (max(1, self.n_iter), max(1, self.n_iter))
        for pos, sampler in enumerate(self.samplers):
            sampler.bind(eng, pos, ns)
            # what the rest of the sweep samples after this block (a Normal-Normal block does not take the fused quadratic
            # form of a term whose other side is about to be replaced)
            sampler._later_params = frozenset(s.param for s in self.samplers[pos + 1:])
            if sampler.param not in self.state:  # mcmc.py:79-80: draw the start from the prior
                self.state[sampler.param] = sampler.model[sampler.param].rvs(
                    self.state, engine=eng, draw_index=(1 << 40) + pos)
            elif not is_chain(self.state[sampler.param]):
                v = np.asarray(self.state[sampler.param], dtype=np.float64)
                self.state[sampler.param] = ChainArray(eng.to_device(np.broadcast_to(v, (C,) + v.shape).copy()))
            self.store = sampler.init_store(current_state=self.state, store=self.store, n_iterations=self._n_dev)
        if self.model.response is not None:
            for response in self.model.response.keys():
                self.store[response] = eng.full((self._n_dev, C, self.state[response].size), float("nan"))
        self.store["log_post"] = eng.full((self._n_dev, C), float("nan"))
        self._fused = self._fusion_plan() if self.fuse else None
        self._sweeps_done = 0
        if self.store_ring:
            self._drain = _RingDrain(self)

    def _check_stream(self):
        """The context issues every library call on the stream it was created with; the mirror's own torch operations run
        on torch's current stream.  Sampling under another current stream would let the two race."""
        import torch

        if torch.cuda.current_stream(self.engine.device).cuda_stream != self.engine._stream.cuda_stream:
            raise RuntimeError("the current torch stream is not the one the Engine was created under: create the Engine (or "
                               "the MCMC object) and call run_mcmc under the same torch.cuda.stream")

    # ------------------------------------------------------------------ fusion
    def _fusion_plan(self):
        """[NormalNormal(x), NormalGamma(s_1), ...] with every s_j the ScaledMatrix scalar of one of
        x's Gaussian terms, no fitted-value store, and the full model made of exactly those pieces."""
        if len(self.samplers) < 2 or self.model.response is not None:
            return None
        nn = self.samplers[0]
        if type(nn) is not NormalNormal or nn.inject is not None:
            return None
        gammas = self.samplers[1:]
        if any(type(g) is not NormalGamma for g in gammas):
            return None
        try:
            plan = nn.plan(self.state)
        except NotImplementedError:
            return None
        if plan["kind"] != "tridiag" or plan.get("offsets") or plan.get("chain_rhs") or plan.get("center_chain") or plan.get("replicated") \
                or plan.get("limits") is not None:
            return None
        term_of = {}
        for k, key in enumerate(plan["keys"]):
            prec = nn.model[key].precision
            if isinstance(prec, ScaledMatrix):
                term_of[prec.scalar] = k
        blocks = [None] * len(plan["keys"])
        for pos, g in enumerate(gammas, start=1):
            k = term_of.get(g.param)
            if k is None or g.normal_param != plan["keys"][k] or blocks[k] is not None:
                return None
            blocks[k] = (pos, g)
        expected = set(plan["keys"]) | {g.param for g in gammas}
        full_model = set(self.model.keys()) == expected
        return {"nn": nn, "plan": plan, "blocks": blocks, "log_post": full_model}

    def _fused_sweep(self, store_it):
        eng, f = self.engine, self._fused
        nn, plan = f["nn"], f["plan"]
        ns, t = len(self.samplers), nn._sweep
        n = plan["n"]
        specs = []
        for k, key in enumerate(plan["keys"]):
            dist = nn.model[key]
            st = dist.structure(self.state)
            x_or_y, m = (None, None)
            cache = eng._model_cache[(id(dist), id(st.matrix))]
            spec = {"enabled": False, "logdet": cache["logdet"]}
            if f["blocks"][k] is not None:
                pos, g = f["blocks"][k]
                a0, b0 = g.prior_shape_rate(self.state)
                spec.update(enabled=True, a0=a0, b0=b0, n_pos=st.n_pos, draw_index=t * ns + pos,
                            g=g.inject(g, g._sweep) if g.inject is not None else None,
                            store=self.store[g.param][store_it, :, 0] if store_it is not None else None)
            specs.append(spec)
        x_out = self.store[nn.param][store_it] if store_it is not None else self._scratch(n)
        lp = self.store["log_post"][store_it] if (store_it is not None and f["log_post"]) else None
        z = nn.inject(nn, nn._sweep) if getattr(nn, "inject", None) is not None else None  # (test hook set after construction)
        eng.gmrf_sweep(n, plan["terms"], specs, x_out, z=z, draw_index=t * ns, log_post_out=lp)
        self.state[nn.param] = ChainArray(x_out)
        for s in self.samplers:
            s._sweep += 1

    def _direct_slab(self, sampler, i_it):
        """The store slab of this iteration if `sampler` can draw straight into it (a fixed-size NormalNormal block whose
        store is the plain (n_iter, C, n) tensor): the store step then has nothing to copy."""
        if type(sampler) is not NormalNormal or sampler.max_variable_size is not None:
            return None
        if getattr(sampler.sample, "__func__", None) is not NormalNormal.sample:
            return None  # a replaced sample method (tests count the calls): it gets the reference's signature
        cur = self.state.get(sampler.param)
        st = self.store.get(sampler.param)
        if not is_chain(cur) or cur.ragged is not None or cur.shape[1] != 1 or st is None or st.dim() != 3 or st.shape[2] != cur.shape[0]:
            return None
        return st[i_it]

    def _scratch(self, n):
        if getattr(self, "_scratch_x", None) is None or self._scratch_x.shape[1] != n:
            self._scratch_x = self.engine.empty(self.n_chains, n)
        return self._scratch_x

    # ------------------------------------------------------------------ the loop (mcmc.py:87-115)
    def _run_fused_in_c(self):
        """The whole loop as one omc_gmrf_run call: possible when no draws are injected."""
        eng, f = self.engine, self._fused
        nn, plan = f["nn"], f["plan"]
        ns, n = len(self.samplers), plan["n"]
        specs = []
        for k, key in enumerate(plan["keys"]):
            dist = nn.model[key]
            st = dist.structure(self.state)
            spec = {"enabled": False, "logdet": eng._model_cache[(id(dist), id(st.matrix))]["logdet"]}
            if f["blocks"][k] is not None:
                pos, g = f["blocks"][k]
                a0, b0 = g.prior_shape_rate(self.state)
                spec.update(enabled=True, a0=a0, b0=b0, n_pos=st.n_pos, draw_index=pos,
                            store=self.store[g.param][:, :, 0])
            specs.append(spec)
        # one library call for the whole run -- or, with a ring store, one per half of the ring: the drain stream empties
        # the half just filled while the next call fills the other (same draw indices, same results as the single call)
        it, burn = 0, self.n_burn
        while it < self.n_iter:
            k = min(self._half, self.n_iter - it)
            if self._drain is not None:
                self._drain.acquire(it)
            eng.gmrf_run(n, plan["terms"], specs, burn, k, self.n_thin, self.store[nn.param],
                         self._scratch(n), draw_index0=nn._sweep * ns, draws_per_sweep=ns, first_slot=it % self._n_dev,
                         log_post_store=self.store["log_post"] if f["log_post"] else None)
            for s in self.samplers:
                s._sweep += (burn + k) * self.n_thin
            if self._drain is not None:
                self._drain.release(it, it + k)
            it, burn = it + k, 0
        last = self.store[nn.param][(self.n_iter - 1) % self._n_dev] if self.n_iter > 0 else self._scratch(n)
        self.state[nn.param] = ChainArray(last)

    def _early_freeze_plan(self):
        """{sampler index: [(response key, predictor parameter)]}: stored LinearCombination predictors that may be evaluated
        right after that sampler because nothing later in the sweep changes their inputs.  Only when every later sampler is a
        conjugate one (it replaces its own parameter and nothing else) whose parameter the predictor does not use."""
        from openmcmc_amd.parameter import LinearCombination
        from openmcmc_amd.sampler.sampler import MixtureAllocation, NormalGamma, NormalNormal

        plan = {}
        if self._fused is not None or self.model.response is None:
            return plan
        for response, predictor in self.model.response.items():
            par = getattr(self.model[response], predictor)
            if not isinstance(par, LinearCombination):
                continue
            used = set(par.form.keys()) | set(par.form.values())
            touching = [k for k, s in enumerate(self.samplers)
                        if s.param in used or not isinstance(s, (NormalNormal, NormalGamma, MixtureAllocation))]
            if not touching:
                continue
            k = max(touching)
            if k == len(self.samplers) - 1 or self.samplers[k].param not in used:
                continue  # nothing to gain, or the last sampler that matters is one that may change anything
            if not isinstance(self.samplers[k], (NormalNormal, NormalGamma, MixtureAllocation)):
                continue
            plan.setdefault(k, []).append((response, par))
        return plan

    def run_mcmc(self):
        eng = self.engine
        self._check_stream()
        if not self.store_ring:  # (a caller may have re-sized n_iter and the store after construction)
            self._n_dev = self._half = self.n_iter
        if (self._fused is not None and self._fused["log_post"] and self.n_iter > 0
                and all(getattr(s, "inject", None) is None for s in self.samplers)):
            self._run_fused_in_c()
            eng.check_status()
            if self._drain is not None:
                self._drain.wait()  # like the reference, run_mcmc returns with the whole store where the user reads it
            return
        if self._mala_block_route():
            self._run_mala_blocks()
            eng.check_status()
            if self._drain is not None:
                self._drain.wait()
            self._print_acceptance()
            return
        early = self._early_freeze_plan()
        for i_it in range(-self.n_burn, self.n_iter):
            storing = i_it >= 0
            slot = i_it % self._n_dev if storing else None  # the iteration's slab on the device (a ring with store_ring)
            if storing and self._drain is not None and i_it % self._half == 0:
                self._drain.acquire(i_it)
            for i_thin in range(self.n_thin):
                last = i_thin == self.n_thin - 1
                if self._fused is not None:
                    self._fused_sweep(slot if (storing and last) else None)
                else:
                    for k, sampler in enumerate(self.samplers):
                        slab = self._direct_slab(sampler, slot) if (storing and last) else None
                        self.state = sampler.sample(self.state) if slab is None else sampler.sample(self.state, out=slab)
                        if storing and last:
                            # a stored predictor whose inputs no later sampler of the sweep touches: evaluated here, into
                            # its store slab, and reused by the samplers that follow (a NormalGamma's residual), by the store
                            # and by log_post
                            for response, par in early.get(k, ()):
                                par._frozen = {}
                                par.predictor_device(self.state, eng, out=self.store[response][slot])
            if not storing:
                continue
            if self._fused is None:
                for sampler in self.samplers:
                    self.store = sampler.store(current_state=self.state, store=self.store, iteration=slot)
            # The state does not change any more in this sweep: the fitted values go straight into their store slab and
            # log_post's residual reads them from there (mcmc.py:99-111 evaluates the predictor once for each; for cfg2 that
            # is a 5 GFLOP product per evaluation)
            frozen = []
            if self.model.response is not None:
                for response, predictor in self.model.response.items():
                    par = getattr(self.model[response], predictor)
                    if hasattr(par, "predictor_device") and any(is_chain(self.state[k]) for k in par.form):
                        if getattr(par, "_frozen", None) is None:
                            par._frozen = {}
                        frozen.append(par)
                        par.predictor_device(self.state, eng, out=self.store[response][slot])  # (a no-op when evaluated early)
                        continue
                    fitted = par.predictor(self.state)
                    if is_chain(fitted):
                        eng.chain_copy(fitted.data.reshape(self.n_chains, -1), self.store[response][slot])
                    else:
                        self.store[response][slot].copy_(eng.to_device(np.asarray(fitted).reshape(1, -1)).expand(self.n_chains, -1))
            try:
                if self._fused is None or not self._fused["log_post"]:
                    # one sampler on a one-distribution model whose fused step has just left the target's log density of
                    # this very state behind (the whitened MALA / random-walk steps): that IS the model's log_p
                    lp = getattr(self.samplers[0], "last_log_p", None) if len(self.samplers) == 1 and len(self.model) == 1 else None
                    cur = self.state.get(self.samplers[0].param) if lp is not None else None
                    if lp is not None and is_chain(cur) and cur.data.data_ptr() == lp[1].data_ptr():
                        self.store["log_post"][slot].copy_(lp[0])
                    else:
                        self.model.log_p(self.state, engine=eng, out=self.store["log_post"][slot])
            finally:
                for par in frozen:
                    par._frozen = None
            if self._drain is not None and ((i_it + 1) % self._half == 0 or i_it == self.n_iter - 1):
                self._drain.release(i_it - i_it % self._half, i_it + 1)
        eng.check_status()  # raises numpy.linalg.LinAlgError like gmrf.py:518 if a factorisation failed
        if self._drain is not None:
            self._drain.wait()
        self._print_acceptance()

    def _print_acceptance(self):
        from openmcmc_amd.sampler.metropolis_hastings import MetropolisHastings

        for sampler in self.samplers:  # mcmc.py:113-115
            if isinstance(sampler, MetropolisHastings):
                print(f"{sampler.param}: {sampler.accept_rate.get_acceptance_rate()}")

    # ------------------------------------------------------------------ one ManifoldMALA sampler on a Gaussian target (cfg4)
    def _mala_block_route(self):
        """[ManifoldMALA(x)] alone on the one-Normal model of the fused whitened step, every iteration stored: the loop is then
        blocks of steps issued by the library (omc_mala_run_white), the store slabs written by one product per block."""
        from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA

        if len(self.samplers) != 1 or self.n_thin != 1 or self.model.response is not None or len(self.model) != 1:
            return False
        smp = self.samplers[0]
        if type(smp) is not ManifoldMALA or smp.max_variable_size is not None or not smp.can_run_block(self.state):
            return False
        cur, st = self.state.get(smp.param), self.store.get(smp.param)
        return (is_chain(cur) and cur.ragged is None and cur.shape[1] == 1 and st is not None and st.dim() == 3
                and st.shape[2] == cur.shape[0] and st.is_contiguous())

    def _run_mala_blocks(self):
        smp = self.samplers[0]
        if self.n_burn > 0:
            self.state = smp.run_block(self.state, self.n_burn)  # nothing stored: no product until the last step
        it = 0
        while it < self.n_iter:
            k = min(self._half, self.n_iter - it)
            if self._drain is not None:
                self._drain.acquire(it)
            lo = it % self._n_dev
            self.state = smp.run_block(self.state, k, x_store=self.store[smp.param][lo: lo + k],
                                       logp_store=self.store["log_post"][lo: lo + k])
            if self._drain is not None:
                self._drain.release(it, it + k)
            it += k

    # ------------------------------------------------------------------ results
    def _whole_store_on_device(self, what):
        if self.store_ring and self._n_dev != self.n_iter:
            raise ValueError(f"{what} reduces the device store, and with store_ring the device holds the last {self._n_dev} of "
                             f"{self.n_iter} iterations only: reduce the drained store (host_store) instead")

    def summary(self, key, pooled=True):
        """Posterior mean and variance of store[key] computed on the device (no gather of the store):
        pooled over chains and iterations -> ((size,), (size,)), else per chain -> ((C, size), (C, size))."""
        self._whole_store_on_device("summary")
        t = self.store[key]
        t = t.unsqueeze(-1) if t.dim() == 2 else t.reshape(t.shape[0], t.shape[1], -1)
        mean, var = self.engine.store_moments(t.contiguous(), pooled=pooled)
        return mean.cpu().numpy(), var.cpu().numpy()

    def quantiles(self, key, q, pooled=True, omit_nan=True):
        """np.quantile(..., q) of store[key] over the stored iterations, computed on the device (exact order statistics by
        radix refinement, numpy's default "linear" interpolation; the store is neither sorted nor moved):
        pooled over chains and iterations -> (len(q), size), else per chain -> (len(q), C, size) -- what a user of the
        reference gets from np.quantile(mcmc.store[key], q, axis=-1) per chain.  omit_nan: the NaN padding of variable-size
        parameters is left out (np.nanquantile); False propagates NaN like np.quantile."""
        self._whole_store_on_device("quantiles")
        t = self.store[key]
        t = t.unsqueeze(-1) if t.dim() == 2 else t.reshape(t.shape[0], t.shape[1], -1)
        return self.engine.store_quantiles(t.contiguous(), q, pooled=pooled, omit_nan=omit_nan).cpu().numpy()

    def _thinned(self, every):
        """{key: device tensor (ceil(n_iter / every), C, ...)}: every `every`-th stored iteration, packed on the device"""
        self._whole_store_on_device("a thinned transfer")
        if int(every) <= 1:
            return self.store
        return {key: self.engine.store_thin(t.contiguous(), every) for key, t in self.store.items()}

    def collect(self, every=1):
        """Host copy of the store in the reference's per-chain layout: {key: (C, size, n_iter)},
        log_post: (C, n_iter, 1).  every=k: iterations 0, k, 2k, ... only (thinned on the device before the transfer).
        With store_ring: the drained store (every stored iteration, from pinned host memory)."""
        from openmcmc_amd.parallel import store_to_reference_layout

        if self._drain is not None and self._n_dev != self.n_iter:
            host = self.host_store
            return {key: store_to_reference_layout(key, t.numpy()[:: int(every)]) for key, t in host.items()}
        return {key: store_to_reference_layout(key, t.detach().cpu().numpy()) for key, t in self._thinned(every).items()}

    @property
    def host_store(self):
        """store_ring with the default sink: {key: pinned host tensor (n_iter, C, ...)} once run_mcmc has returned"""
        if self._drain is None or self._drain.host is None:
            raise ValueError("no drained store: run with store_ring and the default sink")
        self._drain.wait()
        return self._drain.host

    def gather(self, dst=0, comm=None, group=None, every=1):
        """The one collective of the path: gather every rank's store on rank `dst` over RCCL (xGMI).
        Returns the host dict of `collect()` for all chains on dst, None elsewhere.  `comm`: the library's own
        communicator (parallel.make_communicator(self.engine)) -> omc_gather_samples; "auto" makes one when the
        process group runs on RCCL; None -> torch.distributed's gather on the group's backend.  every=k: a thinned gather
        (iterations 0, k, 2k, ...: 1/k of the bytes over the links).  A run with store_ring gathers WHILE it samples:
        give it sink=parallel.GatherSink(...) instead."""
        import torch.distributed as dist

        from openmcmc_amd.parallel import gather_store, make_communicator

        if comm == "auto":
            comm = None
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1 and dist.get_backend(group) == "nccl":
                comm = make_communicator(self.engine, group)
        return gather_store(self._thinned(every), dst=dst, group=group, comm=comm)


class _RingDrain:
    """The ring store's second stream.  Iterations are cut into chunks of `half` slabs; chunk j lives in half j % 2 of the ring.
    release(): the chunk just filled is handed to the sink on the drain stream (behind an event of the sampling stream);
    acquire(): before the sampling stream writes into a half again it waits -- on the device, not on the host -- for the
    event that closed that half's drain.  The host never blocks inside the loop."""

    def __init__(self, mcmc):
        import torch

        self.m = mcmc
        self.dev = mcmc.engine.device
        self.stream = torch.cuda.Stream(device=self.dev)
        self.done = [None, None]
        self.sink = mcmc.sink
        self.host = None
        if self.sink is None:
            self.host = {key: torch.empty((mcmc.n_iter,) + tuple(t.shape[1:]), dtype=t.dtype, pin_memory=True)
                         for key, t in mcmc.store.items()}
        if self.sink is not None and hasattr(self.sink, "bind"):
            self.sink.bind(mcmc, self.stream)
        # entries whose slabs rely on the NaN fill beyond the live part (variable-size parameters, sampler.py:81-87, 105-116)
        self.refill = [s.param for s in mcmc.samplers if getattr(s, "max_variable_size", None) is not None]

    def acquire(self, it0):
        import torch

        h = (it0 // self.m._half) % 2
        ev = self.done[h]
        if ev is None:
            return
        torch.cuda.current_stream(self.dev).wait_event(ev)
        lo = it0 % self.m._n_dev
        for key in self.refill:  # a reused slab starts as the reference's fresh store does: NaN
            self.m.store[key][lo: lo + self.m._half].fill_(float("nan"))

    def release(self, it0, it1):
        import torch

        m = self.m
        h = (it0 // m._half) % 2
        filled = torch.cuda.Event()
        filled.record(torch.cuda.current_stream(self.dev))
        lo = it0 % m._n_dev
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(filled)
            for key, t in m.store.items():
                block = t[lo: lo + (it1 - it0)]
                if self.sink is None:
                    self.host[key][it0:it1].copy_(block, non_blocking=True)
                else:
                    self.sink(key, it0, it1, block)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.done[h] = ev

    def wait(self):
        self.stream.synchronize()
