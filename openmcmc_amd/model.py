"""Model: dict of distributions keyed by response (reference model.py:21-112)."""

from openmcmc_amd.distribution.distribution import Distribution


class Model(dict):
    """Same constructor and sub-model selection as the reference (model.py:34-55)."""

    def __init__(self, distributions: list, response: dict = None):
        super().__init__({dist.response: dist for dist in distributions})
        self.response = response

    def conditional(self, param: str):
        """Distributions that mention `param` (model.py:41-55)."""
        return Model([dst for dst in self.values() if param in dst.param_list])

    def affected_by(self, changed_keys) -> list:
        """Responses of the member distributions that read any of the given state entries."""
        changed = set(changed_keys)
        return [key for key, dst in self.items() if changed.intersection(dst.param_list)]

    def log_p(self, state: dict, engine=None, out=None, members=None):
        """Sum of the members' log densities, one value per chain (model.py:57-70).  `members`: restrict the sum to
        these responses (Metropolis-Hastings needs only the terms a proposal can change: the others are the same
        number on both sides of the acceptance ratio)."""
        if engine is None:
            raise RuntimeError("Model.log_p needs the engine that holds the chains")
        out = engine.empty(engine.n_chains) if out is None else out
        if self._log_p_in_one_launch(state, engine, out, members):
            return out
        first = True
        host_sum = 0.0
        for key, dst in self.items():
            if members is not None and key not in members:
                continue
            from openmcmc_amd.chains import is_chain

            touches_chain = any(is_chain(state.get(k)) for k in dst.param_list)
            if not touches_chain:
                host_sum += dst.log_p(state)
                continue
            dst.log_p(state, engine=engine, out=out, accumulate=not first)
            first = False
        if first:
            out.fill_(host_sum)
        elif host_sum:
            out += host_sum
        return out


    def _log_p_in_one_launch(self, state, engine, out, members):
        """The sum as ONE launch (omc_log_post_sum) when every member that reads per-chain state can hand over its piece -- a
        Normal with a scalar x shared-matrix precision (its residual quadratic form cached by the draw or computed here), a Gamma
        on a per-chain scalar -- in the members' order and with each piece's own arithmetic, so the value is the one the
        member-by-member loop gives.  False: nothing was written, the loop takes over."""
        import os

        from openmcmc_amd.chains import is_chain

        if os.environ.get("OMC_NO_FUSED_LOGP"):  # (A/B switch)
            return False
        device = [dst for key, dst in self.items() if (members is None or key in members)
                  and any(is_chain(state.get(k)) for k in dst.param_list)]
        # (one piece: the member's own kernel is the same single launch; eligibility first, before anything is computed)
        try:
            if not 2 <= len(device) <= 8 or not all(hasattr(d, "log_p_piece") and d.log_p_piece(state, engine, dry=True) for d in device):
                return False
        except NotImplementedError:  # (a structure the member-by-member loop reports in its own words)
            return False
        pieces, host_sum = [], 0.0
        for key, dst in self.items():
            if members is not None and key not in members:
                continue
            if not any(is_chain(state.get(k)) for k in dst.param_list):
                host_sum += dst.log_p(state)
            else:
                pieces.append(dst.log_p_piece(state, engine))
        engine.log_post_sum(pieces, host_sum, out)
        return True

    def grad_terms(self, state: dict, param: str, engine):
        """(grad (C, p), terms, diag): the members' gradients summed and their Hessians collected as per-chain-scalar x
        shared-matrix terms plus an optional per-chain diagonal (model.py:72-112 for Hessians that are constant in
        `param`): what the dense ManifoldMALA route factorises per chain."""
        grad, terms, diag = None, [], None
        for dst in self.values():
            if not hasattr(dst, "grad_terms"):
                if param in dst.param_list:
                    raise NotImplementedError(f"{type(dst).__name__}: no structured Hessian")
                continue
            part = dst.grad_terms(state, param, engine)
            if part is None:
                continue
            g, t, d = part
            grad = g if grad is None else grad + g
            terms += t
            if d is not None:
                diag = d if diag is None else diag + d
        if grad is None:
            raise ValueError(f"no distribution depends on '{param}'")
        return grad, terms, diag

    def grad_log_p_diag(self, state: dict, param: str, engine):
        """(grad, hdiag) summed over the members when every contribution has a per-chain DIAGONAL Hessian
        (model.py:72-112 restricted to that structure; distributions raise NotImplementedError otherwise)."""
        grad, hdiag = None, None
        for dst in self.values():
            part = dst.grad_log_p_diag(state, param, engine) if hasattr(dst, "grad_log_p_diag") else None
            if part is None:
                if param in dst.param_list and not hasattr(dst, "grad_log_p_diag"):
                    raise NotImplementedError(f"{type(dst).__name__}: no diagonal-Hessian gradient")
                continue
            g, h = part
            grad, hdiag = (g, h) if grad is None else (grad + g, hdiag + h)
        if grad is None:
            raise ValueError(f"no distribution depends on '{param}'")
        return grad, hdiag

    def _grad_log_p(self, state: dict, param: str, hessian_required: bool = True, engine=None):
        grad, hess = None, None
        for dst in self.values():
            out = dst.grad_log_p(state, param, hessian_required=hessian_required, engine=engine)
            g, h = out if hessian_required else (out, None)
            if grad is None:
                grad, hess = g, h
            else:
                from openmcmc_amd.chains import ChainArray

                grad = ChainArray(grad.data + g.data)
                hess = hess + h if hessian_required else None
        return (grad, hess) if hessian_required else grad


Model.grad_log_p = Model._grad_log_p  # summed over the member distributions (model.py:72-112)

__all__ = ["Model", "Distribution"]
