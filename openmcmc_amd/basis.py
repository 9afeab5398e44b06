"""Gaussian-kernel basis on per-chain knots: the design matrix of the reference's reversible-jump example
(tests/test_reversible_jump.py:24-40, `make_basis`: one norm.pdf column per knot) as a library object.

An instance is what user model code passes where the reference takes callbacks:

    basis = GaussianKnotBasis(engine, X, knots="theta", matrix="B")
    state["B"] = basis.make(state)
    RandomWalkLoop("theta", model, ..., state_update_function=basis)           # metropolis_hastings.py:264-267
    ReversibleJump("n_basis", model, ..., state_birth_function=basis.birth)    # reversible_jump.py:123,170

Called as a `state_update_function` it does what a hand-written callback would (recompute the moved column of the
proposed state's basis).  Because the library can see WHAT the callback computes, RandomWalkLoop can hand the whole loop
over the knots to one kernel launch when the rest of the model has the matching shape (omc_knot_loop; see
RandomWalkLoop._knot_plan) -- same draws, same decisions up to the rounding of the log-likelihood difference."""

import numpy as np

from openmcmc_amd.chains import ChainArray


class GaussianKnotBasis:
    def __init__(self, engine, X, knots="theta", matrix="B", scale=1.0):
        import torch

        self.engine = engine
        self.X = torch.as_tensor(np.asarray(X, dtype=np.float64).reshape(-1), device=engine.device)
        self.knots, self.matrix, self.scale = knots, matrix, float(scale)
        self._buf, self._last = None, None
        self._prop, self._prop_count = None, None

    def make(self, state):
        """The whole basis of `state`: an (n, k_max) ChainArray, column-major per chain, zero columns beyond the live
        knots of each chain."""
        theta = state[self.knots]
        C, _, k_max = theta.data.shape
        out = self.engine.empty(C, k_max, self.X.numel())
        self.engine.gaussian_basis(self.X, theta.data[:, 0, :], out, count=theta.count(state), scale=self.scale)
        return ChainArray(out.transpose(1, 2), ragged=(theta.ragged[0], 1))

    def __call__(self, state, col):
        """state_update_function of RandomWalk / RandomWalkLoop: knot `col` of the proposed state moved.  The proposed
        basis differs from the current one in that column only; it lives in one scratch buffer that is cloned from
        the current basis at the first knot of a sweep and afterwards re-synchronised by copying back the single
        column the previous step may have left different (RandomWalkLoop visits 0, 1, 2, ...)."""
        if col is None:
            state[self.matrix] = self.make(state)
            return state, 0.0, 0.0
        store = state[self.matrix].columns()  # the CURRENT basis: `state` is a shallow copy of the current state
        buf = self._buf
        if buf is None or buf.shape != store.shape or col == 0 or self._last != col - 1:
            buf = self._buf = store.clone()
        else:
            buf[:, col - 1, :].copy_(store[:, col - 1, :])
        theta = state[self.knots]
        self.engine.gaussian_basis(self.X, theta.data[:, 0, :], buf, count=theta.count(state), scale=self.scale, column=col)
        self._last = col
        state[self.matrix] = state[self.matrix].like(buf.transpose(1, 2))
        return state, 0.0, 0.0

    def birth(self, current_state, prop_state):
        """state_birth_function of ReversibleJump (births and deaths alike): the basis of the proposed knots.  It is
        written into one buffer kept from sweep to sweep (the sampler copies an accepted proposal into the current
        state, it does not keep the proposal): the buffer is known to hold zeros beyond the live columns of the previous
        proposal, so only the live columns of this one -- a quarter of the k_max at BASELINE configs[4] -- and the
        columns that have just died are written."""
        theta = prop_state[self.knots]
        C, _, k_max = theta.data.shape
        count = theta.count(prop_state)
        shape = (C, k_max, self.X.numel())
        if self._prop is None or tuple(self._prop.shape) != shape:
            self._prop, self._prop_count = self.engine.zeros(*shape), self.engine.zeros(C)
        self.engine.gaussian_basis(self.X, theta.data[:, 0, :], self._prop, count=count, scale=self.scale,
                                   prev_count=self._prop_count)
        self._prop_count.copy_(count)
        prop_state[self.matrix] = ChainArray(self._prop.transpose(1, 2), ragged=(theta.ragged[0], 1))
        return prop_state, 0.0, 0.0
