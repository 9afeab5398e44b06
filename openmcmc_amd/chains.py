"""Chain-batched state values.

The reference keeps every state entry as a (p, n_rep) ndarray or a sparse matrix
(mcmc.py:65-76).  Here an entry is either
  * a host constant (ndarray / scipy.sparse), shared by all chains, exactly as in the reference, or
  * a ChainArray: one (p, n_rep) value per chain, resident on the GPU as a (C, p, n_rep) tensor.
"""

import numpy as np


class ChainArray:
    """(C, p, n_rep) float64 device tensor; `shape` reports the reference's per-chain (p, n_rep)."""

    __slots__ = ("data",)

    def __init__(self, data):
        if data.dim() == 2:
            data = data.unsqueeze(-1)
        if data.dim() != 3:
            raise ValueError("ChainArray wants (C, p) or (C, p, n_rep)")
        self.data = data

    @property
    def n_chains(self):
        return self.data.shape[0]

    @property
    def shape(self):
        return tuple(self.data.shape[1:])

    @property
    def size(self):
        return int(self.data.shape[1] * self.data.shape[2])

    def vector(self):
        """(C, p) contiguous view for vector-valued entries (n_rep == 1)."""
        if self.data.shape[2] != 1:
            raise ValueError("vector() needs n_rep == 1")
        return self.data[:, :, 0]

    def scalar(self):
        """(C,) view for scalar entries."""
        if self.size != 1:
            raise ValueError("scalar() needs a (1, 1) entry")
        return self.data[:, 0, 0]

    def numpy(self):
        return self.data.detach().cpu().numpy()

    def chain(self, c):
        """Host copy of one chain's (p, n_rep) value: what the reference would hold in state[key]."""
        return self.data[c].detach().cpu().numpy()

    def __repr__(self):
        return f"ChainArray(chains={self.n_chains}, shape={self.shape})"


def is_chain(value):
    return isinstance(value, ChainArray)


def host_2d(value):
    """The reference's coercion of a state entry (mcmc.py:69-76): >= 2-D float64, vectors as columns."""
    if not isinstance(value, np.ndarray):
        value = np.array(value, ndmin=2, dtype=np.float64)
        if value.shape[0] == 1:
            value = value.T
    elif value.ndim < 2:
        value = np.atleast_2d(value).T
    return value
