"""Chain-batched state values.

The reference keeps every state entry as a (p, n_rep) ndarray or a sparse matrix
(mcmc.py:65-76).  Here an entry is either
  * a host constant (ndarray / scipy.sparse), shared by all chains, exactly as in the reference, or
  * a ChainArray: one (p, n_rep) value per chain, resident on the GPU as a (C, p, n_rep) tensor.

Variable-dimension entries (reversible jump: reversible_jump.py:126-131, 171-175 grow and shrink
state arrays by one element) are RAGGED ChainArrays: padded to the maximum size with zeros, with
`ragged = (count_key, axis)` naming the state entry that holds each chain's live length (the
reference's state["n_basis"]) and the axis it applies to (0: rows, as beta (k, 1); 1: columns, as
theta (1, k) or a basis matrix (n, k)).
"""

import numpy as np


class ChainArray:
    """(C, p, n_rep) float64 device tensor; `shape` reports the reference's per-chain (p, n_rep)."""

    __slots__ = ("data", "ragged")

    def __init__(self, data, ragged=None):
        if data.dim() == 2:
            data = data.unsqueeze(-1)
        if data.dim() != 3:
            raise ValueError("ChainArray wants (C, p) or (C, p, n_rep)")
        if ragged is not None and (len(ragged) != 2 or ragged[1] not in (0, 1)):
            raise ValueError("ragged = (count_key, axis) with axis 0 or 1")
        self.data = data
        self.ragged = ragged

    def like(self, data):
        """A ChainArray around `data` with this one's ragged description."""
        return ChainArray(data, ragged=self.ragged)

    def storage(self):
        """The contiguous tensor behind `data`: data itself, or its (C, n_rep, p) transpose for a matrix kept
        column-major per chain (a basis whose columns are the ragged axis)."""
        if self.data.is_contiguous():
            return self.data
        t = self.data.transpose(1, 2)
        if t.is_contiguous():
            return t
        raise ValueError("ChainArray data is neither row- nor column-major contiguous")

    def columns(self):
        """(C, n_rep, p) contiguous tensor: column j of chain c contiguous (the layout omc_design_*_batched take)."""
        t = self.data.transpose(1, 2)
        return t if t.is_contiguous() else t.contiguous()

    def count(self, state):
        """(C,) live lengths of a ragged entry (None for a fixed-size one)."""
        return None if self.ragged is None else state[self.ragged[0]].scalar()

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
        tail = "" if self.ragged is None else f", ragged={self.ragged}"
        return f"ChainArray(chains={self.n_chains}, shape={self.shape}{tail})"


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


def ragged_from_lists(values, n_max, axis, count_key, device):
    """Padded ragged ChainArray from one 1-D array per chain (lengths may differ): (C, n_max, 1) for
    axis 0, (C, 1, n_max) for axis 1, zeros beyond each chain's length."""
    import torch

    C = len(values)
    host = np.zeros((C, n_max))
    for c, v in enumerate(values):
        v = np.asarray(v, dtype=np.float64).reshape(-1)
        if v.size > n_max:
            raise ValueError("chain value longer than n_max")
        host[c, : v.size] = v
    t = torch.as_tensor(host, device=device)
    return ChainArray(t.unsqueeze(2) if axis == 0 else t.unsqueeze(1), ragged=(count_key, axis))
