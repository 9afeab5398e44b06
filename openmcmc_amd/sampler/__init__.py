"""Samplers of the hot path: NormalNormal and NormalGamma (conjugate Gibbs)."""
