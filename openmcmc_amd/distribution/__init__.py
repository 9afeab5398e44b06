"""Distributions of the hot path: Normal (location_scale) and Gamma (distribution)."""
