"""Tensor / distributed helpers used by the per-frame path."""
