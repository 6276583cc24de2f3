"""Thin host models around the drop-in modules: callers of the operator boundary, here only so that
step-level numbers and data-parallel runs can be produced without any reference Python.  Not part of the hot path."""
from .dit import DiT_MHLA, DiT_configs  # noqa: F401
from .gpt import GPT_MHLA, GPT_configs  # noqa: F401
from .wan import WanAttentionBlock_MHLA, WanStack_MHLA  # noqa: F401
