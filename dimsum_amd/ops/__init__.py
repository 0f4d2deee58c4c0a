"""Operator API of the hot path (same names as the reference's Python op modules)."""
from .causal_conv1d_interface import causal_conv1d_fn  # noqa: F401
from .layernorm import RMSNorm, layer_norm_fn, rms_norm_fn  # noqa: F401
from .selective_scan_interface import (  # noqa: F401
    bimamba_inner_fn,
    mamba_inner_fn,
    mamba_inner_fn_cond,
    mamba_inner_fn_no_out_proj,
    mamba_inner_fn_no_out_proj_cond,
    selective_scan_fn,
)
